#!/bin/bash
# usage: tools/prof_pmc.sh <tag> -- <command...>   (run on the GPU box; writes gpurun_out/prof_<tag>/)
tag=$1; shift; shift
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp
rocprofv3 --kernel-trace --stats -d $out/trace -o t -- "$@" > $out/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $out/pmc1 -o p -- "$@" > $out/pmc1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM -d $out/pmc2 -o p -- "$@" > $out/pmc2.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum -d $out/pmc3 -o p -- "$@" > $out/pmc3.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $out/pmc4 -o p -- "$@" > $out/pmc4.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $out/pmc5 -o p -- "$@" > $out/pmc5.log 2>&1
cd $out && find . -name "*.csv" | head -40
