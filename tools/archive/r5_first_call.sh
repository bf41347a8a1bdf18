#!/bin/bash
# Round 5, first GPU call: the memory-path probe, then A/B of the LZ4 window block with the speculative next-window
# gather (-DS3S_ENGINE_SPEC, lib exp_spec) against the shipped block — shipped first and last — and three PMC passes
# of one compress launch pair for each of the two builds.
#   CPU side first:  make -C spark-s3-shuffle_amd/csrc exp EXPNAME=spec EXPFLAGS=-DS3S_ENGINE_SPEC
#                    (cd tools/probe && hipcc --offload-arch=gfx950 -O2 -o mem_latency_probe mem_latency_probe.hip)
#   usage (GPU): tools/r5_first_call.sh <tag> [libs...]
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
tag=${1:-r05a}; shift
LIBS=${*:-default spec default}
O=gpurun_out/$tag; mkdir -p $O
timeout 120 tools/probe/mem_latency_probe > $O/mem_latency_probe.txt 2>&1
cat $O/mem_latency_probe.txt
bash tools/ab.sh $tag $LIBS
for n in $(echo $LIBS | tr ' ' '\n' | sort -u); do
  if [ "$n" = default ]; then unset S3S_CODEC_LIB; else export S3S_CODEC_LIB=$R/spark-s3-shuffle_amd/lib/libs3shuffle_codec_exp_$n.so; fi
  P=$R/gpurun_out/$tag/pmc_$n; mkdir -p $P
  CMD="python $R/bench.py --no-cpu-baseline --no-secondary --maps-per-gpu 2 --task-threads 1 --steps 2 --warmup 1"
  i=0
  while read -r line; do
    i=$((i+1))
    (cd /tmp && timeout 300 rocprofv3 --pmc $line -d $P/pmc_$i -o p -- $CMD > $P/pmc_$i.log 2>&1)
  done <<'PMC'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_SMEM GRBM_GUI_ACTIVE
TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum
PMC
  python tools/summarize_prof.py $P > $P/summary.txt 2>&1
  echo "== PMC $n"; grep -A12 "lz4_compress" $P/summary.txt | head -40
done
