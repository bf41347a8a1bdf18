#!/bin/bash
# Round 5: A/B of window-block builds + PMC passes of some of them.  usage (GPU): tools/r5_ab_call.sh <tag> "<libs>" "<pmc libs>"
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
tag=$1
LIBS=$2
PMCLIBS=$3
O=gpurun_out/$tag; mkdir -p $O
bash tools/ab.sh $tag $LIBS
for n in $PMCLIBS; do
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
PMC
  python tools/summarize_prof.py $P > $P/summary.txt 2>&1
  echo "== PMC $n"; grep "lz4_compress" $P/summary.txt | cut -c1-600
done
