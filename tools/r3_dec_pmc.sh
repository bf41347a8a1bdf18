#!/bin/bash
# LDS-side PMC passes of the batch decoder (round 3).  usage: tools/r3_dec_pmc.sh <tag> [bench args]
tag=$1; shift
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
P=$R/gpurun_out/$tag; mkdir -p $P
CMD="python $R/bench.py --no-cpu-baseline --no-secondary --direction decompress --maps-per-gpu 4 --steps 2 --warmup 1 $*"
cd /tmp
i=0
while read -r line; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $line -d $P/pmc_$i -o p -- $CMD > $P/pmc_$i.log 2>&1
done <<'PMC'
SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS
GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL
PMC
cd $R
python tools/summarize_prof.py $P > $P/summary.txt 2>&1
grep -A40 "batch_decode" $P/summary.txt | head -60
