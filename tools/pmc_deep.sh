#!/bin/bash
# Deep PMC passes of the LZ4 compress kernel (round 3): vector-memory latency and the TA / TCP / TD path, instruction
# fetch, LDS conflicts.  usage: tools/pmc_deep.sh <tag> [bench args]
tag=$1; shift
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
P=$R/gpurun_out/$tag; mkdir -p $P
CMD="python $R/bench.py --no-cpu-baseline --maps-per-gpu 2 --task-threads 1 --steps 2 --warmup 1 $*"
cd /tmp
i=0
while read -r line; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $line -d $P/pmc_$i -o p -- $CMD > $P/pmc_$i.log 2>&1
done <<'PMC'
SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_LEVEL_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES
TA_TA_BUSY_sum TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
TCP_GATE_EN1_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum
TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_REQ SQ_IFETCH SQ_IFETCH_LEVEL SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_ANY
TD_TD_BUSY_sum TD_TC_STALL_sum SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES SQ_WAVES
PMC
cd $R
python tools/summarize_prof.py $P > $P/summary.txt 2>&1
grep -A60 "lz4_compress\|snappy_compress\|batch_decode" $P/summary.txt | head -100
