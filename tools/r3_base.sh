#!/bin/bash
# Round 3, first GPU call: re-profile what ships (the r02k kernels) — bench lines + rocprofv3 kernel trace and separate
# PMC passes (FETCH / WRITE / SQ / TCC) of the LZ4 compress, Snappy compress and LZ4 decode commands.
tag=${1:-r03a}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
B="timeout 400 python bench.py --cpu-seconds 6"
$B > $O/bench.json 2> $O/bench.err
$B --no-cpu-baseline --workload tpcds-wide-100g-200p-lz4 > $O/bench_tpcds_lz4.json 2>> $O/bench.err
$B --no-cpu-baseline --workload terasort-100g-2000p-lz4-crc32 > $O/bench_2000p.json 2>> $O/bench.err
$B --no-cpu-baseline --workload tpcds-wide-100g-200p-snappy > $O/bench_snappy.json 2>> $O/bench.err
$B --no-cpu-baseline --direction decompress > $O/bench_decompress.json 2>> $O/bench.err
prof() {  # prof <name> <bench args...>
  local P=$O/prof_$1; shift; mkdir -p $P
  local CMD="python $R/bench.py --no-cpu-baseline --maps-per-gpu 4 --steps 3 --warmup 1 $*"
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats -d $P/trace -o t -- $CMD > $P/trace.log 2>&1
  timeout 300 rocprofv3 --pmc FETCH_SIZE -d $P/pmc_fetch -o p -- $CMD > $P/pmc_fetch.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE -d $P/pmc_write -o p -- $CMD > $P/pmc_write.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $P/pmc_sq1 -o p -- $CMD > $P/pmc_sq1.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_BUSY_CU_CYCLES -d $P/pmc_sq2 -o p -- $CMD > $P/pmc_sq2.log 2>&1
  timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $P/pmc_tcc -o p -- $CMD > $P/pmc_tcc.log 2>&1
  cd $R
  python tools/summarize_prof.py $P --md > $P/summary.md 2>&1
}
SETS=${2:-compress snappy_compress decompress}
for s in $SETS; do
  case $s in
    compress) prof compress ;;
    snappy_compress) prof snappy_compress --workload tpcds-wide-100g-200p-snappy ;;
    decompress) prof decompress --direction decompress ;;
  esac
done
for f in $O/bench*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    cb = d.get("cpu_baseline") or {}
    print(sys.argv[1].split("/")[-1], d["value"], "GB/s", d["ms_per_step"], "ms/step; roofline frac", d["roofline"]["frac"], "kernel ms", d["roofline"]["avg_launch_ms"], "| cpu", cb.get("value"), cb.get("cores"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
head -20 $O/prof_${SETS%% *}/summary.md | cut -c1-220
