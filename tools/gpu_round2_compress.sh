#!/bin/bash
# Map-side measurement set after the hand-written window blocks (round 2, r02g..r02i): bench lines of the LZ4 and Snappy map
# side, compress block-size points, rocprofv3 kernel trace + separate PMC passes of the headline and the Snappy command.
tag=${1:-r02g}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
B="timeout 400 python bench.py --cpu-seconds 8"
$B --verify > $O/bench.json 2> $O/bench.err
$B --no-cpu-baseline --workload tpcds-wide-100g-200p-lz4 > $O/bench_tpcds_lz4.json 2>> $O/bench.err
$B --no-cpu-baseline --workload terasort-100g-2000p-lz4-crc32 > $O/bench_2000p.json 2>> $O/bench.err
$B --no-cpu-baseline --task-threads 1 > $O/bench_1thread.json 2>> $O/bench.err
$B --no-cpu-baseline --lz4-variant 1 > $O/bench_lz4_variant1.json 2>> $O/bench.err
$B --workload tpcds-wide-100g-200p-snappy --verify > $O/bench_snappy.json 2>> $O/bench.err
: > $O/sweep_compress.jsonl
for spec in "8 32 8" "32 16 4" "128 8 2" "512 2 2" "1024 2 2"; do set -- $spec
  timeout 300 python bench.py --no-cpu-baseline --workload skew-1part-lz4 --map-mib $1 --maps-per-gpu $2 --task-threads $3 --steps 5 --warmup 2 2>/dev/null >> $O/sweep_compress.jsonl
done
P=$O/prof_compress; mkdir -p $P
CMD="python $R/bench.py --no-cpu-baseline --maps-per-gpu 4 --steps 3 --warmup 1"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $P/trace -o t -- $CMD > $P/trace.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $P/pmc_fetch -o p -- $CMD > $P/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $P/pmc_write -o p -- $CMD > $P/pmc_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $P/pmc_sq1 -o p -- $CMD > $P/pmc_sq1.log 2>&1
cd $R
python tools/summarize_prof.py $P --md > $P/summary.md 2>&1
P2=$O/prof_snappy_compress; mkdir -p $P2
CMD2="python $R/bench.py --no-cpu-baseline --maps-per-gpu 4 --steps 3 --warmup 1 --workload tpcds-wide-100g-200p-snappy"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $P2/trace -o t -- $CMD2 > $P2/trace.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $P2/pmc_fetch -o p -- $CMD2 > $P2/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $P2/pmc_write -o p -- $CMD2 > $P2/pmc_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $P2/pmc_sq1 -o p -- $CMD2 > $P2/pmc_sq1.log 2>&1
cd $R
python tools/summarize_prof.py $P2 --md > $P2/summary.md 2>&1
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
python - <<'PY'
import json
for l in open("gpurun_out/%s/sweep_compress.jsonl" % "TAG".replace("TAG", __import__("os").environ.get("S3S_TAG", "r02g"))):
    d = json.loads(l); print("sweep", d["config"]["map_task_bytes"] >> 20, "MiB x", d["config"]["map_tasks_per_gpu"], ":", d["value"], "GB/s")
PY
head -30 $P/summary.md
