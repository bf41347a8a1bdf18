#!/bin/bash
# Round-2 measurement set (runs on the GPU box via gpurun): GPU tests, bench lines, rocprofv3 kernel stats and
# separate PMC passes for the four directions/codecs.  usage: tools/gpu_round2.sh <tag>
tag=${1:-r02}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
B="timeout 400 python bench.py --cpu-seconds 8"
$B --verify > $O/bench.json 2> $O/bench.err
$B --direction decompress > $O/bench_decompress.json 2>> $O/bench.err
$B --workload tpcds-wide-100g-200p-lz4 > $O/bench_tpcds_lz4.json 2>> $O/bench.err
$B --workload tpcds-wide-100g-200p-snappy --verify > $O/bench_snappy.json 2>> $O/bench.err
$B --no-cpu-baseline --workload tpcds-wide-100g-200p-snappy --direction decompress > $O/bench_snappy_decompress.json 2>> $O/bench.err
$B --no-cpu-baseline --workload tpcds-wide-100g-200p-lz4 --direction decompress > $O/bench_tpcds_lz4_decompress.json 2>> $O/bench.err
$B --no-cpu-baseline --workload terasort-100g-2000p-lz4-crc32 > $O/bench_2000p.json 2>> $O/bench.err
$B --no-cpu-baseline --workload terasort-100g-2000p-lz4-crc32 --direction decompress > $O/bench_2000p_decompress.json 2>> $O/bench.err
$B --no-cpu-baseline --workload skew-1part-lz4 --map-mib 1024 --maps-per-gpu 2 > $O/bench_skew1g.json 2>> $O/bench.err
$B --no-cpu-baseline --workload skew-1part-lz4 --map-mib 1024 --maps-per-gpu 2 --direction decompress > $O/bench_skew1g_decompress.json 2>> $O/bench.err
$B --no-cpu-baseline --lz4-variant 1 > $O/bench_lz4_variant1.json 2>> $O/bench.err
$B --no-cpu-baseline --direction decompress --lz4-decode-variant 3 > $O/bench_decompress_ring_decoder.json 2>> $O/bench.err
timeout 200 python tools/host_path_bench.py 1,3 4 > $O/host_path.txt 2>&1
timeout 200 python tools/checksum_bench.py > $O/checksum_bench.txt 2>&1
cat /sys/fs/cgroup/cpu.max > $O/host_cgroup.txt 2>&1; nproc >> $O/host_cgroup.txt; python -c "import os;print(len(os.sched_getaffinity(0)))" >> $O/host_cgroup.txt
prof() {  # prof <name> <bench args...>
  name=$1; shift
  P=$O/prof_$name; mkdir -p $P
  CMD="python $R/bench.py --no-cpu-baseline --maps-per-gpu 4 --steps 3 --warmup 1 $*"
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats -d $P/trace -o t -- $CMD > $P/trace.log 2>&1
  timeout 300 rocprofv3 --pmc FETCH_SIZE -d $P/pmc_fetch -o p -- $CMD > $P/pmc_fetch.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE -d $P/pmc_write -o p -- $CMD > $P/pmc_write.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $P/pmc_sq1 -o p -- $CMD > $P/pmc_sq1.log 2>&1
  cd $R
  python tools/summarize_prof.py $P --md > $P/summary.md 2>&1
}
prof compress
prof decompress --direction decompress
prof snappy_compress --workload tpcds-wide-100g-200p-snappy
prof snappy_decompress --workload tpcds-wide-100g-200p-snappy --direction decompress
prof crc32_2000p --workload terasort-100g-2000p-lz4-crc32
tail -3 $O/pytest_gpu.log; cat $O/smoke.log | tail -1
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
cat $O/host_path.txt | grep -v amdgpu; cat $O/checksum_bench.txt | grep -v amdgpu
