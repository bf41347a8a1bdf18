#!/bin/bash
# bench lines with the final defaults (four task threads on the map side)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02j; mkdir -p $O
B="timeout 300 python bench.py --cpu-seconds 8"
$B --verify > $O/bench.json 2> $O/bench.err
$B --no-cpu-baseline --workload tpcds-wide-100g-200p-lz4 > $O/bench_tpcds_lz4.json 2>> $O/bench.err
$B --no-cpu-baseline --workload terasort-100g-2000p-lz4-crc32 > $O/bench_2000p.json 2>> $O/bench.err
$B --no-cpu-baseline --workload tpcds-wide-100g-200p-snappy > $O/bench_snappy.json 2>> $O/bench.err
for f in $O/bench*.json; do python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); cb=d.get('cpu_baseline') or {}; print('$f'.split('/')[-1], d['value'], 'GB/s', d['ms_per_step'], 'ms/step, threads', d['config']['task_threads_per_gpu'], '| cpu', cb.get('value'))"; done
timeout 200 python -m pytest tests/test_gpu_launch.py -m gpu -q 2>&1 | tail -2
timeout 120 python bench.py --no-cpu-baseline --direction decompress 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('decompress', d['value'], 'GB/s', d['ms_per_step'], 'ms/step')"
