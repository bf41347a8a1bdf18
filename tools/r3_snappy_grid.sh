#!/bin/bash
# Snappy compress as a persistent grid (experiment of profiles/r03_experiments.md §8; the kernel change itself was not kept — it is
# in the git history of spark-s3-shuffle_amd/csrc/snappy_compress.hip): parity, then wide rows with four task threads by grid
# size (S3S_SNAPPY_GRID), and a launch alone
R=$GRAFT_REPO_ROOT; cd $R
timeout 120 python -m pytest tests/test_gpu_snappy.py tests/test_gpu_batch.py -x -q 2>&1 | tail -2
one() { timeout 60 python bench.py --no-cpu-baseline --no-secondary --workload tpcds-wide-100g-200p-snappy "${@:2}" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['roofline']['avg_launch_ms'])" || echo "$1 FAILED"; }
one default_3_per_cu --verify
S3S_SNAPPY_GRID=1280 one grid_5_per_cu
S3S_SNAPPY_GRID=512 one grid_2_per_cu
one default_3_per_cu
one alone_2tasks --maps-per-gpu 2 --task-threads 1
S3S_SNAPPY_GRID=100000 one one_workgroup_per_block
