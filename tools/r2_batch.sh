#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_batch.py -x -q 2>&1 | tail -5
for args in "--batch 0 --task-threads 2" "--batch 0 --task-threads 1" "--task-threads 1" "--task-threads 2" ; do
  echo "== $args"; timeout 200 python bench.py --no-cpu-baseline --lz4-variant 10 $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'GB/s', d['ms_per_step'], 'ms/step', d['stages_ms_per_library_call'])"
done
echo "== 8 MiB blocks, batch of 32, variant 10"; timeout 200 python bench.py --no-cpu-baseline --lz4-variant 10 --workload skew-1part-lz4 --map-mib 8 --maps-per-gpu 32 --task-threads 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'GB/s', d['ms_per_step'], 'ms/step')"
