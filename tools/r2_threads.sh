#!/bin/bash
# task-thread / batch mini sweep of the headline command (compress side)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/thr; : > gpurun_out/thr/sweep.txt
for spec in "2 -1" "4 -1" "3 -1" "2 2" "4 1" "2 -1"; do set -- $spec
  timeout 120 python bench.py --no-cpu-baseline --task-threads $1 --batch $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('threads $1 batch $2:', d['value'], 'GB/s', d['ms_per_step'], 'ms/step')" | tee -a gpurun_out/thr/sweep.txt
done
