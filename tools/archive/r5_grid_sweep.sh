#!/bin/bash
# (GPU) round 5: launch-shape sweep of the headline with the deferred-emission kernels: persistent-grid size when calls overlap
# (S3S_LZ4_GRID; default 1280 = 5 wavefronts per CU per launch), task threads x map tasks per batched call
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
tag=${1:-r05q}
O=gpurun_out/$tag; mkdir -p $O
: > $O/sweep.txt
run() { # label, env, args
  env $2 timeout 200 python bench.py --no-cpu-baseline --no-secondary $3 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], 'GB/s', 'ms/step', d['ms_per_step'], 'codec ms', d['roofline']['avg_launch_ms'])" | tee -a $O/sweep.txt
}
run "default" "A=1" ""
for g in 1024 1536 1792 2048 2560; do run "grid=$g" "S3S_LZ4_GRID=$g" ""; done
run "default" "A=1" ""
run "threads=8,batch=1" "A=1" "--task-threads 8 --batch 1"
run "threads=2,batch=4" "A=1" "--task-threads 2 --batch 4"
run "threads=4,batch=1,maps=8" "A=1" "--task-threads 4 --batch 1"
run "threads=3,batch=2" "A=1" "--task-threads 3 --batch 2"
run "default" "A=1" ""
