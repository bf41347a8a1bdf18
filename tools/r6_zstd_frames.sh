#!/bin/bash
# (GPU) round 6: Zstandard reduce side by map outputs per step and task threads (each thread = one batched call over its share).
# The chip holds 1 024 three-wavefront workgroups = 2 048 partition frames: a step of 10 map outputs x 200 partitions is ONE round of them.
cd $GRAFT_REPO_ROOT; O=gpurun_out/${1:-r06j}; mkdir -p $O
run() { python bench.py --direction decompress --steps 6 --warmup 2 --no-cpu-baseline --no-secondary "$@" 2>/dev/null | tail -n 1 | python -c "
import sys,json,os; d=json.loads(sys.stdin.read()); print('$*:', d['value'], 'GB/s ms/step', d['ms_per_step'], 'kernel GB/s', d['roofline']['achieved'], 'ok', d.get('bytes_verified'))" | tee -a $O/bench.txt; }
for w in terasort-10g-200p-zstd tpcds-wide-100g-200p-zstd; do
  run --workload $w --maps-per-gpu 10 --task-threads 1
  run --workload $w --maps-per-gpu 10 --task-threads 2
  run --workload $w --maps-per-gpu 20 --task-threads 2
  run --workload $w --maps-per-gpu 20 --task-threads 4
  run --workload $w --maps-per-gpu 16 --task-threads 2
done
run --workload terasort-100g-2000p-zstd --maps-per-gpu 1 --task-threads 1
run --workload terasort-100g-2000p-zstd --maps-per-gpu 2 --task-threads 2
run --workload terasort-100g-2000p-zstd --maps-per-gpu 4 --task-threads 2
