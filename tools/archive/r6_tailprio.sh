#!/bin/bash
# (GPU) round 6: tail kernels on a high-priority stream (S3S_TAIL_PRIO=1) against the same stream as the codec kernel
cd $GRAFT_REPO_ROOT; O=gpurun_out/${1:-r06h}; mkdir -p $O
run() { python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-secondary "$@" 2>/dev/null | tail -n 1 | python -c "
import sys,json,os; d=json.loads(sys.stdin.read()); print('tail_prio=%s $*:' % os.environ.get('S3S_TAIL_PRIO','0'), d['value'], 'GB/s ms/step', d['ms_per_step'], 'stages', d['stages_ms_per_library_call'], 'ok', d.get('image_verified'), d.get('bytes_verified'))" | tee -a $O/bench.txt; }
for p in 0 1 0 1; do
  export S3S_TAIL_PRIO=$p
  run
  run --workload skew-1part-lz4 --map-mib 8 --maps-per-gpu 8 --task-threads 2
  run --workload skew-1part-lz4 --map-mib 8 --maps-per-gpu 32 --task-threads 2
  run --workload terasort-100g-2000p-lz4-crc32 --maps-per-gpu 4
done
unset S3S_TAIL_PRIO
run --workload skew-1part-lz4 --map-mib 8 --maps-per-gpu 8 --direction decompress --task-threads 1
run --workload skew-1part-lz4 --map-mib 8 --maps-per-gpu 8 --direction decompress --task-threads 2
run --workload skew-1part-lz4 --map-mib 8 --maps-per-gpu 32 --direction decompress --task-threads 2
