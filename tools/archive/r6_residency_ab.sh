#!/bin/bash
# (GPU) round 6: wavefronts per CU the batch decoder may hold (dynamic LDS pad; 32 = every slot of the CU)
cd $GRAFT_REPO_ROOT; O=gpurun_out/${1:-r06s}; mkdir -p $O
run() { python bench.py --direction decompress --steps 10 --warmup 3 --no-cpu-baseline --no-secondary "$@" 2>/dev/null | tail -n 1 | python -c "
import sys,json,os; d=json.loads(sys.stdin.read()); print('waves/CU', os.environ.get('S3S_DEC_WAVES_PER_CU'), '$*:', d['value'], 'GB/s')" | tee -a $O/bench.txt; }
for w in 32 30 28 26 32 30 28; do
  export S3S_DEC_WAVES_PER_CU=$w
  run --maps-per-gpu 8
  run --workload tpcds-wide-100g-200p-snappy --maps-per-gpu 8
  run --workload skew-1part-lz4 --map-mib 8 --maps-per-gpu 32
  run --workload skew-1part-lz4 --map-mib 8 --maps-per-gpu 8 --task-threads 1
  python bench.py --host-path-only 2>/dev/null | tail -n 1 | python -c "
import sys,json,os; d=json.loads(sys.stdin.read())['host_path']; print('waves/CU', os.environ.get('S3S_DEC_WAVES_PER_CU'), 'host path compress', d['compress_by_task_threads'], 'decompress', d['verify_decompress_by_task_threads'])" | tee -a $O/bench.txt
done
