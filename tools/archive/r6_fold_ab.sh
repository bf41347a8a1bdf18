#!/bin/bash
# (GPU) round 6: frame checks folded into the decode launch (default) against the separate launch (S3S_LZ4_VERIFY_FOLD=0)
cd $GRAFT_REPO_ROOT; tag=${1:-r06d}; O=gpurun_out/$tag; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_decompress.py tests/test_gpu_hardening.py tests/test_gpu_batch.py tests/test_gpu_fullsize.py tests/test_gpu_host_batch.py -x -q 2>&1 | tail -3 | tee $O/pytest.txt
run() { python bench.py --direction decompress --steps 10 --warmup 3 --no-cpu-baseline --no-secondary "$@" 2>/dev/null | tail -n 1 | python -c "
import sys,json,os; d=json.loads(sys.stdin.read()); print('fold=%s $*:' % os.environ.get('S3S_LZ4_VERIFY_FOLD','1'), d['value'], 'GB/s ms/step', d['ms_per_step'], 'stages', d['stages_ms_per_library_call'], 'verified', d.get('bytes_verified'))" | tee -a $O/bench.txt; }
for fold in 1 0 1 0; do
  export S3S_LZ4_VERIFY_FOLD=$fold
  run --maps-per-gpu 8 --task-threads 1
  run --maps-per-gpu 8 --task-threads 2
  run --workload skew-1part-lz4 --map-mib 1024 --maps-per-gpu 1 --task-threads 1
done
