#!/bin/bash
# (GPU) round 6: decoder parity suite + reduce-side throughput of the current build (two task threads = bench.py's default)
cd $GRAFT_REPO_ROOT; tag=${1:-r06c}; O=gpurun_out/$tag; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_decompress.py tests/test_gpu_hardening.py tests/test_gpu_batch.py tests/test_gpu_snappy.py tests/test_gpu_lzf.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -3 | tee $O/pytest.txt
run() { python bench.py --direction decompress --steps 10 --warmup 3 --no-cpu-baseline --no-secondary "$@" 2>/dev/null | tail -n 1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$*:', d['value'], 'GB/s ms/step', d['ms_per_step'], 'stages', d['stages_ms_per_library_call'], 'verified', d.get('bytes_verified'))" | tee -a $O/bench.txt; }
run --maps-per-gpu 8
run --workload tpcds-wide-100g-200p-lz4 --maps-per-gpu 8
run --workload tpcds-wide-100g-200p-snappy --maps-per-gpu 8
run --workload terasort-10g-200p-lzf --maps-per-gpu 8
run --workload tpcds-wide-100g-200p-snappy --maps-per-gpu 8 --task-threads 1
