#!/bin/bash
# (GPU) round 6: reduce side by task threads / map tasks per call, new lds_store16 + round-loop head; parity first
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_decompress.py tests/test_gpu_hardening.py tests/test_gpu_batch.py tests/test_gpu_snappy.py tests/test_gpu_lzf.py tests/test_gpu_zstd.py -x -q 2>&1 | tail -3 | tee $O/pytest.txt
for t in 1 2 4; do
  for b in -1 1 2; do
    python bench.py --direction decompress --maps-per-gpu 8 --task-threads $t --batch $b --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -n 1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('threads $t batch $b:', d['value'], 'GB/s ms/step', d['ms_per_step'], 'stages', d['stages_ms_per_library_call'], 'verified', d.get('bytes_verified'))" | tee -a $O/threads.txt
  done
done
python bench.py --direction decompress --workload tpcds-wide-100g-200p-snappy --maps-per-gpu 8 --task-threads 2 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -n 1 | cut -c1-400 | tee -a $O/threads.txt
