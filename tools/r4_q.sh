#!/bin/bash
mkdir -p gpurun_out/$1
timeout 600 python -m pytest tests/test_gpu_zstd.py tests/test_gpu_hardening.py -x -q -m gpu > gpurun_out/$1/pytest.txt 2>&1
tail -2 gpurun_out/$1/pytest.txt
for w in terasort-10g-200p-zstd tpcds-wide-100g-200p-zstd; do
for i in 1 2; do python bench.py --workload $w --direction decompress --maps-per-gpu 8 --steps 8 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', d['value'], d['ms_per_step'])"; done
done
