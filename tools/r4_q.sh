#!/bin/bash
mkdir -p gpurun_out/$1
timeout 900 python -m pytest tests/test_gpu_compress.py tests/test_gpu_batch.py tests/test_gpu_host_batch.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/$1/pytest.txt 2>&1
tail -5 gpurun_out/$1/pytest.txt
for i in 1 2; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline', d['value'], d['ms_per_step'])"; done
