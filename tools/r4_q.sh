#!/bin/bash
mkdir -p gpurun_out/$1
python -m pytest tests/test_gpu_zstd.py tests/test_gpu_hardening.py tests/test_gpu_host_batch.py -x -q -m gpu > gpurun_out/$1/pytest.txt 2>&1
tail -3 gpurun_out/$1/pytest.txt
