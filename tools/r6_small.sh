#!/bin/bash
# (GPU) round 6: small batched calls — wall clock against the library's stage events, then the sweep's small points
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/${1:-r06f}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_compress.py tests/test_gpu_batch.py tests/test_gpu_decompress.py tests/test_gpu_snappy.py tests/test_gpu_host_batch.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -3 | tee $O/pytest.txt
python tools/small_blocks_probe.py 8 8 30 2>&1 | grep -v "Warn\|amdgpu.ids" | tee $O/probe.txt
python tools/small_blocks_probe.py 8 32 20 2>&1 | grep -v "Warn\|amdgpu.ids" | tee -a $O/probe.txt
python tools/small_blocks_probe.py 1 8 30 2>&1 | grep -v "Warn\|amdgpu.ids" | tee -a $O/probe.txt
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -n 1 | cut -c1-1500 | tee $O/headline.txt
