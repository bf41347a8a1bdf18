#!/bin/bash
# A/B: hand-written run loop (exp lib built with -DS3S_ASM_RUN_LOOP) vs the compiled one
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
E=$R/spark-s3-shuffle_amd/lib/libs3shuffle_codec_exp_asm.so
S3S_CODEC_LIB=$E S3S_TEST_LZ4_VARIANTS=10 timeout 600 python -m pytest tests/test_gpu_compress.py tests/test_gpu_batch.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -4
echo "== compiled"; timeout 300 python tools/lz4_dense_bench.py 134217728 10 2>&1 | grep -v amdgpu.ids
echo "== asm"; S3S_CODEC_LIB=$E timeout 300 python tools/lz4_dense_bench.py 134217728 10 2>&1 | grep -v amdgpu.ids
