#!/bin/bash
# quick GPU check of LZ4 compress variants: parity (variant list in $1) + single-stream bench + phase timing
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
V=${1:-10}
O=$R/gpurun_out/r2q
mkdir -p $O
cd $R
S3S_TEST_LZ4_VARIANTS=$V timeout 600 python -m pytest tests/test_gpu_compress.py -x -q 2>&1 | tail -3
timeout 300 python tools/lz4_dense_bench.py 134217728 2,$V 2>&1 | grep -v amdgpu.ids
if [ -f spark-s3-shuffle_amd/lib/libs3shuffle_codec_dbg.so ]; then
timeout 100 python tools/lz4_timing.py terasort 67108864 $V 2>&1 | tail -2
timeout 100 python tools/lz4_timing.py tpcds 67108864 $V 2>&1 | tail -2
fi
