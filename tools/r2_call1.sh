#!/bin/bash
# round 2, call 1: new lean window parse (variant 10): parity + single-stream comparison
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2c1
mkdir -p $O
cd $R
S3S_TEST_LZ4_VARIANTS=10 timeout 600 python -m pytest tests/test_gpu_compress.py -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
timeout 300 python tools/lz4_dense_bench.py 134217728 1,2,10 > $O/dense.log 2>&1
tail -5 $O/pytest.log; cat $O/dense.log
