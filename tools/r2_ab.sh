#!/bin/bash
# A/B experiment libs: usage r2_ab.sh <variant> <exp names...>
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
V=$1; shift
cd $R
echo "== default"; timeout 300 python tools/lz4_dense_bench.py 134217728 $V 2>&1 | grep -v amdgpu.ids
for e in "$@"; do
  echo "== exp $e"
  S3S_CODEC_LIB=$R/spark-s3-shuffle_amd/lib/libs3shuffle_codec_exp_$e.so timeout 300 python tools/lz4_dense_bench.py 134217728 $V 2>&1 | grep -v amdgpu.ids
done
