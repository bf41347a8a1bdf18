#!/bin/bash
# timing ablations of the LZ4 compress kernel (exp libs whose OUTPUT IS WRONG by design): single-stream stage times only
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/abl
: > gpurun_out/abl/abl.txt
for n in "$@"; do
  echo "== $n" | tee -a gpurun_out/abl/abl.txt
  L=""; [ "$n" != default ] && L=$GRAFT_REPO_ROOT/spark-s3-shuffle_amd/lib/libs3shuffle_codec_exp_$n.so
  S3S_CODEC_LIB=$L timeout 300 python tools/lz4_dense_bench.py 134217728 10 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/abl/abl.txt
done
