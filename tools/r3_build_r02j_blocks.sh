#!/bin/bash
# Builds ../lib/libs3shuffle_codec_exp_r02j.so: today's sources with the window blocks as they were at the last GPU
# measurement of round 2 (commit a16bbe8, profiles/r02j_*), for the A/B the CPU-only trims of r02k still need:
#   tools/r3_build_r02j_blocks.sh                      (here, no GPU)
#   gpurun -- 'bash tools/r2_eng.sh r02j; bash tools/r2_sn_ab.sh r02j'      (GPU tests of the default lib first, then both libs timed)
set -e
cd "$(dirname "$0")/.."
C=spark-s3-shuffle_amd/csrc
BASE=${1:-a16bbe8}
for f in lz4_window_engine.inc snappy_window_engine.inc; do
  cp $C/$f /tmp/$f.now
  git show $BASE:$C/$f > $C/$f
done
trap 'for f in lz4_window_engine.inc snappy_window_engine.inc; do cp /tmp/$f.now '$C'/$f; done' EXIT
make -C $C exp EXPNAME=r02j EXPFLAGS=
ls -la spark-s3-shuffle_amd/lib/libs3shuffle_codec_exp_r02j.so
