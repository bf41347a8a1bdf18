#!/bin/bash
# persistent-grid experiment (exp build: make -C spark-s3-shuffle_amd/csrc exp EXPNAME=persist EXPFLAGS=-DS3S_X_PERSIST):
# the LZ4 compress kernel as a persistent grid with static round-robin of the blocks; headline by grid size, the
# shipped library in between (profiles/r03_experiments.md §8)
R=$GRAFT_REPO_ROOT; cd $R
one() { timeout 120 python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['roofline']['avg_launch_ms'])"; }
for rep in 1 2; do
  unset S3S_CODEC_LIB; one default
  export S3S_CODEC_LIB=$R/spark-s3-shuffle_amd/lib/libs3shuffle_codec_exp_persist.so
  for g in 640 1280 1920 2560 3840; do S3S_X_PERSIST_GRID=$g one persist_$g; done
done
unset S3S_CODEC_LIB; one default
