#!/bin/bash
mkdir -p gpurun_out/$1
for v in shipped $2; do
  if [ $v = shipped ]; then unset S3S_CODEC_LIB; else export S3S_CODEC_LIB=$PWD/spark-s3-shuffle_amd/lib/libs3shuffle_codec_exp_$v.so; fi
  for i in 1 2; do
  python bench.py --workload terasort-10g-200p-zstd --direction decompress --maps-per-gpu 8 --steps 8 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/$1/zstd_$v.json 2> gpurun_out/$1/zstd_$v.err
  python - <<P
import json
d=json.loads(open("gpurun_out/$1/zstd_$v.json").read().strip().splitlines()[-1])
print("$v", d["value"], d["ms_per_step"])
P
  done
done
