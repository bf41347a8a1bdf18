#!/bin/bash
# (GPU) round 6: LZ4 window block with the software-pipelined run loop (S3S_ENGINE_EARLY, default build) against -DS3S_ENGINE_NO_EARLY
cd $GRAFT_REPO_ROOT; O=gpurun_out/${1:-r06v}; mkdir -p $O
run() { python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-secondary "$@" 2>/dev/null | tail -n 1 | python -c "
import sys,json,os; d=json.loads(sys.stdin.read()); print(os.environ.get('LIBTAG'), '$*:', d['value'], 'GB/s codec ms', d['stages_ms_per_library_call']['codec'], 'image_verified', d.get('image_verified'))" | tee -a $O/bench.txt; }
for rep in 1 2 3; do
for lib in early noearly; do
  export LIBTAG=$lib
  if [ $lib = early ]; then unset S3S_CODEC_LIB; else export S3S_CODEC_LIB=$GRAFT_REPO_ROOT/spark-s3-shuffle_amd/lib/libs3shuffle_codec_exp_$lib.so; fi
  run
  run --workload tpcds-wide-100g-200p-lz4 --maps-per-gpu 4
  run --workload skew-1part-lz4 --map-mib 128 --maps-per-gpu 8
done; done
