#!/bin/bash
# (GPU) round 6: the batch decoder's LDS output window again, after the instruction diet: window bytes / history kept
cd $GRAFT_REPO_ROOT; O=gpurun_out/${1:-r06p}; mkdir -p $O
run() { python bench.py --direction decompress --steps 10 --warmup 3 --no-cpu-baseline --no-secondary "$@" 2>/dev/null | tail -n 1 | python -c "
import sys,json,os; d=json.loads(sys.stdin.read()); print(os.environ.get('LIBTAG'), '$*:', d['value'], 'GB/s codec ms', d['stages_ms_per_library_call']['codec'], 'verified', d.get('bytes_verified'))" | tee -a $O/bench.txt; }
for lib in default bw7616_4096 bw6592_4096 bw4544_3072 bw4544_2048 bw3520_2048 default; do
  export LIBTAG=$lib
  if [ $lib = default ]; then unset S3S_CODEC_LIB; else export S3S_CODEC_LIB=$GRAFT_REPO_ROOT/spark-s3-shuffle_amd/lib/libs3shuffle_codec_exp_$lib.so; fi
  run --maps-per-gpu 8
  run --workload tpcds-wide-100g-200p-lz4 --maps-per-gpu 8
  run --workload tpcds-wide-100g-200p-snappy --maps-per-gpu 8
  run --workload skew-1part-lz4 --map-mib 1024 --maps-per-gpu 2
done
