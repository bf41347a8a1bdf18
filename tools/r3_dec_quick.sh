# decoder A/B: the shipped library and the exp builds named in $LIBS, decompress direction of four workloads
set -u
timeout 300 python -m pytest tests/test_gpu_decompress.py tests/test_gpu_snappy.py tests/test_gpu_hardening.py tests/test_gpu_batch.py -x -q 2>&1 | tail -3
for lib in "" ${LIBS:-}; do
  if [ -n "$lib" ]; then export S3S_CODEC_LIB=$PWD/spark-s3-shuffle_amd/lib/libs3shuffle_codec_exp_$lib.so; else unset S3S_CODEC_LIB; fi
  for w in terasort-10g-200p-lz4 tpcds-wide-100g-200p-snappy tpcds-wide-100g-200p-lz4 skew-1part-lz4; do
    timeout 200 python bench.py --workload $w --direction decompress --no-cpu-baseline --no-secondary --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=${lib:-shipped}', '$w', d['value'], d['roofline'].get('avg_launch_ms'))"
  done
done
unset S3S_CODEC_LIB
