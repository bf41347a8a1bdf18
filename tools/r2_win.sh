#!/bin/bash
# decoder window geometry A/B (exp libs built with -DS3S_BWIN / -DS3S_BHIST)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/spark-s3-shuffle_amd/lib
for e in "" w6k w5k w4k; do
  if [ -n "$e" ]; then export S3S_CODEC_LIB=$L/libs3shuffle_codec_exp_$e.so; else unset S3S_CODEC_LIB; fi
  [ -n "$e" ] && timeout 200 python -m pytest tests/test_gpu_decompress.py tests/test_gpu_hardening.py -x -q 2>&1 | tail -1
  for w in terasort-10g-200p-lz4 tpcds-wide-100g-200p-lz4 tpcds-wide-100g-200p-snappy; do
    timeout 200 python bench.py --no-cpu-baseline --direction decompress --workload $w 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('${e:-default}', '$w', d['value'], 'GB/s codec', d['stages_ms_per_library_call']['codec'])"
  done
done
