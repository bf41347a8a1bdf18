#!/bin/bash
# A/B of Snappy compress builds (default lib vs exp libs), GPU parity tests first
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/snab
timeout 600 python -m pytest tests/test_gpu_snappy.py tests/test_gpu_batch.py -x -q 2>&1 | tail -3 | tee gpurun_out/snab/pytest.txt
: > gpurun_out/snab/ab.txt
for n in "$@" default; do
  L=""; [ "$n" != default ] && L=$R/spark-s3-shuffle_amd/lib/libs3shuffle_codec_exp_$n.so
  echo "== $n" | tee -a gpurun_out/snab/ab.txt
  S3S_CODEC_LIB=$L timeout 300 python tools/snappy_bench.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/snab/ab.txt
  S3S_CODEC_LIB=$L timeout 300 python bench.py --no-cpu-baseline --workload tpcds-wide-100g-200p-snappy 2>&1 | grep '^{' | tee gpurun_out/snab/bench_snappy_$n.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('snappy wide rows', d['value'], 'GB/s')" | tee -a gpurun_out/snab/ab.txt
done
