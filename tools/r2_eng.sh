#!/bin/bash
# A/B of LZ4 compress builds: the default lib against exp libs (make -C spark-s3-shuffle_amd/csrc exp EXPNAME=<name> ...)
#   tools/r2_eng.sh <name> [<name> ...]     e.g.  noeng = built with -DS3S_NO_WINDOW_ENGINE (C++ window path)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/eng
timeout 600 python -m pytest tests/test_gpu_compress.py tests/test_gpu_batch.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -4 | tee gpurun_out/eng/pytest.txt
: > gpurun_out/eng/ab.txt
for n in "$@"; do
  E=$R/spark-s3-shuffle_amd/lib/libs3shuffle_codec_exp_$n.so
  echo "== exp $n" | tee -a gpurun_out/eng/ab.txt
  S3S_CODEC_LIB=$E timeout 300 python tools/lz4_dense_bench.py 134217728 10 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/eng/ab.txt
  S3S_CODEC_LIB=$E timeout 300 python bench.py --no-cpu-baseline 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('headline', d['value'], 'GB/s')" | tee -a gpurun_out/eng/ab.txt
done
echo "== default lib" | tee -a gpurun_out/eng/ab.txt
timeout 300 python tools/lz4_dense_bench.py 134217728 10 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/eng/ab.txt
timeout 300 python bench.py --no-cpu-baseline --verify 2>&1 | grep '^{' | tee gpurun_out/eng/bench_headline.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('headline', d['value'], 'GB/s')" | tee -a gpurun_out/eng/ab.txt
timeout 300 python bench.py --no-cpu-baseline --workload tpcds-wide-100g-200p-lz4 2>&1 | grep '^{' | tee gpurun_out/eng/bench_wide.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('wide rows', d['value'], 'GB/s')" | tee -a gpurun_out/eng/ab.txt
