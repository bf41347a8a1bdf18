#!/bin/bash
# A/B of codec-library builds on the GPU (round 3): tools/ab.sh <outdir-tag> <exp-name> [<exp-name> ...]
#   exp libs: make -C spark-s3-shuffle_amd/csrc exp EXPNAME=<name> EXPFLAGS=...; "default" = the shipped library.
# Per library: bench.py --verify (bit-exact vs the oracle + headline), wide rows, and the single-stream kernel times.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
tag=$1; shift
O=gpurun_out/$tag; mkdir -p $O
: > $O/ab.txt
for n in "$@"; do
  if [ "$n" = default ]; then unset S3S_CODEC_LIB; else export S3S_CODEC_LIB=$R/spark-s3-shuffle_amd/lib/libs3shuffle_codec_exp_$n.so; fi
  echo "== $n" | tee -a $O/ab.txt
  timeout 300 python bench.py --no-cpu-baseline ${AB_VERIFY---verify} ${BENCH_ARGS} 2> $O/err_$n.txt | grep '^{' > $O/bench_$n.json
  grep -h "verify" $O/err_$n.txt | tee -a $O/ab.txt
  python -c "import sys,json; d=json.loads(open('$O/bench_$n.json').read()); print('headline', d['value'], 'GB/s  codec ms/launch', d['roofline']['avg_launch_ms'], 'stages', d['stages_ms_per_library_call'])" | tee -a $O/ab.txt
  if [ -z "$AB_QUICK" ]; then
    timeout 300 python bench.py --no-cpu-baseline --workload tpcds-wide-100g-200p-lz4 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('wide rows', d['value'], 'GB/s')" | tee -a $O/ab.txt
    timeout 300 python tools/lz4_dense_bench.py 134217728 10 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
  fi
done
