#!/bin/bash
# (GPU) round 5: A/B of the Snappy window block with deferred emission (default) against the library built before it (exp "nodefer")
#   gpurun --timeout 900 -- 'bash tools/r5_defer_snappy_ab.sh r05o'
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
tag=${1:-r05o}
O=gpurun_out/$tag; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_snappy.py -x -q 2>&1 | tail -3 | tee $O/pytest_gpu_snappy.txt
: > $O/ab.txt
for n in nodefer default nodefer default; do
  if [ "$n" = default ]; then unset S3S_CODEC_LIB; else export S3S_CODEC_LIB=$R/spark-s3-shuffle_amd/lib/libs3shuffle_codec_exp_$n.so; fi
  timeout 200 python bench.py --no-cpu-baseline --no-secondary --verify --workload tpcds-wide-100g-200p-snappy 2>$O/err_$n.txt | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tpcds-wide-100g-200p-snappy', '$n', d['value'], 'GB/s', d.get('stages_ms_per_library_call'))" | tee -a $O/ab.txt
  grep -h verify $O/err_$n.txt | tee -a $O/ab.txt
  timeout 200 python tools/snappy_bench.py 2>&1 | grep -v amdgpu.ids | tail -4 | tee -a $O/ab.txt
done
