#!/bin/bash
# (GPU) round 5, second half: A/B of the LZ4 window block with deferred emission (default) against -DS3S_ENGINE_NO_DEFER
#   make -C spark-s3-shuffle_amd/csrc exp EXPNAME=nodefer EXPFLAGS=-DS3S_ENGINE_NO_DEFER
#   gpurun --timeout 900 -- 'bash tools/r5_defer_ab.sh r05n'
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
tag=${1:-r05n}
O=gpurun_out/$tag; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_compress.py -x -q 2>&1 | tail -3 | tee $O/pytest_gpu_compress.txt
AB_QUICK=1 bash tools/ab.sh $tag nodefer default
mv $O/ab.txt $O/ab_1.txt
bash tools/ab.sh $tag nodefer default nodefer default
for w in terasort-100g-2000p-lz4-crc32 skew-1part-lz4; do
  for n in nodefer default; do
    if [ "$n" = default ]; then unset S3S_CODEC_LIB; else export S3S_CODEC_LIB=$R/spark-s3-shuffle_amd/lib/libs3shuffle_codec_exp_$n.so; fi
    timeout 200 python bench.py --no-cpu-baseline --no-secondary --verify --workload $w 2>$O/err_${w}_$n.txt | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', '$n', d['value'], 'GB/s')" | tee -a $O/ab.txt
    grep -h verify $O/err_${w}_$n.txt | tee -a $O/ab.txt
  done
done
