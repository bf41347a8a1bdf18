#!/bin/bash
# (GPU) timing ablations of the LZ4 window block (same output): what does one more vector-memory instruction of each kind cost?
tag=${1:-r04f}
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
O=gpurun_out/$tag; mkdir -p $O
L=$R/spark-s3-shuffle_amd/lib
head1() { timeout 90 python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline'].get('avg_launch_ms'))" || echo FAILED; }
{
unset S3S_CODEC_LIB; echo "shipped        $(head1)"
for e in $EXPS; do export S3S_CODEC_LIB=$L/libs3shuffle_codec_exp_$e.so; echo "$e  $(head1 --verify)"; done
unset S3S_CODEC_LIB; echo "shipped        $(head1)"
echo "== wide rows"
echo "shipped        $(head1 --workload tpcds-wide-100g-200p-lz4)"
for e in $EXPS; do export S3S_CODEC_LIB=$L/libs3shuffle_codec_exp_$e.so; echo "$e  $(head1 --workload tpcds-wide-100g-200p-lz4)"; done
} 2>&1 | tee $O/ablations.txt
