#!/bin/bash
# decoder with the compressed frame prefetched at frame start (ships) against the exp build without it
# (make -C spark-s3-shuffle_amd/csrc exp EXPNAME=nopf EXPFLAGS=-DS3S_DEC_NO_PREFETCH), then the GPU suite and the driver's line
tag=${1:-r03w}
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/$tag; mkdir -p $O
one() { timeout 40 python bench.py --no-cpu-baseline --no-secondary --direction decompress --steps 10 --warmup 3 "${@:2}" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['roofline']['avg_launch_ms'])" || echo "$1 FAILED"; }
unset S3S_CODEC_LIB; one prefetch_terasort | tee -a $O/ab.txt
export S3S_CODEC_LIB=$R/spark-s3-shuffle_amd/lib/libs3shuffle_codec_exp_nopf.so; one noprefetch_terasort | tee -a $O/ab.txt
unset S3S_CODEC_LIB; one prefetch_terasort | tee -a $O/ab.txt
one prefetch_snappy_wide --workload tpcds-wide-100g-200p-snappy | tee -a $O/ab.txt
timeout 100 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; tail -1 $O/pytest_gpu.txt
timeout 70 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_full.json 2> $O/bench_full.err; python -c "import json; d=json.loads([l for l in open('$O/bench_full.json') if l.startswith('{')][-1]); print('driver-style', d['value'], {k: v.get('value') for k, v in d['secondary'].items() if isinstance(v, dict)})"
