#!/bin/bash
# (GPU) first call of round 4: everything that was built at the end of round 3 without a GPU, measured in ONE call.
#   tools/r4_build_exps.sh   (here, on the CPU box, first — the exp libraries travel with the snapshot)
#   gpurun --timeout 900 -- 'bash tools/r4_first_call.sh r04a'
# A/B rule of profiles/r03_experiments.md: same box, shipped library first and last.
tag=${1:-r04a}
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
O=gpurun_out/$tag; mkdir -p $O
L=$R/spark-s3-shuffle_amd/lib
# the state the round starts from: the whole GPU suite (incl. the tests added without a GPU at the end of round 3) and the driver's bench line
timeout 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest_gpu.txt
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | grep '^{' > $O/bench_full.json
python -c "import json; d=json.loads(open('$O/bench_full.json').read()); print('driver-style headline', d['value'], 'GB/s;', {k: v.get('value') for k, v in d.get('secondary', {}).items() if isinstance(v, dict) and 'value' in v})" | tee $O/bench_full.txt
head1() { timeout 90 python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline'].get('avg_launch_ms'))" || echo FAILED; }
{
echo "== compress headline (GB/s, ms per launch): shipped / rollpf / storent / setprio / grid 1536 / shipped"
unset S3S_CODEC_LIB;                                   echo "shipped  $(head1)"
export S3S_CODEC_LIB=$L/libs3shuffle_codec_exp_rollpf.so;  echo "rollpf   $(head1 --verify)"
export S3S_CODEC_LIB=$L/libs3shuffle_codec_exp_storent.so; echo "storent  $(head1 --verify)"
export S3S_CODEC_LIB=$L/libs3shuffle_codec_exp_setprio.so; echo "setprio  $(head1 --verify)"
export S3S_CODEC_LIB=$L/libs3shuffle_codec_exp_rollpf6k.so; echo "rollpf6k $(head1 --verify)"
export S3S_CODEC_LIB=$L/libs3shuffle_codec_exp_rollpfprio.so; echo "rollpf+prio $(head1 --verify)"
unset S3S_CODEC_LIB; echo "grid1536 $(S3S_LZ4_GRID=1536 head1)"
echo "shipped  $(head1)"
echo "== wide rows LZ4: shipped / rollpf"
echo "shipped  $(head1 --workload tpcds-wide-100g-200p-lz4)"
export S3S_CODEC_LIB=$L/libs3shuffle_codec_exp_rollpf.so; echo "rollpf   $(head1 --workload tpcds-wide-100g-200p-lz4)"; unset S3S_CODEC_LIB
echo "== Snappy wide rows: shipped / setprio"
echo "shipped  $(head1 --workload tpcds-wide-100g-200p-snappy)"
export S3S_CODEC_LIB=$L/libs3shuffle_codec_exp_setprio.so; echo "setprio  $(head1 --workload tpcds-wide-100g-200p-snappy --verify)"; unset S3S_CODEC_LIB
echo "== decoder: shipped / persistent grid (26, 20, 16 wavefronts per CU) / shipped"
LIBS="decpers" bash tools/r3_dec_quick.sh
for g in 5120 4096; do
  export S3S_CODEC_LIB=$L/libs3shuffle_codec_exp_decpers.so S3S_DEC_GRID=$g
  echo "decpers grid=$g $(head1 --direction decompress --steps 10 --warmup 3)"
done
unset S3S_DEC_GRID
export S3S_CODEC_LIB=$L/libs3shuffle_codec_exp_finwide.so; echo "finwide  $(head1 --direction decompress --steps 10 --warmup 3)"
unset S3S_CODEC_LIB
echo "shipped  $(head1 --direction decompress --steps 10 --warmup 3)"
echo "== zstd decode: shipped / one bit window per sequence / shipped"
echo "shipped  $(head1 --workload terasort-10g-200p-zstd --direction decompress --steps 5 --warmup 2)"
export S3S_CODEC_LIB=$L/libs3shuffle_codec_exp_zsfast.so
timeout 120 python -m pytest tests/test_gpu_zstd.py -x -q 2>&1 | tail -1
echo "zsfast   $(head1 --workload terasort-10g-200p-zstd --direction decompress --steps 5 --warmup 2)"
unset S3S_CODEC_LIB
echo "shipped  $(head1 --workload terasort-10g-200p-zstd --direction decompress --steps 5 --warmup 2)"
} 2>&1 | tee $O/first_call.txt
