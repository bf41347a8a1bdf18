#!/bin/bash
# Round 3 experiment: the LZ4 hash table of a block in GLOBAL memory (exp libs built with -DS3S_X_GTAB): the compiled C++
# window path on a per-wavefront table slot, alone at N wavefronts per CU and next to the shipped LDS kernel with a share of
# the blocks.  Every run is verified against the oracle and cut off after 40 s.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/${1:-r03n}; mkdir -p $O
: > $O/gtab.txt
run() {  # label, env...
  echo "== $1" | tee -a $O/gtab.txt; shift
  env "$@" timeout 40 python bench.py --no-cpu-baseline --verify --maps-per-gpu 2 --task-threads 1 --steps 3 --warmup 1 2>$O/err.txt | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('2 tasks, 1 thread:', d['value'], 'GB/s  codec ms/launch', d['roofline']['avg_launch_ms'])" 2>/dev/null | tee -a $O/gtab.txt
  grep -h "verify\|Error\|rror" $O/err.txt | head -2 | tee -a $O/gtab.txt
}
L=$R/spark-s3-shuffle_amd/lib
run "shipped library (LDS table, hand-written block, 10 waves per CU)" X=1
run "compiled C++ window path, LDS table (10 per CU)" S3S_CODEC_LIB=$L/libs3shuffle_codec_exp_noeng.so
for w in 4 10; do
  run "C++ window path, GLOBAL table (store acknowledged before the next access), alone, $w waves per CU" S3S_CODEC_LIB=$L/libs3shuffle_codec_exp_gtabsafe.so S3S_X_GTAB_WAVES=$w S3S_X_GTAB_PCT=100
done
for w in 4 8 10 16 24; do
  run "C++ window path, GLOBAL table, alone, $w waves per CU" S3S_CODEC_LIB=$L/libs3shuffle_codec_exp_gtab.so S3S_X_GTAB_WAVES=$w S3S_X_GTAB_PCT=100
done
for spec in "2 5" "4 10" "4 15" "6 15" "8 20"; do set -- $spec
  run "shipped LDS kernel + $1 global-table waves per CU taking $2 % of the blocks" S3S_CODEC_LIB=$L/libs3shuffle_codec_exp_gtab.so S3S_X_GTAB_WAVES=$1 S3S_X_GTAB_PCT=$2
done
