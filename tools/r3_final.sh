#!/bin/bash
# End-of-round evidence: the GPU test suite, the driver's bench command, and the re-profile of what changed
# (usage: tools/r3_final.sh <tag> "<profile sets>"; writes gpurun_out/<tag>/, tools/r3_report.py turns it into profiles/).
tag=${1:-r03r}; sets=${2:-decompress}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_full.json 2> $O/bench_full.err; tail -c 600 $O/bench_full.json
bash tools/r3_base.sh $tag "$sets" 2>&1 | tail -12
