#!/bin/bash
# last GPU call of round 3: re-profile the compress side as it ships (persistent grid), then the grid size when calls overlap
tag=${1:-r03u}
R=$GRAFT_REPO_ROOT; cd $R
bash tools/r3_base.sh $tag "compress" 2>&1 | tail -8
one() { timeout 60 python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['roofline']['avg_launch_ms'])" || echo "$1 FAILED"; }
for g in 1024 1536 1792; do S3S_LZ4_GRID=$g one grid_$g | tee -a gpurun_out/$tag/grid.txt; done
one shipped | tee -a gpurun_out/$tag/grid.txt
