#!/bin/bash
# Round 5, call 6: the xxHash32 quad kernel with 48 loads in flight per lane, against the per-chunk kernel
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
for m in 1 0 2 1 0 2; do
  echo "== headline, S3S_XXH=$m"
  S3S_XXH=$m timeout 300 python bench.py --no-cpu-baseline --no-secondary --verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('headline', d['value'], 'GB/s', d['stages_ms_per_library_call'])"
done
for m in 1 0 2; do
  echo "== 1 GiB block compress, S3S_XXH=$m"
  S3S_XXH=$m timeout 300 python bench.py --no-cpu-baseline --no-secondary --workload skew-1part-lz4 --map-mib 1024 --maps-per-gpu 1 --steps 6 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'GB/s', d['stages_ms_per_library_call'])"
done
