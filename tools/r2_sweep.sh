#!/bin/bash
# block-size sweep {8,32,128,512,1024} MiB, single-partition TeraSort blocks resident in HBM, LZ4 + Adler32
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2sweep; mkdir -p $O; : > $O/sweep.jsonl
for spec in "8 32 8" "32 16 4" "128 8 2" "512 2 2" "1024 2 2"; do
  set -- $spec
  for dir in compress decompress; do
    timeout 300 python bench.py --no-cpu-baseline --workload skew-1part-lz4 --map-mib $1 --maps-per-gpu $2 --task-threads $3 --direction $dir --steps 5 --warmup 2 2>/dev/null >> $O/sweep.jsonl
    tail -1 $O/sweep.jsonl | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 MiB x $2, $3 threads, $dir:', d['value'], 'GB/s', d['ms_per_step'], 'ms/step')"
  done
done
