#!/bin/bash
# (GPU) block-size sweep of SURVEY §8(d) with the round-3 kernels (the tracked sweep is round 2's, profiles/r02i_* / r02e_*):
# single-partition TeraSort blocks of 8 .. 1024 MiB in HBM, compress and verify + decompress.
#   gpurun --timeout 600 -- 'bash tools/r4_sweep.sh r04b'
tag=${1:-r04b}
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
O=gpurun_out/$tag; mkdir -p $O
for dir in compress decompress; do
  : > $O/block_size_sweep_$dir.jsonl
  for mib in 8 32 128 512 1024; do
    timeout 120 python bench.py --workload skew-1part-lz4 --map-mib $mib --direction $dir --no-cpu-baseline --no-secondary \
      --steps 10 --warmup 3 2>/dev/null | grep '^{' >> $O/block_size_sweep_$dir.jsonl
    tail -n 1 $O/block_size_sweep_$dir.jsonl | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$dir', $mib, 'MiB', d['value'], 'GB/s')"
  done
done
