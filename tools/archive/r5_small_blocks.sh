#!/bin/bash
# Round 5: 8 MiB single-partition blocks (the reference's default buffer, S3ShuffleDispatcher.scala:55): stage times of a call
# and throughput by task threads x blocks per step
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for maps in 8 16 32; do for th in 1 2 4 8; do
  if [ $th -gt $maps ]; then continue; fi
  timeout 120 python bench.py --no-cpu-baseline --no-secondary --workload skew-1part-lz4 --map-mib 8 --maps-per-gpu $maps --task-threads $th --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('compress 8 MiB x $maps, threads $th:', d['value'], 'GB/s  ms/step', d['ms_per_step'], d['stages_ms_per_library_call'])"
done; done
for maps in 8 32; do for th in 1 2; do
  timeout 120 python bench.py --no-cpu-baseline --no-secondary --workload skew-1part-lz4 --direction decompress --map-mib 8 --maps-per-gpu $maps --task-threads $th --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('decompress 8 MiB x $maps, threads $th:', d['value'], 'GB/s  ms/step', d['ms_per_step'], d['stages_ms_per_library_call'])"
done; done
