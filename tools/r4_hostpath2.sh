#!/bin/bash
# (GPU) the host path measured where the driver measures it: at the end of the full secondary pass of bench.py
tag=${1:-r04d}
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
O=gpurun_out/$tag; mkdir -p $O
hp() { timeout 200 python bench.py --no-cpu-baseline --secondary --steps 5 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['secondary']['host_path']; print(d['value'], h['compress_by_task_threads'], h['verify_decompress_by_task_threads'], h['round_trip_bit_exact'], d['secondary']['wall_s_total'])" || echo FAILED; }
{
echo "own streams             $(S3S_HB_SHARED_COPY=0 hp)"
echo "shared lanes, prio      $(S3S_HB_SHARED_COPY=1 hp)"
echo "shared lanes, no prio   $(S3S_HB_SHARED_COPY=1 S3S_HB_LANE_PRIO=0 hp)"
echo "shared, prio, 32 MiB    $(S3S_HB_SHARED_COPY=1 S3S_HB_GROUP_MIB=32 hp)"
echo "own streams             $(S3S_HB_SHARED_COPY=0 hp)"
echo "shared lanes, prio      $(S3S_HB_SHARED_COPY=1 hp)"
} 2>&1 | tee $O/hostpath_full.txt
