#!/bin/bash
tag=${1:-r04g}
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
O=gpurun_out/$tag; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_zstd.py tests/test_gpu_host_batch.py -x -q 2>&1 | tail -4 | tee $O/pytest_zstd.txt
head1() { timeout 120 python bench.py --no-cpu-baseline --no-secondary --workload terasort-10g-200p-zstd --direction decompress --steps 5 --warmup 2 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline'].get('avg_launch_ms'), d['stages_ms_per_library_call'])" || echo FAILED; }
{
echo "single pass (guess 8)   $(head1)"
echo "two passes              $(S3S_ZSTD_GUESS=0 head1)"
echo "single pass (guess 8)   $(head1)"
echo "single pass, 4 tasks    $(head1 --maps-per-gpu 4)"
echo "two passes, 4 tasks     $(S3S_ZSTD_GUESS=0 head1 --maps-per-gpu 4)"
} 2>&1 | tee $O/zstd_single_pass.txt
