#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp; mkdir -p gpurun_out/r04t
timeout 200 python -m pytest tests/test_gpu_zstd.py -x -q 2>&1 | tail -2
h() { timeout 120 python bench.py --no-cpu-baseline --no-secondary --workload terasort-10g-200p-zstd --direction decompress --steps 5 --warmup 2 "$@" 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline'].get('avg_launch_ms'))" || echo FAILED; }
{
echo "pipelined Huffman windows   $(h)"
echo "pipelined Huffman windows   $(h)"
} 2>&1 | tee gpurun_out/r04t/zstd.txt
