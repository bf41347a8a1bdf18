#!/bin/bash
# round 2: LZ4 decode variants — parity + bench (usage: r2_dec.sh <variant>)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
V=${1:-4}
cd $R
S3S_TEST_LZ4_DECODE_VARIANTS=$V timeout 300 python -m pytest tests/test_gpu_decompress.py -x -q 2>&1 | tail -5
for v in 3 $V; do
for w in terasort-10g-200p-lz4 tpcds-wide-100g-200p-lz4; do
timeout 200 python bench.py --no-cpu-baseline --direction decompress --workload $w --lz4-decode-variant $v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w variant $v', d['value'], 'GB/s', d['ms_per_step'], 'ms/step', d['stages_ms_per_library_call'])"
done; done
