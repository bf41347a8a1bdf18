#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_snappy.py tests/test_gpu_decompress.py -x -q 2>&1 | tail -5
for v in 3 4; do
timeout 200 python bench.py --no-cpu-baseline --direction decompress --workload tpcds-wide-100g-200p-snappy --lz4-decode-variant $v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('snappy decode variant $v', d['value'], 'GB/s', d['ms_per_step'], 'ms/step', d['stages_ms_per_library_call'])"
done
timeout 200 python bench.py --no-cpu-baseline --direction decompress 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lz4 decode default', d['value'], 'GB/s', d['ms_per_step'], 'ms/step', d['stages_ms_per_library_call'])"
