#!/bin/bash
# last GPU call of round 2: compress-side measurement set, then the whole GPU suite + smoke
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
S3S_TAG=r02i bash tools/gpu_round2_compress.sh r02i > gpurun_out/r02i_console.txt 2>&1
tail -25 gpurun_out/r02i_console.txt | head -12
O=gpurun_out/r02i
timeout 300 python -m pytest tests -m gpu -q --ignore=tests/test_gpu_launch.py --ignore=tests/test_gpu_hardening.py > $O/pytest_gpu_a.log 2>&1; echo "rc=$?" >> $O/pytest_gpu_a.log; tail -3 $O/pytest_gpu_a.log
timeout 200 python -m pytest tests/test_gpu_hardening.py tests/test_gpu_launch.py -m gpu -q > $O/pytest_gpu_b.log 2>&1; echo "rc=$?" >> $O/pytest_gpu_b.log; tail -3 $O/pytest_gpu_b.log
timeout 100 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
