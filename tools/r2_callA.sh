#!/bin/bash
# round 2, call A: variant 10 as default — parity, headline bench, phase timing, SQ counters, decode timing
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2A
mkdir -p $O
cd $R
S3S_TEST_LZ4_VARIANTS=10 timeout 600 python -m pytest tests/test_gpu_compress.py tests/test_gpu_batch.py tests/test_gpu_fullsize.py -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
timeout 300 python bench.py > $O/bench.json 2> $O/bench.err; cat $O/bench.json
timeout 300 python bench.py --no-cpu-baseline --direction decompress > $O/bench_dec.json 2>> $O/bench.err; cat $O/bench_dec.json
timeout 100 python tools/lz4_timing.py terasort 67108864 10 > $O/timing_terasort.log 2>&1; tail -22 $O/timing_terasort.log
timeout 100 python tools/lz4_timing.py tpcds 67108864 10 > $O/timing_tpcds.log 2>&1; tail -22 $O/timing_tpcds.log
timeout 100 python tools/dec_timing.py > $O/dec_timing.log 2>&1; tail -6 $O/dec_timing.log
bash tools/pmc_one.sh 10 > $O/pmc_v10.log 2>&1; tail -30 $O/pmc_v10.log
