#!/bin/bash
# Runs on the GPU box (via gpurun): GPU tests, bench lines, rocprofv3 kernel stats + PMC passes.
# usage: tools/gpu_round.sh <tag>
tag=${1:-r1}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
timeout 600 python bench.py --verify > $O/bench.json 2> $O/bench.err
timeout 300 python bench.py --no-cpu-baseline --task-threads 2 > $O/bench_t2.json 2>> $O/bench.err
timeout 300 python bench.py --no-cpu-baseline --workload terasort-100g-2000p-lz4-crc32 > $O/bench_2000p.json 2>> $O/bench.err
timeout 300 python bench.py --no-cpu-baseline --workload skew-1part-lz4 --map-mib 1024 --maps-per-gpu 1 > $O/bench_skew1g.json 2>> $O/bench.err
timeout 300 python bench.py --no-cpu-baseline --workload skew-1part-lz4 --map-mib 1024 --maps-per-gpu 1 --direction decompress > $O/bench_skew1g_decompress.json 2>> $O/bench.err
timeout 400 python bench.py --direction decompress > $O/bench_decompress.json 2>> $O/bench.err
timeout 600 python bench.py --workload tpcds-wide-100g-200p-snappy --verify > $O/bench_snappy.json 2>> $O/bench.err
timeout 300 python bench.py --no-cpu-baseline --workload tpcds-wide-100g-200p-snappy --direction decompress > $O/bench_snappy_decompress.json 2>> $O/bench.err
timeout 600 python bench.py --workload tpcds-wide-100g-200p-lz4 --verify > $O/bench_tpcds_lz4.json 2>> $O/bench.err
timeout 300 python bench.py --no-cpu-baseline --lz4-variant 1 > $O/bench_lz4_variant1.json 2>> $O/bench.err
BENCH="python $R/bench.py --no-cpu-baseline"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- $BENCH > $O/trace.log 2>&1
BENCH="python $R/bench.py --no-cpu-baseline --maps-per-gpu 2 --steps 5 --warmup 1"
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o p -- $BENCH > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o p -- $BENCH > $O/pmc_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $O/pmc_sq1 -o p -- $BENCH > $O/pmc_sq1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS TA_BUSY_avr -d $O/pmc_sq2 -o p -- $BENCH > $O/pmc_sq2.log 2>&1
cd $R
python tools/summarize_prof.py $O > $O/summary.txt 2>&1
cat /sys/fs/cgroup/cpu.max > $O/host.txt 2>&1; nproc >> $O/host.txt; python -c "import os;print(len(os.sched_getaffinity(0)))" >> $O/host.txt
cat $O/bench.json; tail -3 $O/pytest_gpu.log; grep -E "lz4_compress|xxh32" $O/summary.txt | head
