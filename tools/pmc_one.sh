#!/bin/bash
# usage: tools/pmc_one.sh <lz4-variant>   — SQ counters + kernel time of one LZ4 compress variant
v=$1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc1_$v; mkdir -p $O; cd /tmp
CMD="python $R/bench.py --no-cpu-baseline --task-threads 1 --maps-per-gpu 2 --steps 3 --warmup 1 --lz4-variant $v"
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS -d $O/sq -o p -- $CMD > $O/sq.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM_WR TA_BUSY_avr SQ_WAVES -d $O/sq2 -o p -- $CMD > $O/sq2.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- $CMD > $O/trace.log 2>&1
cd $R
python - <<PY
import sqlite3,glob
for kind in ("sq","sq2"):
    for f in glob.glob("$O/"+kind+"/**/*.db",recursive=True):
        c=sqlite3.connect(f)
        for name,counter,mean,n in c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%lz4_compress%' group by kernel_name, counter_name"):
            print("v$v",counter,round(mean/4096,1),"per chunk")
for f in glob.glob("$O/trace/**/*.db",recursive=True):
    c=sqlite3.connect(f)
    for name,cnt,avg in c.execute("select name,count(*),avg(duration) from kernels where name like '%lz4_compress%' group by name"):
        print("v$v trace avg_us",round(avg/1e3,1),cnt)
PY
