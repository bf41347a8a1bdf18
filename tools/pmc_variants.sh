#!/bin/bash
# usage: tools/pmc_variants.sh <tag>  — PMC comparison of the LZ4 compress variants (GPU box)
tag=${1:-pv}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O; cd /tmp
rocprofv3 -L > $O/counters.txt 2>&1
grep -oE "\b(TA_[A-Z_a-z0-9]+|TCP_[A-Z_a-z0-9]+|TD_[A-Z_a-z0-9]+)\b" $O/counters.txt | sort -u | head -150 > $O/ta_tcp_names.txt
for v in 1 2 3; do
  CMD="python $R/bench.py --no-cpu-baseline --maps-per-gpu 2 --steps 3 --warmup 1 --lz4-variant $v"
  timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS -d $O/v${v}_sq -o p -- $CMD > $O/v${v}_sq.log 2>&1
  timeout 200 rocprofv3 --pmc TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum -d $O/v${v}_ta -o p -- $CMD > $O/v${v}_ta.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --stats -d $O/v${v}_trace -o t -- $CMD > $O/v${v}_trace.log 2>&1
done
cd $R
for v in 1 2 3; do
python - <<PY
import sqlite3,glob
for kind in ("sq","ta"):
    for f in glob.glob("$O/v${v}_"+kind+"/**/*.db",recursive=True):
        c=sqlite3.connect(f)
        for name,counter,mean,n in c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%lz4_compress%' group by kernel_name, counter_name"):
            print("v$v",kind,counter,round(mean,1),n)
for f in glob.glob("$O/v${v}_trace/**/*.db",recursive=True):
    c=sqlite3.connect(f)
    for name,cnt,avg in c.execute("select name,count(*),avg(duration) from kernels where name like '%lz4_compress%' group by name"):
        print("v$v trace avg_us",round(avg/1e3,1),cnt)
PY
done
tail -3 $O/v1_ta.log
