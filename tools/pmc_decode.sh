#!/bin/bash
# usage: tools/pmc_decode.sh "<variants>" [workload]  — SQ counters + kernel time of the LZ4 decode kernels (per 32 KiB frame)
export TMPDIR=/tmp
VS=${1:-4}; W=${2:-terasort-10g-200p-lz4}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmcdec; mkdir -p $O; cd /tmp
for v in $VS; do
CMD="python $R/bench.py --no-cpu-baseline --direction decompress --workload $W --task-threads 1 --maps-per-gpu 2 --steps 3 --warmup 1 --lz4-decode-variant $v"
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS -d $O/v${v}_sq -o p -- $CMD > $O/v${v}_sq.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_WAVES -d $O/v${v}_sq2 -o p -- $CMD > $O/v${v}_sq2.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d $O/v${v}_trace -o t -- $CMD > $O/v${v}_trace.log 2>&1
done
cd $R
for v in $VS; do
python - <<PY
import sqlite3,glob
for kind in ("sq","sq2"):
    for f in glob.glob("$O/v${v}_"+kind+"/**/*.db",recursive=True):
        c=sqlite3.connect(f)
        for name,counter,mean,n in c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%lz4_decompress%' group by kernel_name, counter_name"):
            print("v$v $W",name.split("(")[0][-30:],counter,round(mean/4096,1),"per frame")
for f in glob.glob("$O/v${v}_trace/**/*.db",recursive=True):
    c=sqlite3.connect(f)
    for name,cnt,avg in c.execute("select name,count(*),avg(duration) from kernels where name like '%lz4_decompress%' group by name"):
        print("v$v $W trace avg_us",name.split("(")[0][-30:],round(avg/1e3,1),cnt)
PY
done
