#!/bin/bash
# PMC of the two zstd passes (size, decode): instructions and waits per launch
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
P=$R/gpurun_out/${1:-r03m}; mkdir -p $P
CMD="python $R/bench.py --no-cpu-baseline --workload terasort-10g-200p-zstd --direction decompress --steps 1 --warmup 1"
cd /tmp
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS -d $P/pmc_1 -o p -- $CMD > $P/pmc_1.log 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES SQ_WAVES -d $P/pmc_2 -o p -- $CMD > $P/pmc_2.log 2>&1
cd $R
python - <<PY
import sqlite3, glob, collections
for d in ("pmc_1","pmc_2"):
    for f in glob.glob("$P/%s/**/*.db" % d, recursive=True):
        c = sqlite3.connect(f)
        rows = c.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection where kernel_name like '%zstd%' order by dispatch_id").fetchall()
        by = collections.OrderedDict()
        for k, cn, v, did in rows:
            by.setdefault(did, {}).setdefault(cn, 0)
            by[did][cn] += v
        for did, cs in list(by.items())[:4]:
            print(d, "dispatch", did, {k: round(v) for k, v in cs.items()})
PY
