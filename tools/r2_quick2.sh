#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
V=${1:-10}
O=$R/gpurun_out/r2q
mkdir -p $O
cd $R
timeout 100 python tools/lz4_timing.py terasort 67108864 $V 2>&1 | tail -20
timeout 100 python tools/lz4_timing.py tpcds 67108864 $V 2>&1 | tail -20
cd /tmp
timeout 200 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS -d $O/pmc_v$V -o p -- python $R/tools/lz4_dense_bench.py 134217728 $V > $O/pmc_v$V.log 2>&1
cd $R
python - <<PY
import sqlite3, glob, collections
for db in glob.glob("$O/pmc_v$V/**/*.db", recursive=True):
    con = sqlite3.connect(db)
    try:
        tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
        pmc = [t for t in tabs if t.startswith('rocpd_pmc_event')][0]
        info = [t for t in tabs if t.startswith('rocpd_info_pmc')][0]
        disp = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
        sym = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
        q = f"select s.kernel_name, i.name, sum(e.value), count(distinct d.id) from {pmc} e join {info} i on e.pmc_id=i.id join {disp} d on e.event_id=d.event_id join {sym} s on d.kernel_id=s.id group by 1,2"
        acc = collections.defaultdict(dict)
        for k, n, v, c in con.execute(q):
            if 'lz4_compress' in k: acc[k[:60]][n] = (v / c, c)
        for k, d in acc.items():
            print(k, {n: round(v[0]) for n, v in d.items()}, 'launches', list(d.values())[0][1])
    except Exception as ex:
        print('pmc parse failed', ex, tabs[:10])
PY
