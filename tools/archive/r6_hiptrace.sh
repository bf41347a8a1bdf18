#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/${1:-r06g}; mkdir -p $O
cd /tmp; rocprofv3 --hip-trace --kernel-trace --output-format csv -d /tmp/ht -o t -- python $GRAFT_REPO_ROOT/tools/small_blocks_probe.py 8 32 8 > $GRAFT_REPO_ROOT/$O/trace.log 2>&1
cd $GRAFT_REPO_ROOT; python - <<'PY' | tee $O/api_sequence.txt
import csv
rows=list(csv.DictReader(open("/tmp/ht/t_hip_api_trace.csv")))
print(rows[0].keys())
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# find the compress phase: the last 9 hipStreamSynchronize before the decompress phase... print the API sequence of the 5th..6th sync window
prev_end=int(rows[0]["Start_Timestamp"]); T0=prev_end
for i,r in enumerate(rows):
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    if r["Function"].startswith("__hipRegister"): prev_end=e; continue
    if e-s>150000 or s-prev_end>150000:
        print("%6d t=%10.1f us gap_before %8.1f us dur %9.1f us %s"%(i,(s-T0)/1e3,(s-prev_end)/1e3,(e-s)/1e3,r["Function"]))
    prev_end=e
PY
