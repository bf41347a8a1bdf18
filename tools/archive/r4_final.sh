#!/bin/bash
# (GPU) round-4 closing call: GPU suite, the driver's bench line, then the PMC profile of exactly these sources
#   gpurun --timeout 1700 -- 'bash tools/r4_final.sh r04z [sets]'      then here: python tools/r4_report.py r04z
tag=${1:-r04z}
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
O=gpurun_out/$tag; mkdir -p $O
timeout 500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest_gpu.txt
timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 2>$O/bench_full.err | grep '^{' > $O/bench_full.json
python - <<PY | tee $O/bench_full.txt
import json
d = json.loads(open("$O/bench_full.json").read())
print("headline", d["value"], "GB/s; cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["kind"], "speedup", d.get("speedup_vs_cpu_all_cores"), "traffic", d["roofline"].get("traffic"), d["roofline"].get("traffic_over_algorithmic"))
for k, v in d.get("secondary", {}).items():
    if isinstance(v, dict) and "value" in v:
        cb = v.get("cpu_baseline") or {}
        print(" ", k, v["value"], "| cpu", cb.get("value"), cb.get("kind"), "x", v.get("speedup_vs_cpu_all_cores"))
    elif isinstance(v, dict) and "error" in v:
        print(" ", k, "ERROR", v["error"])
for p in d.get("secondary", {}).get("block_size_sweep", {}).get("points", []):
    print("  sweep", p["block_MiB"], "MiB x", p["blocks_per_step"], ":", p.get("compress"), "/", p.get("decompress"))
h = d.get("secondary", {}).get("host_path", {})
print("  host_path", h.get("compress_by_task_threads"), h.get("verify_decompress_by_task_threads"))
print("  wall", d.get("secondary", {}).get("wall_s_total"))
PY
if [ -n "$2" ]; then bash tools/r4_profile.sh $tag "$2" 2>&1 | tail -40; fi
