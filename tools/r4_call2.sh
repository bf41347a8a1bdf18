#!/bin/bash
# (GPU) round 4, second call: GPU suite, the driver's bench line (now with secondary.block_size_sweep), then the profile of the
# kernels as they ship.   gpurun --timeout 1500 -- 'bash tools/r4_call2.sh r04b'
tag=${1:-r04b}
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
O=gpurun_out/$tag; mkdir -p $O
timeout 500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest_gpu.txt
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 2>$O/bench_full.err | grep '^{' > $O/bench_full.json
python - <<PY | tee $O/bench_full.txt
import json
d = json.loads(open("$O/bench_full.json").read())
print("headline", d["value"], "GB/s; cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["kind"], "speedup", d.get("speedup_vs_cpu_all_cores"), "traffic", d["roofline"].get("traffic"))
for k, v in d.get("secondary", {}).items():
    if isinstance(v, dict) and "value" in v:
        cb = v.get("cpu_baseline") or {}
        print(" ", k, v["value"], "| cpu", cb.get("value"), cb.get("kind"), "x", v.get("speedup_vs_cpu_all_cores"))
sw = d.get("secondary", {}).get("block_size_sweep", {})
for p in sw.get("points", []):
    print("  sweep", p)
print("  host_path", d.get("secondary", {}).get("host_path", {}).get("compress_by_task_threads"), d.get("secondary", {}).get("host_path", {}).get("verify_decompress_by_task_threads"))
print("  wall", d.get("secondary", {}).get("wall_s_total"))
PY
bash tools/r4_profile.sh $tag "${2:-compress snappy_compress decompress crc2000 snappy_decompress}" 2>&1 | tail -70
