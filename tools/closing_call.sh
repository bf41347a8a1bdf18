#!/bin/bash
# (GPU) closing call of a round — ONE per round, at the final sources: GPU suite, then the rocprofv3 trace + PMC passes of the
# kernels as they ship (tools/profile_sets.sh) and their report (tools/profile_report.py, run here on the box so that
# profiles/traffic_latest.json carries the stamp of exactly these sources), then the driver's bench command — whose line
# therefore has roofline.traffic of the same sources — and everything the repo tracks copied under gpurun_out/<tag>/profiles_out.
#   gpurun --timeout 2700 -- 'bash tools/closing_call.sh r06z "compress decompress crc2000 snappy_compress snappy_decompress zstd hbm" <commit>'
#   then here: cp gpurun_out/r06z/profiles_out/* profiles/
tag=${1:-r06z}
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
O=gpurun_out/$tag; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest_gpu.txt
if [ -n "$2" ]; then
  bash tools/profile_sets.sh $tag "$2" 2>&1 | tail -60
  python tools/profile_report.py $tag ${3:-unknown} 2>&1 | tail -12
fi
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_stdout.txt 2>$O/bench_full.err
tail -n 1 $O/bench_stdout.txt > $O/bench_headline.json      # THE line the driver parses (compact)
tail -n 2 $O/bench_stdout.txt | head -n 1 > $O/bench_secondary_summary.json
cp bench_secondary.json $O/bench_full.json                   # full record: headline with long descriptions + every secondary leg
rm -f $O/bench_stdout.txt
python - <<PY | tee $O/bench_full.txt
import json
d = json.loads(open("$O/bench_full.json").read())
print("headline", d["value"], "GB/s; ms/step", d["ms_per_step"], "; cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["kind"], "speedup", d.get("speedup_vs_cpu_all_cores"),
      "roofline", {k: d["roofline"].get(k) for k in ("achieved", "frac", "avg_launch_ms", "traffic", "traffic_over_algorithmic")})
for k, v in d.get("secondary", {}).items():
    if isinstance(v, dict) and "value" in v:
        cb = v.get("cpu_baseline") or {}
        print(" ", k, v["value"], "| cpu", cb.get("value"), cb.get("kind"), "x", v.get("speedup_vs_cpu_all_cores"), "| verified", v.get("bytes_verified"))
    elif isinstance(v, dict) and "error" in v:
        print(" ", k, "ERROR", v["error"])
for p in d.get("secondary", {}).get("block_size_sweep", {}).get("points", []):
    print("  sweep", p["block_MiB"], "MiB x", p["blocks_per_step"], ":", p.get("compress"), "/", p.get("decompress"), "threads", p.get("compress_task_threads"))
for p in d.get("secondary", {}).get("block_size_sweep", {}).get("points_more_blocks_in_flight", []):
    print("  sweep (more blocks in flight)", p["block_MiB"], "MiB x", p["blocks_per_step"], ":", p.get("compress"), "/", p.get("decompress"))
for k, v in d.get("secondary", {}).get("hbm_bound_stages", {}).items():
    if isinstance(v, dict) and "roofline" in v:
        print("  hbm", k, v["value"], "achieved", v["roofline"]["achieved"], "frac", v["roofline"]["frac"], "traffic", v["roofline"].get("traffic"))
h = d.get("secondary", {}).get("host_path", {})
print("  host_path", h.get("compress_by_task_threads"), h.get("verify_decompress_by_task_threads"))
print("  wall", d.get("secondary", {}).get("wall_s_total"))
PY
mkdir -p $O/profiles_out
cp profiles/${tag}_* profiles/traffic_latest.json $O/profiles_out/ 2>/dev/null
cp $O/bench_headline.json $O/profiles_out/${tag}_bench_headline.json
cp $O/bench_secondary_summary.json $O/profiles_out/${tag}_bench_secondary_summary.json
cp $O/bench_full.json $O/profiles_out/${tag}_bench_full.json
cp $O/bench_full.txt $O/profiles_out/${tag}_bench_full.txt
cp $O/pytest_gpu.txt $O/profiles_out/${tag}_pytest_gpu.txt
