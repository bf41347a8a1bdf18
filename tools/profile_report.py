"""Profile report: gpurun_out/<tag>/prof_*/summary.json (tools/summarize_prof.py output of tools/profile_sets.sh) ->
profiles/<tag>_<name>_rocprofv3_summary.{md,json} with derived figures, and profiles/traffic_latest.json, every entry stamped
with the sha256 of the kernel sources it was taken with (gpurun_out/<tag>/kernel_sources_sha256.txt, tools/src_stamp.py):
bench.py uses an entry as roofline.traffic only when that stamp equals the one of the library it runs.

usage: python tools/profile_report.py <tag> [<commit>]
Derived per dominant kernel (per launch):
  cycles        GRBM_GUI_ACTIVE / 8 (the counter is summed over the 8 XCDs; PMC passes serialise the kernels)
  issue util    (SALU + VALU + LDS + VMEM_RD + VMEM_WR instructions) / (1024 SIMDs x cycles)
  scalar / vector issue busy   SALU (VALU) instructions / (256 CUs x cycles): a CU issues one of each per cycle
  waves / CU    4 x SQ_WAVE_CYCLES / (cycles x 256)        (SQ_*_CYCLES count quad-cycles)
  wait share    SQ_WAIT_ANY / SQ_WAVE_CYCLES
  L2 hit rate   TCC_HIT / (TCC_HIT + TCC_MISS)
  HBM bytes     (2 x FETCH_SIZE + WRITE_SIZE) KB: gfx950 FETCH_SIZE counts half (calibrated in the same pass on
                xxh32_items_wave_kernel, which reads its input exactly once)
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import subprocess

tag = sys.argv[1]
commit = sys.argv[2] if len(sys.argv) > 2 else subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
G = os.path.join(ROOT, "gpurun_out", tag)
P = os.path.join(ROOT, "profiles")

# name: (dominant kernel, traffic key, map tasks per launch of that kernel under the profiled command)
#   compress: --maps-per-gpu 8, four task threads -> two map tasks per launch (the headline's shape);
#   the others: compress side --maps-per-gpu 4, four threads -> one task per launch; reduce side (round 6) --maps-per-gpu 8, two task
#   threads -> four per launch
SETS = {
    "compress": ("lz4_compress_l2_kernel<true>", "terasort-10g-200p-lz4:compress", 2),
    "snappy_compress": ("snappy_compress_kernel<true>", "tpcds-wide-100g-200p-snappy:compress", 1),
    "decompress": ("batch_decode_kernel<0>", "terasort-10g-200p-lz4:decompress", 4),   # --maps-per-gpu 8, two task threads x 4 (round 6)
    "crc2000": ("lz4_compress_l2_kernel<true>", "terasort-100g-2000p-lz4-crc32:compress", 1),
    "snappy_decompress": ("batch_decode_kernel<1>", "tpcds-wide-100g-200p-snappy:decompress", 4),
    "zstd": ("zstd_partitions_kernel", "terasort-10g-200p-zstd:decompress", 4),
}
# round 5: the HBM-bound stage lines (bench.py --hbm-stages-only): traffic per launch of each kernel, keyed for run_hbm_stages
HBM_KERNELS = {"checksum_segments_kernel<1>": "hbm-stages:adler32", "checksum_segments_kernel<2>": "hbm-stages:crc32",
               "xxh32_items_quad_kernel<false>": "hbm-stages:xxh32"}
stamp_file = os.path.join(G, "kernel_sources_sha256.txt")
stamp = open(stamp_file).read().strip() if os.path.exists(stamp_file) else None
traffic_file = os.path.join(P, "traffic_latest.json")
traffic = json.load(open(traffic_file)) if os.path.exists(traffic_file) else {}
for name, (kernel, key, tasks) in SETS.items():
    src = os.path.join(G, "prof_" + name, "summary.json")
    if not os.path.exists(src):
        continue
    cmd_file = os.path.join(G, "prof_" + name, "command.txt")
    cmd = open(cmd_file).read().strip().replace(ROOT + "/", "") if os.path.exists(cmd_file) else "bench.py"
    cmd = cmd[cmd.find("bench.py"):] if "bench.py" in cmd else cmd
    d = json.load(open(src))
    k = d["kernels"].get(kernel)
    c = d["pmc"].get(kernel, {})
    if not k or not c:
        continue
    cyc = c.get("GRBM_GUI_ACTIVE", 0) / 8.0
    insts = sum(c.get(x, 0) for x in ("SQ_INSTS_SALU", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"))
    der = {
        "kernel": kernel, "command": cmd, "commit": commit, "kernel_sources_sha256": stamp,
        "avg_kernel_us_trace": round(k["avg_ns"] / 1e3, 1), "min_kernel_us_trace": round(k["min_ns"] / 1e3, 1),
        "calls": k["calls"], "vgpr": k["vgpr"], "sgpr": k["sgpr"], "lds_bytes": k["lds"], "workgroups": k["grid"] // max(k["wg"], 1),
        "kernel_cycles_alone": round(cyc), "kernel_ms_alone_at_2.4GHz": round(cyc / 2.4e6, 3),
        "instructions_per_launch": round(insts),
        "issue_slot_utilisation": round(insts / (1024 * cyc), 4) if cyc else None,
        # a CU issues at most ONE scalar and ONE vector instruction per cycle for all of its wavefronts (the scalar unit is
        # shared by the four SIMDs; a wave64 vector instruction occupies its SIMD16 for four cycles)
        "scalar_issue_busy": round(c.get("SQ_INSTS_SALU", 0) / (256 * cyc), 4) if cyc else None,
        "vector_issue_busy": round(c.get("SQ_INSTS_VALU", 0) / (256 * cyc), 4) if cyc else None,
        "resident_waves_per_cu_avg": round(4 * c.get("SQ_WAVE_CYCLES", 0) / (cyc * 256), 2) if cyc else None,
        "wait_share_of_wave_cycles": round(c.get("SQ_WAIT_ANY", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1), 3),
        "l2_hit_rate": round(c.get("TCC_HIT_sum", 0) / max(c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0), 1), 3),
        "fetch_kb": round(c.get("FETCH_SIZE", 0), 1), "write_kb": round(c.get("WRITE_SIZE", 0), 1),
        "hbm_bytes_per_launch": int((2 * c.get("FETCH_SIZE", 0) + c.get("WRITE_SIZE", 0)) * 1024),
        "pmc": {x: round(v, 1) for x, v in sorted(c.items())},
    }
    cal = (d["pmc"].get("xxh32_items_quad_kernel<false>", {}).get("FETCH_SIZE") or d["pmc"].get("xxh32_items_wave_kernel", {}).get("FETCH_SIZE")
           or d["pmc"].get("lz4_verify_frames_kernel", {}).get("FETCH_SIZE"))
    der["fetch_calibration_kernel_kb"] = cal
    json.dump({"derived": der, "kernels": d["kernels"], "pmc": d["pmc"]}, open(os.path.join(P, f"{tag}_{name}_rocprofv3_summary.json"), "w"), indent=1)
    with open(os.path.join(P, f"{tag}_{name}_rocprofv3_summary.md"), "w") as f:
        f.write(f"# {tag} — rocprofv3 of `{cmd}` (commit {commit})\n\n")
        f.write("Kernel trace (`--kernel-trace --stats`; four task threads share the GPU, so per-launch durations overlap) and separate "
                "`--pmc` passes (FETCH_SIZE / WRITE_SIZE / SQ x2 / TCC; kernels run one at a time there). Produced by `tools/profile_sets.sh` + `tools/profile_report.py`.\n\n")
        f.write("| kernel | calls | avg us | min us | max us | % of GPU time | vgpr | sgpr | lds B | workgroups |\n|---|---|---|---|---|---|---|---|---|---|\n")
        for kn, kv in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["total_ns"]):
            f.write(f"| {kn} | {kv['calls']} | {kv['avg_ns']/1e3:.1f} | {kv['min_ns']/1e3:.1f} | {kv['max_ns']/1e3:.1f} | {kv['pct']:.2f} | {kv['vgpr']} | {kv['sgpr']} | {kv['lds']} | {kv['grid']//max(kv['wg'],1)} |\n")
        f.write(f"\n## Dominant kernel `{kernel}` — derived per launch\n\n")
        for a, b in der.items():
            if a != "pmc":
                f.write(f"* {a}: {b}\n")
        f.write("\nRaw PMC means per launch:\n\n```\n" + json.dumps(der["pmc"], indent=1) + "\n```\n")
    traffic[key] = {"kernel": kernel, "hbm_bytes_per_launch": der["hbm_bytes_per_launch"], "fetch_kb": der["fetch_kb"],
                    "write_kb": der["write_kb"], "map_tasks_per_launch": tasks, "avg_kernel_us": der["avg_kernel_us_trace"],
                    "kernel_ms_alone": der["kernel_ms_alone_at_2.4GHz"], "l2_hit_rate": der["l2_hit_rate"],
                    "issue_slot_utilisation": der["issue_slot_utilisation"], "resident_waves_per_cu_avg": der["resident_waves_per_cu_avg"],
                    "commit": commit, "kernel_sources_sha256": stamp,
                    "source": f"profiles/{tag}_{name}_rocprofv3_summary.md ({cmd}; 2 x FETCH_SIZE + WRITE_SIZE, gfx950 half-count "
                              f"correction calibrated on the streaming kernel of the same run)"}
    print(name, json.dumps({a: b for a, b in der.items() if a != "pmc"}))
hsrc = os.path.join(G, "prof_hbm", "summary.json")
if os.path.exists(hsrc):
    d = json.load(open(hsrc))
    rows = []
    for kn, key in HBM_KERNELS.items():
        c, k = d["pmc"].get(kn, {}), d["kernels"].get(kn)
        if not c or not k:
            continue
        hbm = int((2 * c.get("FETCH_SIZE", 0) + c.get("WRITE_SIZE", 0)) * 1024)
        traffic[key] = {"kernel": kn, "hbm_bytes_per_launch": hbm, "fetch_kb": round(c.get("FETCH_SIZE", 0), 1), "write_kb": round(c.get("WRITE_SIZE", 0), 1),
                        "avg_kernel_us": round(k["avg_ns"] / 1e3, 1), "commit": commit, "kernel_sources_sha256": stamp,
                        "source": f"profiles/{tag}_hbm_rocprofv3_summary.md (bench.py --hbm-stages-only; 2 x FETCH_SIZE + WRITE_SIZE per launch over a 1 GiB range)"}
        rows.append((kn, k, c, hbm))
    json.dump({"kernels": d["kernels"], "pmc": d["pmc"]}, open(os.path.join(P, f"{tag}_hbm_rocprofv3_summary.json"), "w"), indent=1)
    with open(os.path.join(P, f"{tag}_hbm_rocprofv3_summary.md"), "w") as f:
        f.write(f"# {tag} — rocprofv3 of `bench.py --hbm-stages-only` (commit {commit}): checksums and xxHash32 over a 1 GiB range\n\n")
        f.write("| kernel | calls | avg us (trace) | HBM bytes per launch (2 x FETCH + WRITE) | bytes / 2^30 | SALU + VALU instructions | LDS instructions | wave cycles in s_waitcnt |\n|---|---|---|---|---|---|---|---|\n")
        for kn, k, c, hbm in rows:
            f.write(f"| {kn} | {k['calls']} | {k['avg_ns']/1e3:.1f} | {hbm} | {hbm / 2**30:.3f} | {c.get('SQ_INSTS_SALU', 0) + c.get('SQ_INSTS_VALU', 0):.0f} | "
                    f"{c.get('SQ_INSTS_LDS', 0):.0f} | {c.get('SQ_WAIT_ANY', 0) / max(c.get('SQ_WAVE_CYCLES', 1), 1):.3f} |\n")
    print("hbm", {k: traffic[k]["hbm_bytes_per_launch"] for k in traffic if k.startswith("hbm-stages")})
json.dump(traffic, open(traffic_file, "w"), indent=1)
for f in sorted(os.listdir(G)):  # the closing call's bench files (tools/closing_call.sh): headline line, summary line, full record
    if f.startswith("bench") and f.endswith(".json"):
        open(os.path.join(P, f"{tag}_{f}"), "w").write(open(os.path.join(G, f)).read())
