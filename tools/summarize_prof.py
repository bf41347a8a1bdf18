"""Summarise rocprofv3 (rocpd sqlite) output of tools/gpu_round.sh: per-kernel stats from the
--kernel-trace --stats run and per-launch PMC means from the separate --pmc passes.
usage: python tools/summarize_prof.py <dir> [--md]   -> prints text (or markdown), writes summary.json
"""
import collections, glob, json, os, sqlite3, sys

root = sys.argv[1]
md = "--md" in sys.argv


def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("s3s::", "").replace("void ", "")
    n = n.split("(")[0]
    return n.strip()


out = {"kernels": {}, "pmc": {}}
for f in sorted(glob.glob(os.path.join(root, "trace", "**", "*.db"), recursive=True)):
    c = sqlite3.connect(f)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                     "max(vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(workgroup_x) "
                     "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print("== kernel trace (--kernel-trace --stats):", os.path.relpath(f, root))
    if md:
        print("| kernel | calls | avg us | min us | max us | % | vgpr | sgpr | lds B | grid | wg |\n|---|---|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        k = short(r[0])
        out["kernels"][k] = {"calls": r[1], "total_ns": r[2], "avg_ns": r[3], "min_ns": r[4], "max_ns": r[5],
                             "pct": 100.0 * r[2] / total, "vgpr": r[6], "sgpr": r[7], "lds": r[8], "grid": r[9], "wg": r[10]}
        if md:
            print(f"| {k} | {r[1]} | {r[3]/1e3:.1f} | {r[4]/1e3:.1f} | {r[5]/1e3:.1f} | {100.0*r[2]/total:.2f} | {r[6]} | {r[7]} | {r[8]} | {r[9]} | {r[10]} |")
        else:
            print(f"  {k:42s} calls={r[1]:4d} avg_us={r[3]/1e3:10.1f} min={r[4]/1e3:9.1f} max={r[5]/1e3:9.1f} pct={100.0*r[2]/total:6.2f} vgpr={r[6]} sgpr={r[7]} lds={r[8]} grid={r[9]} wg={r[10]}")
for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True)):
        c = sqlite3.connect(f)
        agg = collections.defaultdict(dict)
        for name, counter, mean, n in c.execute(
                "select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name"):
            agg[short(name)][counter] = (mean, n)
        print("== pmc pass:", os.path.relpath(f, root))
        for k, cs in agg.items():
            if k.startswith("__amd"):
                continue
            out["pmc"].setdefault(k, {}).update({cn: v[0] for cn, v in cs.items()})
            print(f"  {k:42s}", {cn: round(v[0], 1) for cn, v in cs.items()}, "launches=", max(v[1] for v in cs.values()))
json.dump(out, open(os.path.join(root, "summary.json"), "w"), indent=1)
# HBM traffic of the dominant kernel per launch: 2 x FETCH_SIZE (gfx950 half-count, calibrated on
# xxh32_items_kernel which reads its input exactly once) + WRITE_SIZE, KB -> bytes
for k, c in out["pmc"].items():
    if k.startswith("lz4_compress") and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        t = {"kernel": k, "lz4_compress_hbm_bytes_per_launch": int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024),
             "fetch_kb": c["FETCH_SIZE"], "write_kb": c["WRITE_SIZE"],
             "xxh32_fetch_kb (calibration: input read once)": (out["pmc"].get("xxh32_items_wave_kernel") or out["pmc"].get("xxh32_items_kernel", {})).get("FETCH_SIZE")}
        json.dump(t, open(os.path.join(root, "traffic.json"), "w"), indent=1)
        print("== traffic", t)
