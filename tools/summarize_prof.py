"""Summarise rocprofv3 CSV output under gpurun_out/prof_<tag>/ (kernel stats + PMC means)."""
import csv, glob, os, sys, collections
root = sys.argv[1]
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats", f)
    for i, row in enumerate(csv.DictReader(open(f))):
        if i < 12:
            print({k: row[k] for k in row if k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage")})
for d in sorted(glob.glob(os.path.join(root, "pmc*"))):
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            agg[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        print("== pmc", f)
        for k, cs in agg.items():
            if "lz4" in k or "checksum" in k or "gather" in k or "decompress" in k or "xxh" in k:
                print(" ", k, {c: round(sum(v) / len(v), 1) for c, v in cs.items()}, "n=", len(next(iter(cs.values()))))
