"""Where a compiled compressor spends its instructions and its model cycles, per label of the hand-written block
(TEST INFRASTRUCTURE, no GPU):   python tests/tools/block_profile.py terasort|tpcds lz4|snappy [blocks]"""
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
ROOT = os.path.dirname(TESTS)
for p in (ROOT, TESTS, os.path.join(TESTS, "isa"), os.path.join(ROOT, "spark-s3-shuffle_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
import numpy as np  # noqa: E402

from s3shuffle import datagen  # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "terasort"
    codec = sys.argv[2] if len(sys.argv) > 2 else "lz4"
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    if codec == "lz4":
        import lz4_kernel as mod
        pre = ".Lw_"
    else:
        import snappy_kernel as mod
        pre = ".Ls_"
    gen, seed = (datagen.terasort_map_output, 2) if wl == "terasort" else (datagen.tpcds_wide_map_output, 3)
    data, offs = gen(8 << 20, 12, seed=seed, map_id=0)
    data = np.asarray(data, dtype=np.uint8)
    p0 = int(offs[3])
    chunks = [data[p0 + k * 32768: p0 + (k + 1) * 32768] for k in range(n)]
    prof = {}
    out = mod.compress_chunks(chunks, profile=prof)
    ws = [o[2] for o in out]
    tot_i = sum(v[0] for v in prof.values())
    tot_c = sum(w.clock for w in ws)
    agg = {}
    for k, v in prof.items():
        kk = re.sub(r"\d*_\d+$", "", k) if k.startswith(pre) else "compiled code"
        a = agg.setdefault(kk, [0, 0, 0])
        a[0] += v[0]
        a[1] += v[1]
        a[2] += v[2] if len(v) > 2 else 0
    print("%s %s: %.0f instructions, %.0f SALU + VALU (%.2f per byte), %.0f model cycles per 32 KiB block" % (
        wl, codec, tot_i / n, sum(w.n_salu + w.n_valu for w in ws) / n, sum(w.n_salu + w.n_valu for w in ws) / n / 32768, tot_c / n))
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][2])[:int(os.environ.get("ROWS", "16"))]:
        print("   %-16s instructions %7d (%4.1f %%)  taken branches %5d  model cycles %8d (%4.1f %%)" % (
            k, v[0] / n, 100 * v[0] / tot_i, v[1] / n, v[2] / n, 100 * v[2] / tot_c))


if __name__ == "__main__":
    main()
