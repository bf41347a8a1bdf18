"""A/B of two versions of the window blocks in the interpreter's cycle model (TEST INFRASTRUCTURE, no GPU).

    python tests/tools/engine_ab_model.py [<base commit, default b044428 = the window blocks before the deferred emission of round 5>]

Builds hipcc's assembly of lz4_compress.hip / snappy_compress.hip once with the working tree's *_window_engine.inc and
once with the base commit's, runs three 32 KiB blocks of each bench.py generator through both in tests/isa/gfx950_emu.py
(outputs must be identical) and prints SALU + VALU instructions per byte and the model's cycles per block (Cost in
gfx950_emu.py: 5 per issued instruction, 25 per taken branch, 128 / 700 / 300 for an LDS / load / store round trip the
wave waits for).  The model knows nothing about other wavefronts or cache misses: use it for ratios, not for times.
"""
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
ROOT = os.path.dirname(TESTS)
for p in (ROOT, TESTS, os.path.join(TESTS, "isa"), os.path.join(ROOT, "spark-s3-shuffle_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
import numpy as np  # noqa: E402

import gfx950_emu as emu  # noqa: E402
import lz4_kernel as lk  # noqa: E402
import snappy_kernel as sk  # noqa: E402
from s3shuffle import datagen  # noqa: E402


def base_asm(commit):
    d = tempfile.mkdtemp(prefix="s3s_ab_")
    c = os.path.join(d, "a", "b", "csrc")
    shutil.copytree(lk.CSRC, c)
    shutil.copytree(os.path.join(ROOT, "include"), os.path.join(d, "a", "include"))
    for f in ("lz4_window_engine.inc", "snappy_window_engine.inc"):
        blob = subprocess.run(["git", "-C", ROOT, "show", "%s:spark-s3-shuffle_amd/csrc/%s" % (commit, f)], check=True,
                              capture_output=True).stdout
        open(os.path.join(c, f), "wb").write(blob)
    out = {}
    for src in ("lz4_compress.hip", "snappy_compress.hip"):
        o = os.path.join(d, src + ".s")
        subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-S",
                        "--cuda-device-only", "-w", "-I" + os.path.join(d, "a", "include"), src, "-o", o], check=True, cwd=c)
        out[src] = open(o).read()
    shutil.rmtree(d)
    return out


def main():
    commit = sys.argv[1] if len(sys.argv) > 1 else "b044428"
    old = base_asm(commit)
    t = old["lz4_compress.hip"]
    e = lk.find_kernel(t, "lz4_compress_l2_kernelILb1E")
    lz_old = (emu.Program(t, e), e)
    t = old["snappy_compress.hip"]
    e = lk.find_kernel(t, "snappy_compress_kernelILb1E")
    sn_old = (emu.Program(t, e), e, t)
    lz_new, sn_new = lk.program(True), sk.program(True)
    for wl, gen, seed in (("terasort", datagen.terasort_map_output, 2), ("wide rows", datagen.tpcds_wide_map_output, 3)):
        data, offs = gen(8 << 20, 12, seed=seed, map_id=0)
        data = np.asarray(data, dtype=np.uint8)
        p0 = int(offs[3])
        chunks = [data[p0 + k * 32768: p0 + (k + 1) * 32768] for k in range(3)]
        for codec, mod, progs in (("lz4", lk, (lz_old, lz_new)), ("snappy", sk, (sn_old, sn_new))):
            r = []
            for pr in progs:
                mod._PROGS[(True, ())] = pr
                out = mod.compress_chunks(chunks)
                ws = [o[2] for o in out]
                sig = [bytes(o[0]) if codec == "lz4" else bytes(o[0][:o[1] + 32]) for o in out]
                r.append((sum(w.clock for w in ws) / len(ws), sum(w.n_salu + w.n_valu for w in ws) / len(ws), sig))
            mod._PROGS[(True, ())] = progs[1]
            assert r[0][2] == r[1][2], "outputs differ"
            print("%-9s %-6s  SALU+VALU per byte %5.2f -> %5.2f (x%.3f)   model cycles per block %8.0f -> %8.0f (x%.3f)" % (
                wl, codec, r[0][1] / 32768, r[1][1] / 32768, r[1][1] / r[0][1], r[0][0], r[1][0], r[1][0] / r[0][0]))


if __name__ == "__main__":
    main()
