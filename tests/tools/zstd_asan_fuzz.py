"""Memory-safety fuzz of the Zstandard decoder core under ASan / UBSan (TEST INFRASTRUCTURE).

    python tests/tools/zstd_asan_fuzz.py --seed 3 --minutes 20 [--flags -DZS_X_SOMETHING]

Seed streams are written by libzstd 1.4.8 (oracle/zstd_ref.py: streaming frames as zstd-jni writes them, levels 1 / 5 / 19,
with and without checksums, concatenated frames); tests/model/zstd_asan_fuzz.cpp mutates and decodes them from exact-size
heap buffers.  Exit status 1 and the sanitizer's report if the decoder ever touches a byte outside its buffers."""
import argparse
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
ROOT = os.path.dirname(TESTS)
for p in (ROOT, TESTS, os.path.join(ROOT, "spark-s3-shuffle_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def build(flags=(), out_dir=None):
    out_dir = out_dir or os.path.join(TESTS, "model")
    exe = os.path.join(out_dir, "zstd_asan_fuzz" + "".join("_" + f.lstrip("-D").lower() for f in flags))
    src = os.path.join(TESTS, "model", "zstd_asan_fuzz.cpp")
    core = os.path.join(ROOT, "spark-s3-shuffle_amd", "csrc", "zstd_decode_core.h")
    if not os.path.exists(exe) or max(os.path.getmtime(src), os.path.getmtime(core)) > os.path.getmtime(exe):
        subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-Wextra", "-fsanitize=address,undefined",
                        "-fno-sanitize-recover=undefined", *flags, src, "-o", exe], check=True)
    return exe


def write_seeds(path, seed=1):
    import corpus
    from oracle import zstd_ref as z
    from s3shuffle import datagen

    rng = np.random.default_rng(seed)
    srcs = [datagen.terasort_map_output(150_000, 1, seed=seed)[0], datagen.tpcds_wide_map_output(140_000, 1, seed=seed + 1)[0],
            datagen.kv_int_map_output(60_000, 1, seed=seed + 2)[0], np.zeros(70_000, np.uint8),
            rng.integers(0, 256, 30_000, dtype=np.uint8), rng.integers(0, 4, 300, dtype=np.uint8)]
    srcs += [corpus.chunk_corpus(k, 3000 if k == 6 else 50_000, rng) for k in range(corpus.N_KINDS)]
    n = 0
    with open(path, "wb") as f:
        for i, data in enumerate(srcs):
            for level in ((1, 5, 19) if i < 3 else (1,)):
                for checksum in (False, True):
                    comp = z.compress_stream(data, level, checksum=checksum)
                    f.write(struct.pack("<II", comp.size, data.size) + comp.tobytes())
                    n += 1
        a, b = srcs[0][:40_000], srcs[1][:30_000]  # two frames in one partition (a spill merge)
        comp = np.concatenate([z.compress_stream(a, 1, checksum=False), z.compress_stream(b, 1, checksum=False)])
        f.write(struct.pack("<II", comp.size, a.size + b.size) + comp.tobytes())
    return n + 1


def run(seed, seconds, flags=(), max_mutations=-1):
    exe = build(flags)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "seeds.bin")
        write_seeds(path, seed)
        return subprocess.run([exe, path, str(seed), str(seconds), str(max_mutations)], capture_output=True, text=True,
                              env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--minutes", type=float, default=10)
    ap.add_argument("--flags", nargs="*", default=[])
    a = ap.parse_args()
    r = run(a.seed, a.minutes * 60, tuple(a.flags))
    print(r.stdout, r.stderr[-6000:])
    sys.exit(r.returncode)
