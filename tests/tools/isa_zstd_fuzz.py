"""Fuzz of the COMPILED Zstandard decoder in the gfx950 interpreter (TEST INFRASTRUCTURE).

    python tests/tools/isa_zstd_fuzz.py --seed 5 --minutes 20

libzstd streams of small corpora are mutated (bit flips, truncations, random bytes, inserted / deleted runs) and decoded by
zstd_partitions_kernel as hipcc compiles it, both passes, with source, destination and literal scratch of exactly their sizes
(tests/isa/zstd_kernel.py): the device-only parts of the decoder (wavefront-wide copies through the LDS ring, the literal
window, lane-strided table builds) under the interpreter's bounds checks.  A mutation must be refused or decode to what
libzstd decodes; any fault of the interpreter's memory is a failure.  The host build of the same core runs millions of
mutations under ASan (tests/tools/zstd_asan_fuzz.py); this is the slow, exact counterpart for the compiled device code."""
import argparse
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
ROOT = os.path.dirname(TESTS)
for p in (ROOT, TESTS, os.path.join(TESTS, "isa"), os.path.join(ROOT, "spark-s3-shuffle_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
import corpus  # noqa: E402
import zstd_kernel as zk  # noqa: E402
from oracle import zstd_ref as z  # noqa: E402
from s3shuffle import datagen  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--minutes", type=float, default=10)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    srcs = [datagen.terasort_map_output(12_000, 1, seed=a.seed)[0], datagen.tpcds_wide_map_output(9_000, 1, seed=a.seed + 1)[0],
            np.zeros(5000, np.uint8), rng.integers(0, 256, 3000, dtype=np.uint8), rng.integers(0, 4, 700, dtype=np.uint8)]
    srcs += [corpus.chunk_corpus(k, 2500 if k == 6 else 6000, rng) for k in range(corpus.N_KINDS)]
    seeds = [(z.compress_stream(d, lvl, checksum=False), d) for d in srcs for lvl in (1, 5)]
    t0, n, refused, same, bad = time.time(), 0, 0, 0, 0
    while time.time() - t0 < a.minutes * 60:
        comp, data = seeds[int(rng.integers(0, len(seeds)))]
        m = bytearray(comp.tobytes())
        kind = int(rng.integers(0, 5))
        for _ in range(int(rng.integers(1, 4))):
            if kind == 0:
                m[int(rng.integers(0, len(m)))] ^= 1 << int(rng.integers(0, 8))
            elif kind == 1:
                m = m[: int(rng.integers(1, len(m) + 1))]
            elif kind == 2:
                i = int(rng.integers(0, len(m)))
                m[i:i + 4] = rng.integers(0, 256, len(m[i:i + 4]), dtype=np.uint8).tobytes()
            elif kind == 3:
                i = int(rng.integers(0, len(m)))
                m[i:i] = rng.integers(0, 256, int(rng.integers(1, 9)), dtype=np.uint8).tobytes()
            else:
                i = int(rng.integers(0, len(m)))
                del m[i:i + int(rng.integers(1, 9))]
            if not m:
                m = bytearray(b"\0")
        cap = data.size + int(rng.integers(0, 3)) * 100 if rng.integers(0, 4) else int(rng.integers(0, data.size + 1))
        ref = z.decompress(np.frombuffer(bytes(m), np.uint8), cap)
        n += 1
        try:
            outs, rcs, _ = zk.decode_partitions([(bytes(m), cap)])
        except Exception as e:  # a fault of the interpreter's memory, a missing wait, an unmodelled opcode
            bad += 1
            print("FAULT:", str(e)[:300], flush=True)
            open("/tmp/isa_zstd_fuzz_fault_%d_%d.bin" % (a.seed, n), "wb").write(bytes(m))
            continue
        if rcs[0] != 0:
            refused += 1
        elif ref is not None and outs[0] is not None and bytes(outs[0][: len(ref)]) == bytes(ref):
            same += 1
        elif ref is not None:
            bad += 1
            print("decoded differently from libzstd (mutation %d)" % n, flush=True)
    print("DONE seed %d: %d mutations, %d refused, %d decoded like libzstd, %d failures" % (a.seed, n, refused, same, bad), flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
