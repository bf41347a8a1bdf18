"""Fuzz of WHOLE map-side calls through the compiled kernels in the gfx950 interpreter (TEST INFRASTRUCTURE).

    python tests/tools/isa_map_side_fuzz.py --seed 7 --minutes 15

Random map outputs — partition counts, empty / one-byte / multi-block partitions, compressible and incompressible corpora — go
through tests/isa/map_side.py (frame-check pre-pass, LZ4 or Snappy kernel, item scan, gather, Adler32 / CRC32) with every
buffer of exactly its size and must equal the oracle's image, index and checksums; now and then the destination is a few
bytes short and the call must answer S3S_E_CAPACITY without touching anything behind it."""
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
import map_side as ms  # noqa: E402
from oracle import binding as oracle  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--minutes", type=float, default=10)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    t0, n, bad = time.time(), 0, 0
    while time.time() - t0 < a.minutes * 60:
        parts = []
        for _ in range(int(rng.integers(1, 9))):
            size = int(rng.choice([0, 0, 1, 2, 21, 700, 5000, 32767, 32768, 32769, 50_000]))
            kind = int(rng.integers(0, corpus.N_KINDS + 1))
            if size == 0:
                parts.append(b"")
            elif kind == corpus.N_KINDS:
                parts.append(rng.integers(0, 256, size, dtype=np.uint8).tobytes())
            else:
                parts.append(corpus.chunk_corpus(kind, min(size, 6000) if kind == 6 else size, rng).tobytes())
        codec, algo = int(rng.integers(1, 3)), int(rng.integers(0, 3))
        data = np.frombuffer(b"".join(parts), np.uint8)
        offs = np.concatenate([[0], np.cumsum([len(p) for p in parts])]).astype(np.int64)
        img, idx, sums = oracle.compress_map_output(codec, algo, data, offs)
        short = int(rng.integers(1, 30)) if img.size > 40 and rng.integers(0, 6) == 0 else 0
        n += 1
        try:
            st, got, gi, gs = ms.compress_map_output(parts, algo, img.size - short, codec=codec)
        except Exception as e:  # a fault of the interpreter's memory, a missing wait, an unmodelled opcode
            bad += 1
            print("FAULT (codec %d, algo %d, sizes %s): %s" % (codec, algo, [len(p) for p in parts], str(e)[:300]), flush=True)
            continue
        if short:
            ok = st == -2 and gi == list(idx)
        else:
            ok = st == 0 and got == img.tobytes() and gi == list(idx) and (algo == 0 or gs == [int(x) for x in sums])
        if not ok:
            bad += 1
            print("MISMATCH (codec %d, algo %d, short %d, sizes %s)" % (codec, algo, short, [len(p) for p in parts]), flush=True)
    print("DONE seed %d: %d map outputs, %d failures" % (a.seed, n, bad), flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
