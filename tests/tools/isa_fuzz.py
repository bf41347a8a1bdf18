"""Fuzz of the COMPILED compress kernels in the gfx950 interpreter against the oracle (TEST INFRASTRUCTURE).

    python tests/tools/isa_fuzz.py --seed 3 --minutes 30 [--codec lz4|snappy|both] [--out /tmp/isa_fuzz]

Every chunk is built by an adversarial generator (stitched corpora, records of random shape, planted copies at
window-critical distances and lengths, tiny alphabets, mutated periods), compressed by the compiled kernel
(tests/isa/lz4_kernel.py / snappy_kernel.py: hipcc's assembly incl. the hand-written window blocks) and compared
byte for byte with the oracle's block compressor.  A mismatch (or an interpreter fault = out-of-bounds access,
missing s_waitcnt) dumps the chunk to --out and is counted; exit status 1 if any.
"""
import argparse
import os
import sys
import time
import traceback

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
ROOT = os.path.dirname(TESTS)
for p in (ROOT, TESTS, os.path.join(TESTS, "isa")):
    if p not in sys.path:
        sys.path.insert(0, p)
import corpus  # noqa: E402
from oracle import binding as oracle  # noqa: E402


def gen_stitched(rng, n):
    parts, have = [], 0
    while have < n:
        k = int(rng.integers(0, corpus.N_KINDS))
        m = int(rng.integers(8, 3000))
        if k == 6:
            m = min(m, 1200)
        parts.append(corpus.chunk_corpus(k, m, rng))
        have += m
    return np.concatenate(parts)[:n].copy()


def gen_records(rng, n):
    rl = int(rng.integers(12, 320))
    nrand = int(rng.integers(0, min(rl, 24)))
    nrec = n // rl + 2
    rec = np.zeros((nrec, rl), np.uint8)
    rec[:, :] = rng.integers(32, 127, rl, dtype=np.uint8)[None, :]
    if nrand:
        rec[:, :nrand] = rng.integers(0, 256, (nrec, nrand))
    ids = np.arange(nrec) + int(rng.integers(0, 1 << 30))
    nd = int(rng.integers(0, min(12, rl - nrand) + 1))
    for k in range(nd):
        rec[:, nrand + k] = 48 + (ids // (10 ** (nd - 1 - k))) % 10
    if rng.random() < 0.5 and rl - nrand - nd > 8:
        c0 = nrand + nd
        rec[:, c0:c0 + 8] = (65 + (ids % int(rng.integers(2, 26))))[:, None]
    return rec.reshape(-1)[:n].copy()


def gen_planted(rng, n):
    """random bytes with copies planted at window-critical distances / lengths / alignments"""
    a = rng.integers(0, 256, n, dtype=np.uint8)
    if rng.random() < 0.5:
        a = (a % int(rng.integers(2, 64))).astype(np.uint8)
    dists = [1, 2, 3, 4, 5, 7, 8, 15, 16, 17, 60, 61, 62, 63, 64, 65, 66, 67, 68, 127, 128, 129, 191, 192, 193, 255,
             256, 257, 1000, 4095, 4096, 8191, 8192, 16383, 16384, 20000, 32000]
    lens = [4, 5, 6, 7, 8, 9, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 33, 60, 63, 64, 65, 68, 70, 128, 200, 255, 256,
            257, 258, 259, 260, 261, 262, 268, 269, 270, 271, 272, 273, 274, 280, 300, 500, 515, 530, 1000, 2000]
    i = int(rng.integers(4, 80))
    density = rng.random()
    while i < n - 4:
        if rng.random() < density:
            d = int(rng.choice(dists)) if rng.random() < 0.7 else int(rng.integers(1, 33000))
            L = int(rng.choice(lens)) if rng.random() < 0.7 else int(rng.integers(4, 400))
            if rng.random() < 0.3:  # end exactly at / around a 64-byte boundary
                e = ((i + L) | 63) + 1 + int(rng.integers(-3, 4))
                L = max(4, e - i)
            if d <= i:
                L = min(L, n - i)
                for k in range(L):
                    a[i + k] = a[i + k - d]
                i += L
                continue
        i += int(rng.integers(1, 1 + int(rng.choice([2, 6, 20, 70, 200]))))
    return a


def gen_period_mut(rng, n):
    p = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 12, 16, 31, 32, 33, 63, 64, 65, 100, 128, 250, 1000]))
    a = np.resize(rng.integers(0, 256, p, dtype=np.uint8), n).copy()
    m = int(rng.integers(0, max(1, n // int(rng.choice([8, 40, 200, 1000])))))
    if m:
        a[rng.integers(0, n, m)] = rng.integers(0, 256, m, dtype=np.uint8)
    return a


def gen_text(rng, n):
    words = [rng.integers(97, 123, int(rng.integers(2, 11)), dtype=np.uint8) for _ in range(int(rng.integers(5, 400)))]
    out, have = [], 0
    sep = np.array([32], np.uint8)
    while have < n:
        w = words[int(rng.integers(0, len(words)))]
        out += [w, sep]
        have += len(w) + 1
    return np.concatenate(out)[:n].copy()


GENS = [gen_stitched, gen_records, gen_planted, gen_period_mut, gen_text]


def pick_len(rng):
    r = rng.random()
    if r < 0.35:
        return 32768
    if r < 0.5:
        return int(rng.integers(32768 - 600, 32769))
    if r < 0.65:
        return int(rng.integers(1, 600))
    if r < 0.8:
        return 64 * int(rng.integers(1, 40)) + int(rng.integers(-3, 4))
    return int(rng.integers(600, 32768))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--minutes", type=float, default=10.0)
    ap.add_argument("--codec", default="both")
    ap.add_argument("--out", default="/tmp/isa_fuzz")
    ap.add_argument("--batch", type=int, default=6)
    ap.add_argument("--flags", default="", help="extra hipcc flags for the LZ4 kernel, e.g. -DS3S_ENGINE_SPEC (experiment builds)")
    args = ap.parse_args()
    lz4_flags = tuple(args.flags.split())
    os.makedirs(args.out, exist_ok=True)
    import lz4_kernel as lk
    import snappy_kernel as sk

    rng = np.random.default_rng(args.seed)
    t_end = time.time() + 60 * args.minutes
    n_chunks = n_bytes = n_bad = 0
    rounds = 0
    while time.time() < t_end:
        chunks = []
        for _ in range(args.batch):
            g = GENS[int(rng.integers(0, len(GENS)))]
            chunks.append(g(rng, max(1, pick_len(rng))))
        rounds += 1
        for codec in (("lz4", "snappy") if args.codec == "both" else (args.codec,)):
            try:
                if codec == "lz4":
                    res = lk.compress_chunks(chunks, flags=lz4_flags)
                    for c, (payload, hdr, w) in zip(chunks, res):
                        ref = oracle.lz4_compress_block(c)
                        ok = (len(ref) >= len(c)) if payload is None else np.array_equal(payload, ref)
                        if not ok:
                            raise AssertionError("lz4 mismatch len %d" % len(c))
                else:
                    res = sk.compress_chunks(chunks)
                    for c, (slot, sz, w) in zip(chunks, res):
                        ref = bytes(oracle.snappy_compress_block(c))
                        if bytes(slot[32:32 + sz - 4]) != ref:
                            raise AssertionError("snappy mismatch len %d" % len(c))
            except Exception:
                n_bad += 1
                tag = os.path.join(args.out, "bad_seed%d_round%d_%s" % (args.seed, rounds, codec))
                np.savez(tag + ".npz", *chunks)
                with open(tag + ".txt", "w") as f:
                    f.write(traceback.format_exc())
                print("FAIL", tag, flush=True)
        n_chunks += len(chunks)
        n_bytes += sum(len(c) for c in chunks)
        if rounds % 20 == 0:
            print("seed %d: %d rounds, %d chunks, %.1f MB, %d failures" % (args.seed, rounds, n_chunks, n_bytes / 1e6,
                                                                           n_bad), flush=True)
    print("DONE seed %d: %d chunks, %.1f MB, %d failures" % (args.seed, n_chunks, n_bytes / 1e6, n_bad), flush=True)
    return 1 if n_bad else 0


if __name__ == "__main__":
    sys.exit(main())
