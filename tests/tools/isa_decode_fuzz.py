"""Fuzz of the COMPILED batch decoder in the gfx950 interpreter (TEST INFRASTRUCTURE).

    python tests/tools/isa_decode_fuzz.py --seed 5 --minutes 20 [--fmt lz4|snappy|lzf|both] [--kernel batch|ring]

Valid blocks (sequence lists with chained / periodic / literal sources, oracle-compressed corpora) must decode to the
reference decoder's bytes; mutated blocks must be refused or decode to exactly what the reference decoder produces — and the
interpreter's memory has the payload buffer end with the last payload byte and the destination with the last declared byte,
so any access outside either is a fault, whatever the bytes say (tests/isa/decode_kernel.py).
"""
import argparse
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
ROOT = os.path.dirname(TESTS)
for p in (ROOT, TESTS, os.path.join(TESTS, "isa")):
    if p not in sys.path:
        sys.path.insert(0, p)
import corpus  # noqa: E402
import decode_kernel as dk  # noqa: E402
import framing  # noqa: E402
import test_batch_decode_model as tm  # noqa: E402
from oracle import binding as oracle  # noqa: E402


def _check(data):
    """the LZ4Block frame's check field (the LZ4 ring kernel verifies it; the other kernels ignore the field)"""
    import xxhash

    return xxhash.xxh32(bytes(data), seed=0x9747B28C).intdigest() & 0x0FFFFFFF


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--minutes", type=float, default=10)
    ap.add_argument("--fmt", default="both")
    ap.add_argument("--kernel", default="batch", choices=["batch", "ring"],
                    help="ring = lz4_decompress_valu_kernel / snappy_decompress_valu_kernel (decode variant 3, the fallback for big frames)")
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    fmts = [0, 1] if args.fmt == "both" else [{"lz4": 0, "snappy": 1, "lzf": 2}[args.fmt]]
    t0, n_valid, n_mut, bad = time.time(), 0, 0, 0
    while time.time() - t0 < args.minutes * 60:
        for fmt in fmts:
            cases = []
            for _ in range(6):
                if fmt == 2:  # LZF (round 6: its front end is a hand-written block too): the oracle's encoder over the corpora, liblzf-pinned decoder as reference
                    c = corpus.chunk_corpus(int(rng.integers(0, corpus.N_KINDS)), int(rng.integers(20, 9000)), rng)
                    blk = bytes(oracle.lzf_compress_block(c))
                    want = c.tobytes()
                elif rng.integers(0, 3) == 0:
                    c = corpus.chunk_corpus(int(rng.integers(0, corpus.N_KINDS)), int(rng.integers(20, 6000)), rng)
                    blk = bytes(oracle.lz4_compress_block(c) if fmt == 0 else oracle.snappy_compress_block(c))
                    want = c.tobytes()
                else:
                    seqs = tm._chain_sequences(rng, int(rng.integers(1, 300)))
                    if fmt == 0:
                        blk = framing.lz4_block([(rng.integers(0, 256, l).astype(np.uint8).tobytes(), o, m) for l, o, m in seqs],
                                                rng.integers(0, 256, 5 + int(rng.integers(0, 12))).astype(np.uint8).tobytes())
                        want = framing.lz4_decode_py(blk)
                    else:
                        els = []
                        for l, o, m in seqs:
                            if l:
                                els.append(("lit", rng.integers(0, 256, l).astype(np.uint8).tobytes()))
                            while m > 0:
                                k = min(m, 64)
                                els.append(("copy", o, k, 2))
                                m -= k
                        blk = framing.snappy_block(els)
                        want = framing.snappy_decode_py(blk)
                if want is None or not 0 < len(want) <= 32768:
                    continue
                cases.append((blk, want))
            if not cases:
                continue
            try:
                res, st, _ = dk.decode_blocks([(b, len(w)) for b, w in cases], fmt=fmt, kernel=args.kernel,
                                              checks=[_check(w) for _, w in cases])
                n_valid += len(cases)
                if st != 0 or res != [w for _, w in cases]:
                    bad += 1
                    print("VALID BLOCK MISMATCH fmt", fmt, flush=True)
            except Exception as e:  # a fault of the interpreter's memory, a missing wait, an unmodelled opcode
                bad += 1
                print("FAULT on valid blocks:", str(e)[:200], flush=True)
            for blk, want in cases:  # mutations, one block per launch (the status word is per launch)
                p = bytearray(blk)
                for _ in range(int(rng.integers(1, 4))):
                    p[int(rng.integers(0, len(p)))] = int(rng.integers(0, 256))
                if fmt == 2:  # (the block does not carry its decoded length: the frame record's length is the source's)
                    r = oracle.lzf_decompress_block(np.frombuffer(bytes(p), np.uint8), len(want))
                    ref = None if isinstance(r, int) or len(r) != len(want) else r.tobytes()
                else:
                    ref = (framing.lz4_decode_py if fmt == 0 else framing.snappy_decode_py)(bytes(p))
                olen = len(ref) if ref is not None and 0 < len(ref) <= 32768 else len(want)
                if fmt == 1:  # a Snappy block declares its length: the frame record must agree or the kernel refuses
                    pass
                try:
                    res, st, _ = dk.decode_blocks([(bytes(p), olen)], fmt=fmt, kernel=args.kernel,
                                                  checks=[_check(ref if ref is not None and len(ref) == olen else want)])
                    n_mut += 1
                    if st == 0 and ref is not None and len(ref) == olen and res[0] != ref:
                        bad += 1
                        print("MUTATION decoded differently from the reference, fmt", fmt, flush=True)
                    elif st not in (0, -3):
                        bad += 1
                        print("unexpected status", st, flush=True)
                except Exception as e:
                    bad += 1
                    print("FAULT on a mutated block:", str(e)[:200], flush=True)
    print("DONE seed %d: %d valid blocks, %d mutations, %d failures" % (args.seed, n_valid, n_mut, bad), flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
