"""(GPU) The adversarial chunk generators of isa_fuzz.py through the product's map-side call on the real machine (TEST INFRASTRUCTURE).

    python tests/tools/gpu_compress_fuzz.py --seed 5 --seconds 120 [--codec lz4|snappy|both]

Every round builds a map output of `--parts` partitions - each a chunk of one of the generators (stitched corpora, records, planted
copies at window-critical distances / lengths, mutated periods, text) plus the sequence shapes of tests/corpus.py's
planted_sequence_shapes, 1 byte .. 96 KiB - compresses it with s3s_compress_map_output (both window blocks, deferred emission
included) and compares image, index and checksums with the oracle's.  The interpreter runs the same generators on the CPU; this is the
check that the hardware agrees (wait states, lane order of stores, unaligned dword stores)."""
import argparse
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
ROOT = os.path.dirname(TESTS)
for p in (ROOT, TESTS, HERE, os.path.join(ROOT, "spark-s3-shuffle_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--parts", type=int, default=64)
    ap.add_argument("--codec", default="both")
    a = ap.parse_args()
    import corpus
    import isa_fuzz as F
    import s3shuffle
    from oracle import binding as oracle

    rng = np.random.default_rng(a.seed)
    c = s3shuffle.Codec(0)
    codecs = [1, 2] if a.codec == "both" else [1 if a.codec == "lz4" else 2]
    t_end = time.time() + a.seconds
    rounds = n_bytes = bad = 0
    while time.time() < t_end:
        parts = []
        for _ in range(a.parts):
            r = rng.random()
            n = max(1, F.pick_len(rng)) if r < 0.7 else int(rng.integers(32769, 98304))
            if rng.random() < 0.15:
                parts.append(corpus.planted_sequence_shapes(rng, n + 1600, range(0, 21), (4, 5, 7, 11, 12, 13, 17, 18, 19, 20, 33, 64, 65, 70, 150, 272, 273, 280))[:max(n, 1)])
            else:
                parts.append(F.GENS[int(rng.integers(0, len(F.GENS)))](rng, n))
        data = np.concatenate(parts)
        offs = np.concatenate([[0], np.cumsum([len(p) for p in parts])]).astype(np.int64)
        for codec in codecs:
            algo = int(rng.integers(1, 4))
            img, index, sums = c.compress_map_output(codec, algo, data, offs)
            r_img, r_index, r_sums = oracle.compress_map_output(codec, algo, data, offs)
            ok = np.array_equal(index, r_index) and np.array_equal(sums, r_sums) and np.array_equal(img, r_img)
            if not ok:
                bad += 1
                first = int(np.nonzero(img[:min(img.size, r_img.size)] != r_img[:min(img.size, r_img.size)])[0][0]) if img.size and r_img.size and not np.array_equal(img[:min(img.size, r_img.size)], r_img[:min(img.size, r_img.size)]) else -1
                print("MISMATCH seed %d round %d codec %d first differing byte %d" % (a.seed, rounds, codec, first), flush=True)
                np.save("/tmp/gpu_compress_fuzz_bad_%d_%d_%d.npy" % (a.seed, rounds, codec), data)
        rounds += 1
        n_bytes += data.size * len(codecs)
    print("gpu_compress_fuzz seed %d: %d rounds, %d partitions, %.1f MB through the GPU, %d mismatches" % (a.seed, rounds, rounds * a.parts, n_bytes / 1e6, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
