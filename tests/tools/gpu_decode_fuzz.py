"""(GPU) The adversarial chunk generators of isa_fuzz.py through the product's REDUCE-side call on the real machine (TEST
INFRASTRUCTURE; round 6: the batch decoder's parse blocks and exact-length stores are hand-written — the interpreter runs the same
generators on the CPU, this is the check that the hardware agrees: wait states, exec switches around LDS stores, 32 wavefronts per CU).

    python tests/tools/gpu_decode_fuzz.py --seed 5 --seconds 120 [--codec lz4|snappy|lzf|all]

Every round builds map outputs of `--parts` partitions (stitched corpora, records, planted copies at window-critical distances / lengths,
mutated periods, text, the planted sequence shapes; 1 byte .. 96 KiB), writes their images with the ORACLE (LZ4Block / SnappyOutputStream
frames; LZF chunks through the oracle's encoder), decodes them with s3s_decompress_range / s3s_decompress_ranges_batch (both decode
variants for LZ4 / Snappy) and compares with the source; then damages a copy of the image (a few bytes inside payloads) with the partition
checksums switched off: the call must come back (bad frame, or bytes — never a hang or a fault)."""
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


def lzf_image(oracle, data, offs, algo):
    """compress-lzf's LZFOutputStream chunks ('Z' 'V' 1 | clen BE | ulen BE | block, or 'Z' 'V' 0 | len BE | bytes), <= 65 535 bytes each, per
    partition; checksums over the compressed bytes (what bench.py builds through liblzf, here through the oracle's encoder)"""
    import zlib

    out, index, sums = [], [0], []
    for p in range(len(offs) - 1):
        part = data[offs[p]:offs[p + 1]]
        buf = bytearray()
        for pos in range(0, part.size, 65535):
            ch = part[pos:pos + 65535]
            blk = oracle.lzf_compress_block(ch)
            if blk.size < ch.size:
                buf += b"ZV\x01" + int(blk.size).to_bytes(2, "big") + int(ch.size).to_bytes(2, "big") + blk.tobytes()
            else:
                buf += b"ZV\x00" + int(ch.size).to_bytes(2, "big") + ch.tobytes()
        out.append(bytes(buf))
        index.append(index[-1] + len(buf))
        sums.append((zlib.adler32(bytes(buf)) if algo == 1 else zlib.crc32(bytes(buf))) & 0xFFFFFFFF)
    return np.frombuffer(b"".join(out), np.uint8).copy(), np.array(index, np.int64), np.array(sums, np.int64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--parts", type=int, default=48)
    ap.add_argument("--codec", default="all")
    a = ap.parse_args()
    rounds, mb, refused, bad = run(a.seed, a.seconds, a.parts, a.codec)
    print("DONE seed %d: %d rounds, %.1f MB per codec, %d damaged images refused, %d failures" % (a.seed, rounds, mb, refused, bad), flush=True)
    sys.exit(1 if bad else 0)


def run(seed, seconds, parts_n=48, codec_sel="all"):
    """-> (rounds, MB per codec, damaged images refused, failures)"""
    class A:
        pass
    a = A()
    a.seed, a.seconds, a.parts, a.codec = seed, seconds, parts_n, codec_sel
    import corpus
    import isa_fuzz as F
    import s3shuffle
    from oracle import binding as oracle

    rng = np.random.default_rng(a.seed)
    c = s3shuffle.Codec(0)
    codecs = {"lz4": [1], "snappy": [2], "lzf": [4], "all": [1, 2, 4]}[a.codec]
    t_end = time.time() + a.seconds
    rounds = n_bytes = bad = refused = 0
    while time.time() < t_end:
        parts = []
        for _ in range(a.parts):
            r = rng.random()
            n = max(1, F.pick_len(rng)) if r < 0.7 else int(rng.integers(32769, 98304))
            if rng.random() < 0.15:
                parts.append(corpus.planted_sequence_shapes(rng, n + 1600, range(0, 21), (4, 5, 7, 11, 12, 13, 17, 18, 19, 20, 33, 64, 65, 70, 150, 272, 273, 280, 300))[:n + 1600])
            else:
                parts.append(F.GENS[int(rng.integers(0, len(F.GENS)))](rng, n))
        data = np.concatenate(parts)
        offs = np.concatenate([[0], np.cumsum([len(p) for p in parts])]).astype(np.int64)
        for codec in codecs:
            algo = int(rng.integers(1, 3))
            if codec == 4:
                img, index, sums = lzf_image(oracle, data, offs, algo)
            else:
                img, index, sums = oracle.compress_map_output(codec, algo, data, offs)
            for variant in ((4,) if codec == 4 else (4, 3)):
                c.set_option(s3shuffle.codec.OPT_LZ4_DECODE_VARIANT, variant)
                back = c.decompress_range(codec, algo, img, index, sums, dst_capacity=data.size)
                if not np.array_equal(back, data):
                    bad += 1
                    print("MISMATCH codec %d variant %d round %d" % (codec, variant, rounds), flush=True)
            c.set_option(s3shuffle.codec.OPT_LZ4_DECODE_VARIANT, 4)
            # a sub-range as a batch of two ranges (batched entry point, absolute addressing)
            k = int(rng.integers(1, a.parts))
            res = c.decompress_ranges_batch(codec, algo, [
                (np.ascontiguousarray(img[:index[k]]).ctypes.data, int(index[k]), index[:k + 1], sums[:k], (o1 := np.empty(int(offs[k]), np.uint8)).ctypes.data, int(offs[k])),
                ((i2 := np.ascontiguousarray(img[index[k]:])).ctypes.data, int(i2.size), index[k:] - index[k], sums[k:], (o2 := np.empty(int(data.size - offs[k]), np.uint8)).ctypes.data, int(data.size - offs[k]))])
            if [r[0] for r in res] != [0, 0] or not np.array_equal(np.concatenate([o1, o2]), data):
                bad += 1
                print("BATCH MISMATCH codec %d round %d %s" % (codec, rounds, res), flush=True)
            # damage: a few bytes somewhere, checksums off -> must come back
            m = img.copy()
            for _ in range(int(rng.integers(1, 5))):
                m[int(rng.integers(0, m.size))] = int(rng.integers(0, 256))
            try:
                out = c.decompress_range(codec, 0, m, index, None, dst_capacity=data.size + 4096)
                assert out.size <= data.size + 4096
            except s3shuffle.CodecError as e:
                if e.code not in (-3, -2, -6):
                    bad += 1
                    print("unexpected error code", e.code, flush=True)
                refused += 1
            n_bytes += data.size
        rounds += 1
    c.close()
    return rounds, n_bytes / len(codecs) / 1e6, refused, bad


if __name__ == "__main__":
    main()
