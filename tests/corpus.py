"""Seeded byte corpora shared by the parity tests (small enough for the oracle in seconds)."""
import numpy as np


def chunk_corpus(kind: int, n: int, rng: np.random.Generator) -> np.ndarray:
    if kind == 0:
        return rng.integers(0, 256, n, dtype=np.uint8)          # incompressible -> RAW frames
    if kind == 1:
        return np.zeros(n, np.uint8)                             # one giant match
    if kind == 2:
        return rng.integers(0, 4, n, dtype=np.uint8)             # tiny alphabet, dense hash collisions
    if kind == 3:
        w = rng.integers(0, 256, (50, 8), dtype=np.uint8)        # 8-byte dictionary words
        return w[rng.integers(0, 50, n // 8 + 1)].reshape(-1)[:n].copy()
    if kind == 4:
        a = rng.integers(0, 256, n, dtype=np.uint8)              # every 3rd byte fixed
        a[::3] = 65
        return a
    if kind == 5:
        p = int(rng.integers(1, 40))                             # short period
        return np.resize(rng.integers(0, 256, p, dtype=np.uint8), n).copy()
    if kind == 6:
        a = rng.integers(97, 123, n, dtype=np.uint8)             # text with back-references
        i = 0
        while i < n - 20:
            if rng.random() < 0.3 and i > 10:
                L = min(int(rng.integers(4, 60)), n - i)
                d = int(rng.integers(1, min(i, 3000)))
                for k in range(L):
                    a[i + k] = a[i + k - d]
                i += L
            else:
                i += int(rng.integers(1, 12))
        return a
    if kind == 7:
        nrec = n // 100 + 1                                      # terasort-like
        rec = np.zeros((nrec, 100), np.uint8)
        rec[:, :10] = rng.integers(0, 256, (nrec, 10))
        ids = np.arange(nrec) + int(rng.integers(0, 1 << 40))
        hexd = np.frombuffer(b"0123456789ABCDEF", np.uint8)
        for k in range(32):
            sh = 4 * (31 - k)
            rec[:, 10 + k] = hexd[(ids >> sh) & 15] if sh < 63 else 48
        for j in range(7):
            rec[:, 42 + 8 * j:50 + 8 * j] = (65 + ((ids + j) % 26))[:, None]
        rec[:, 98] = 13
        rec[:, 99] = 10
        return rec.reshape(-1)[:n].copy()
    raise ValueError(kind)


N_KINDS = 8

EDGE_LENGTHS = [0, 1, 5, 12, 13, 14, 15, 16, 17, 20, 31, 63, 64, 65, 66, 67, 100, 130, 255, 256,
                257, 1000, 4096, 32767, 32768, 32769, 65535, 65536, 65537, 100000]


def ragged_map_output(rng: np.random.Generator, n_parts: int, max_len: int, kinds=None):
    """A map output with empty, 1-byte, sub-chunk and multi-chunk partitions."""
    parts = []
    for p in range(n_parts):
        r = rng.random()
        if r < 0.15:
            n = 0
        elif r < 0.25:
            n = int(rng.integers(1, 16))
        elif r < 0.35:
            n = int(rng.choice([32767, 32768, 32769, 65536]))
        else:
            n = int(rng.integers(16, max_len))
        kind = int(rng.integers(0, N_KINDS)) if kinds is None else int(rng.choice(kinds))
        if kind == 6 and n > 5000:
            kind = 7
        parts.append(chunk_corpus(kind, n, rng))
    offsets = np.zeros(n_parts + 1, np.int64)
    np.cumsum([p.size for p in parts], out=offsets[1:])
    data = np.concatenate(parts) if parts else np.zeros(0, np.uint8)
    return data.astype(np.uint8), offsets


def planted_sequence_shapes(rng, n, lits, mlens):
    """a dictionary of random bytes, then units of `lit` fresh bytes + a copy of `m` dictionary bytes, every (lit, m) pair in turn"""
    dic = rng.integers(0, 256, 1536, dtype=np.uint8)
    parts, have = [dic], len(dic)
    pairs = [(a, m) for a in lits for m in mlens]
    k = 0
    while have < n:
        a, m = pairs[k % len(pairs)]
        k += 1
        o = int(rng.integers(0, len(dic) - m))
        parts += [rng.integers(0, 256, a, dtype=np.uint8), dic[o:o + m]]
        have += a + m
    return np.concatenate(parts)[:n].copy()


def zstd_frame_with_oversized_huffman_literals(regen: int = 131072, stream_bytes: int = 40) -> "np.ndarray":
    """A hand-built Zstandard frame (RFC 8878) of a few hundred bytes whose only block declares a 4-stream Huffman literals
    section regenerating `regen` bytes out of 4 x `stream_bytes` compressed ones — more than 8 symbols per compressed byte,
    which no Huffman code can do.  A decoder that sizes its literal scratch by the partition's compressed size (the product's
    single pass: min(8 * size + 256, 128 KiB) + 64) must refuse it BEFORE writing stream k at k * regen / 4 (advisor r5)."""
    import numpy as np

    table = bytes([0x80, 0x10])                      # direct weights: one explicit 4-bit weight (1), the second implied: 2 symbols of 1 bit
    stream = bytes([0x55] * (stream_bytes - 1) + [0x01])  # 1-bit symbols, last byte = the end marker alone
    jump = (stream_bytes.to_bytes(2, "little")) * 3
    huf = table + jump + stream * 4
    lcomp = len(huf)
    v = 2 | (3 << 2) | (regen << 4) | (lcomp << 22)  # ltype 2 (Huffman with a table), size format 3: 18 + 18 bits, 4 streams
    lit_hdr = v.to_bytes(5, "little")
    seq = bytes([0])                                  # no sequences: the block is its literals
    block = lit_hdr + huf + seq
    bh = (1 | (2 << 1) | (len(block) << 3)).to_bytes(3, "little")  # last block, compressed
    # frame header: single segment, 4-byte frame content size
    fhd = bytes([0x20 | (2 << 6)]) + regen.to_bytes(4, "little")
    return np.frombuffer(bytes([0x28, 0xB5, 0x2F, 0xFD]) + fhd + bh + block, dtype=np.uint8).copy()
