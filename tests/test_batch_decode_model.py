"""The lock-step CPU model of the batch decoder (tests/model/lz4_batch_decode_model.cpp = the algorithm
lz4_decode_batch.hip runs on one wavefront, LZ4 and Snappy front ends) against independent encoders and decoders:
valid blocks from the oracle, liblz4 (greedy + HC) and pyarrow; hand-made blocks; and mutated blocks, which must be
rejected or decoded exactly like liblz4 / libsnappy do — and never touch a byte outside the buffers (the model
checks every index it forms).  Runs on the CPU-only box."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import corpus
import framing

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "model", "lz4_batch_decode_model.cpp")
SO = os.path.join(HERE, "model", "liblz4_batch_decode_model.so")
LZ4, SNAPPY = 0, 1


@pytest.fixture(scope="module")
def model():
    if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(SRC):
        subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", SO, SRC], check=True)
    L = ctypes.CDLL(SO)
    L.batch_decode_model.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                     ctypes.c_int, ctypes.c_void_p]
    return L


def run(model, fmt, comp, olen, mis=0):
    """Decode into a guarded buffer whose address is `mis` mod 16.  Returns (rc, bytes); rc -2 = the model formed
    an out-of-bounds index."""
    guard = 64
    buf = np.full(olen + 2 * guard + 16, 0xA5, np.uint8)
    o = guard + (mis - (buf.ctypes.data + guard)) % 16
    cc = np.ascontiguousarray(np.frombuffer(bytes(comp), np.uint8)) if not isinstance(comp, np.ndarray) else np.ascontiguousarray(comp)
    if cc.size == 0:
        cc = np.zeros(1, np.uint8)[:0]
    rc = model.batch_decode_model(fmt, cc.ctypes.data, cc.size, buf.ctypes.data + o, olen, mis, None)
    assert (buf[:o] == 0xA5).all() and (buf[o + olen:] == 0xA5).all(), "guard bytes overwritten"
    return rc, buf[o:o + olen].copy()


SIZES = [1, 5, 12, 13, 20, 64, 65, 130, 1000, 4096, 4097, 7616, 8192, 8200, 12288, 20000, 32767, 32768]


@pytest.mark.parametrize("kind", range(corpus.N_KINDS))
def test_lz4_valid_blocks(model, oracle, kind):
    rng = np.random.default_rng(500 + kind)
    for n in SIZES:
        if kind == 6 and n > 6000:
            continue
        d = corpus.chunk_corpus(kind, n, rng)
        for comp in (oracle.lz4_compress_block(d), np.frombuffer(framing.lz4_hc(d, 9), np.uint8)):
            for mis in (0, 7, 15):
                rc, got = run(model, LZ4, comp, d.size, mis)
                assert rc == 0 and np.array_equal(got, d), (kind, n, mis, rc)


@pytest.mark.parametrize("kind", range(corpus.N_KINDS))
def test_snappy_valid_blocks(model, oracle, kind):
    import pyarrow as pa

    sn = pa.Codec("snappy")
    rng = np.random.default_rng(600 + kind)
    for n in [0] + SIZES:
        if kind == 6 and n > 6000:
            continue
        d = corpus.chunk_corpus(kind, n, rng) if n else np.zeros(0, np.uint8)
        for comp in (oracle.snappy_compress_block(d), np.frombuffer(sn.compress(d.tobytes()), np.uint8)):
            for mis in (0, 5):
                rc, got = run(model, SNAPPY, comp, d.size, mis)
                assert rc == 0 and np.array_equal(got, d), (kind, n, mis, rc)


def test_lz4_hand_made_blocks(model):
    rng = np.random.default_rng(3)
    lit = lambda n: rng.integers(0, 256, n).astype(np.uint8).tobytes()  # noqa: E731
    cases = [
        # offset 1 / 2 / 3 overlaps mid block, long lengths with 255 chains in both fields, maximum offsets
        framing.lz4_block([(lit(5), 1, 4), (lit(1), 2, 300), (b"", 3, 19), (lit(20), 26, 4)], lit(7)),
        framing.lz4_block([(lit(15), 15, 15 + 4), (lit(270), 7, 270 + 4 + 255), (lit(14), 1, 18)], lit(15)),
        framing.lz4_block([(lit(600), 600, 600), (b"", 1200, 4), (lit(3), 1199, 5000)], lit(255 + 15)),
        framing.lz4_block([(lit(9000), 9000, 4), (b"", 8999, 64), (b"", 4100, 65), (lit(2), 5, 4)] * 2, lit(5)),
        framing.lz4_block([], lit(1)),
        framing.lz4_block([], lit(300)),
    ]
    for i, blk in enumerate(cases):
        want = framing.lz4_decode_py(blk)
        assert want is not None
        rc, got = run(model, LZ4, blk, len(want), i)
        assert rc == 0 and got.tobytes() == want, i
        # any wrong output length must be refused
        for olen in (len(want) - 1, len(want) + 1):
            if olen > 0:
                assert run(model, LZ4, blk, olen)[0] == -1, (i, olen)


def _chain_sequences(rng, n_seq, max_out=30000):
    """Random (literal length, offset, match length) lists in which matches copy from inside earlier matches, from
    literal runs, from periodic patterns and across several of them — what the decoder's source redirection
    (a match inside the output of one earlier match reads from that match's source) has to get right."""
    seqs, pos, regions = [], 0, []  # regions: (start, end) of earlier match outputs
    for _ in range(n_seq):
        lit = int(rng.choice([0, 0, 0, 1, 2, 3, 5, 10, 14, 15, 16, 40]))
        pos += lit
        if pos == 0:
            lit = 4
            pos = 4
        ml = int(rng.choice([4, 5, 7, 8, 15, 16, 17, 19, 29, 32, 33, 48, 59, 64, 65, 100]))
        how = rng.integers(0, 8)
        if how <= 2 and regions:      # wholly inside one earlier match output (the last few)
            rs, re_ = regions[-int(rng.integers(1, min(len(regions), 6) + 1))]
            ml = min(ml, re_ - rs)
            src = int(rng.integers(rs, re_ - ml + 1))
            off = pos - src
        elif how == 3:                # short period
            off = int(rng.choice([1, 2, 3, 4, 5, 8]))
        elif how == 4 and regions:    # straddles the end of an earlier match output
            rs, re_ = regions[-1]
            off = pos - max(re_ - 2, 0)
        elif how == 5:                # into the own literals
            off = max(1, min(lit, pos))
        else:
            off = int(rng.integers(1, pos + 1))
        off = max(1, min(off, pos, 65535))
        seqs.append((lit, off, ml))
        regions.append((pos, pos + ml))
        pos += ml
        if pos > max_out:
            break
    return seqs


def test_redirected_sources(model):
    rng = np.random.default_rng(11)
    for it in range(300):
        seqs = _chain_sequences(rng, int(rng.integers(1, 400)))
        blk = framing.lz4_block([(rng.integers(0, 256, l).astype(np.uint8).tobytes(), o, m) for l, o, m in seqs],
                                rng.integers(0, 256, 5 + int(rng.integers(0, 20))).astype(np.uint8).tobytes())
        want = framing.lz4_decode_py(blk)
        assert want is not None and len(want) <= 32768
        rc, got = run(model, LZ4, blk, len(want), it)
        assert rc == 0 and got.tobytes() == want, it
        # the same structure as Snappy elements (copies of up to 64 bytes)
        els = []
        for l, o, m in seqs:
            if l:
                els.append(("lit", rng.integers(0, 256, l).astype(np.uint8).tobytes()))
            while m > 0:
                k = min(m, 64)
                els.append(("copy", o, k, 2 if (o > 2047 or k > 11 or k < 4 or rng.integers(0, 2)) else 1))
                m -= k
        sblk = framing.snappy_block(els)
        swant = framing.snappy_decode_py(sblk)
        if swant is not None and 0 < len(swant) <= 32768:
            rc, got = run(model, SNAPPY, sblk, len(swant), it)
            assert rc == 0 and got.tobytes() == swant, it


def test_blocks_above_32k(model):
    """round 4: the batch decoder takes LZ4 blocks of any size (records relative to the batch): liblz4 fast / HC blocks of
    70 000 - 300 000 bytes with offsets up to 65 535, a chained 90 KB block; every index inside the buffers, no round reads a
    byte before its last writer; a malformed big block is refused"""
    from s3shuffle import datagen

    rng = np.random.default_rng(31)
    big1 = datagen.terasort_map_output(70_000, 1, seed=5)[0]
    big2 = np.concatenate([datagen.tpcds_wide_map_output(200_000, 1, seed=6)[0][:200_000], corpus.chunk_corpus(7, 100_000, rng)])
    cases = [(framing.lz4_fast(big1), big1.tobytes()), (framing.lz4_hc(big2, 9), big2.tobytes()), (framing.lz4_fast(big2), big2.tobytes())]
    seqs = _chain_sequences(rng, 6000, max_out=90_000)
    b = framing.lz4_block([(rng.integers(0, 256, l).astype(np.uint8).tobytes(), o, m) for l, o, m in seqs], b"abcdefg")
    cases.append((b, framing.lz4_decode_py(b)))
    assert len(cases[-1][1]) > 40_000
    for mis in (0, 5):
        for c, w in cases:
            rc, got = run(model, 0, c, len(w), mis)
            assert rc == 0 and got.tobytes() == w
    z = rng.integers(0, 256, 40_000).astype(np.uint8).tobytes()
    bad = framing.lz4_block([(z, 40_001, 8)], b"tail!")
    rc, _ = run(model, 0, bad, 40_000 + 8 + 5)
    assert rc == -1


def test_lz4_malformed_blocks_are_refused(model):
    z = b"abcdefgh"
    bad = [
        framing.lz4_block([(z, 9, 4)], z),                      # offset beyond the start of the block
        framing.lz4_block([(z, 0, 4)], z),                      # offset 0
        framing.lz4_block([(z, 8, 4)], z)[:-3],                 # last literals run past the input
        bytes([0xF0]) + b"\xff" * 40,                           # literal length chain runs to the end of input
        framing.lz4_block([(z, 4, 19)], z)[:13] + b"\xff" * 9,  # match length chain runs to the end of input
        framing.lz4_block([(z, 4, 4)], b"")[:-1],               # ends right behind an offset
        b"",
    ]
    for i, blk in enumerate(bad):
        for olen in (16, 100, 32768):
            rc, _ = run(model, LZ4, blk, olen)
            assert rc == -1, (i, olen, rc)
    # a match that runs past the declared original length
    blk = framing.lz4_block([(z, 8, 40)], z)
    assert run(model, LZ4, blk, 56)[0] == 0
    assert run(model, LZ4, blk, 40)[0] == -1


def test_snappy_hand_made_and_malformed(model):
    rng = np.random.default_rng(4)
    lit = lambda n: rng.integers(0, 256, n).astype(np.uint8).tobytes()  # noqa: E731
    good = [
        framing.snappy_block([("lit", lit(1)), ("copy", 1, 64, 2), ("lit", lit(61)), ("copy", 60, 11, 1), ("copy", 3, 9, 2)]),
        framing.snappy_block([("lit", lit(300)), ("copy", 300, 64, 4), ("lit", lit(5)), ("copy", 369, 4, 1)]),
        framing.snappy_block([("lit", lit(70000 // 4))], force_len_bytes=3),
        framing.snappy_block([("lit", lit(10))], force_len_bytes=4),
        framing.snappy_block([("lit", lit(8000)), ("copy", 8000, 64, 2)] + [("copy", 4200, 33, 2), ("lit", lit(2))] * 50),
        framing.snappy_block([]),
    ]
    for i, blk in enumerate(good):
        want = framing.snappy_decode_py(blk)
        assert want is not None, i
        rc, got = run(model, SNAPPY, blk, len(want), i)
        assert rc == 0 and got.tobytes() == want, i
    z = b"abcdefgh"
    bad = [
        framing.snappy_block([("lit", z), ("copy", 9, 4, 2)]),             # offset before the block
        framing.snappy_block([("lit", z), ("copy", 0, 4, 2)]),             # offset 0
        framing.snappy_block([("lit", z), ("copy", 4, 8, 2)], ulen=12),    # copy past the declared length
        framing.snappy_block([("lit", z)], ulen=9),                        # shorter than declared
        framing.snappy_block([("lit", z * 10)])[:-5],                      # literal past the input
        framing.snappy_block([("lit", z), ("copy", 4, 8, 4)])[:-2],        # truncated copy
        b"\xff\xff\xff\xff\xff\xff",                                       # endless varint
    ]
    for i, blk in enumerate(bad):
        assert framing.snappy_decode_py(blk) is None, i
        for olen in (8, 12, 16, 80):
            rc, _ = run(model, SNAPPY, blk, olen)
            assert rc == -1, (i, olen, rc)


def test_mutations_agree_with_the_libraries(model, oracle):
    """Random damage: the model never leaves its buffers; what it accepts is what liblz4 / libsnappy decode."""
    import pyarrow as pa

    L = framing.liblz4()
    sn = pa.Codec("snappy")
    rng = np.random.default_rng(77)
    for it in range(1200):
        kind = int(rng.integers(0, corpus.N_KINDS))
        n = int(rng.choice([40, 200, 1000, 5000, 12000]))
        if kind == 6:
            n = min(n, 5000)
        d = corpus.chunk_corpus(kind, n, rng)
        fmt = it & 1
        comp = (oracle.snappy_compress_block(d) if fmt else
                (np.frombuffer(framing.lz4_hc(d, 9), np.uint8) if it & 2 else oracle.lz4_compress_block(d))).copy()
        for _ in range(int(rng.integers(1, 4))):
            comp[int(rng.integers(0, comp.size))] = rng.integers(0, 256)
        if it % 5 == 0 and comp.size > 8:
            comp = comp[: int(rng.integers(1, comp.size))].copy()
        rc, got = run(model, fmt, comp, d.size, int(rng.integers(0, 16)))
        assert rc in (0, -1), (it, rc)
        if fmt == LZ4:
            ref = np.empty(d.size, np.uint8)
            rn = L.LZ4_decompress_safe(comp.ctypes.data, ref.ctypes.data, comp.size, d.size)
            if rc == 0 and rn == d.size:
                assert np.array_equal(got, ref), it
        else:
            try:
                ref = np.frombuffer(sn.decompress(pa.py_buffer(comp.tobytes()), decompressed_size=d.size), np.uint8)
            except Exception:
                ref = None
            assert (rc == 0) == (ref is not None), it
            if rc == 0:
                assert np.array_equal(got, ref), it
