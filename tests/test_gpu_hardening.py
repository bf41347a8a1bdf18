"""Decoder hardening on the GPU: LZ4Block / Snappy streams whose FRAMING is valid but
whose block payload is malformed must come back as S3S_E_BAD_FRAME without a byte written outside the caller's
destination range; and valid blocks that were NOT made by a greedy liblz4-style compressor (LZ4 HC, pyarrow,
hand-made sequences with offset-1/2/3 overlaps, 255 chains, far offsets, Snappy copy-4 / long literal tags) must
decode exactly.  Streams are built by tests/framing.py — no oracle, no product compressor."""
import numpy as np
import pytest

import corpus
import framing

pytestmark = pytest.mark.gpu

LZ4, SNAPPY = 1, 2
GUARD = 4096


@pytest.fixture(params=[3, 4], ids=["ring-valu", "batch"], autouse=True)
def decode_variant(request, gpu_codec):
    default = gpu_codec.get_option(5)
    gpu_codec.set_option(5, request.param)
    yield request.param
    gpu_codec.set_option(5, default)


def _decode_guarded(gpu_codec, codec, stream: bytes, capacity: int, misalign: int = 0):
    """Decode `stream` (one partition, no checksum) into the middle of a device buffer painted with 0xA5.
    Returns (rc, decoded bytes or None); asserts that nothing outside [dst, dst + capacity) changed."""
    import s3shuffle
    from hipdev import Dev

    dev = Dev()
    try:
        d_comp = dev.upload(np.frombuffer(stream, np.uint8).copy())
        total = capacity + 2 * GUARD + 16
        d_buf = dev.upload(np.full(total, 0xA5, np.uint8))
        o = GUARD + misalign
        rc, n = 0, 0
        try:
            n = gpu_codec.decompress_range_device(codec, 0, d_comp, len(stream), [0, len(stream)], None, d_buf + o, capacity)
        except s3shuffle.CodecError as e:
            rc = e.code
        host = dev.download(d_buf, total)
    finally:
        dev.free()
    assert (host[:o] == 0xA5).all() and (host[o + capacity:] == 0xA5).all(), "bytes outside the destination changed"
    return rc, (host[o:o + n].tobytes() if rc == 0 else None)


def test_lz4_foreign_valid_blocks(gpu_codec):
    rng = np.random.default_rng(21)
    blocks = []
    for kind in (0, 2, 3, 5, 7):
        d = corpus.chunk_corpus(kind, int(rng.choice([900, 9000, 32768])), rng)
        for level in (3, 9, 12):
            blocks.append((framing.lz4_hc(d, level), d.tobytes()))
    try:
        import pyarrow as pa

        d = corpus.chunk_corpus(1, 32768, rng)
        blocks.append((pa.Codec("lz4_raw").compress(d.tobytes()).to_pybytes(), d.tobytes()))
    except Exception:  # pragma: no cover
        pass
    lit = lambda n: rng.integers(0, 256, n).astype(np.uint8).tobytes()  # noqa: E731
    hand = [
        framing.lz4_block([(lit(5), 1, 4), (lit(1), 2, 300), (b"", 3, 19), (lit(20), 26, 4)], lit(7)),
        framing.lz4_block([(lit(15), 15, 19), (lit(270), 7, 270 + 4 + 255), (lit(14), 1, 18)], lit(15)),
        framing.lz4_block([(lit(600), 600, 600), (b"", 1200, 4), (lit(3), 1199, 5000)], lit(270)),
        framing.lz4_block([(lit(9000), 9000, 4), (b"", 8999, 64), (b"", 4100, 65), (lit(2), 5, 4)] * 2, lit(5)),
        framing.lz4_block([], lit(300)),
    ]
    for blk in hand:
        blocks.append((blk, framing.lz4_decode_py(blk)))
    stream = framing.lz4_stream(blocks)
    want = b"".join(o for _, o in blocks)
    for mis in (0, 3):
        rc, got = _decode_guarded(gpu_codec, LZ4, stream, len(want) + 100, mis)
        assert rc == 0 and got == want


def test_lz4_malformed_payload_in_valid_frames(gpu_codec):
    z = b"abcdefgh"
    good = framing.lz4_block([(z, 8, 40)], z)
    orig = framing.lz4_decode_py(good)
    bad_blocks = [
        framing.lz4_block([(z, 9, 4)], z),                       # offset beyond the start of the block
        framing.lz4_block([(z, 0, 4)], z),                       # offset 0
        framing.lz4_block([(z, 8, 4)], z)[:-3],                  # last literals run past compressedLen
        bytes([0xF0]) + b"\xff" * 40,                            # literal length chain to the end of the input
        framing.lz4_block([(z, 4, 19)], z)[:13] + b"\xff" * 9,   # match length chain to the end of the input
        framing.lz4_block([(z * 4, 8, 30000)], z),               # match far past originalLen
        framing.lz4_block([(z, 8, 40)], z * 3),                  # more output than originalLen
    ]
    for i, blk in enumerate(bad_blocks):
        # frame says originalLen = len(orig); check is the hash of `orig`, so only the bounds checks can object
        stream = framing.lz4_frame(good, orig) + framing.lz4_frame(blk, orig) + framing.lz4_end_frame()
        rc, _ = _decode_guarded(gpu_codec, LZ4, stream, 2 * len(orig) + 64)
        assert rc == -3, (i, rc)
    # random damage inside payloads (headers and checks untouched): BAD_FRAME or — if the bytes decode to the same
    # block — success; never a write outside the destination
    rng = np.random.default_rng(5)
    d = corpus.chunk_corpus(2, 32768, rng)
    payload = bytearray(framing.lz4_hc(d, 9))
    for it in range(24):
        p = bytearray(payload)
        for _ in range(int(rng.integers(1, 6))):
            p[int(rng.integers(0, len(p)))] = int(rng.integers(0, 256))
        stream = framing.lz4_frame(bytes(p), d.tobytes()) + framing.lz4_end_frame()
        rc, got = _decode_guarded(gpu_codec, LZ4, stream, d.size, int(rng.integers(0, 16)))
        assert rc == -3 or (rc == 0 and got == d.tobytes()), (it, rc)


def test_snappy_foreign_valid_and_malformed_blocks(gpu_codec):
    rng = np.random.default_rng(9)
    lit = lambda n: rng.integers(0, 256, n).astype(np.uint8).tobytes()  # noqa: E731
    good = [
        framing.snappy_block([("lit", lit(1)), ("copy", 1, 64, 2), ("lit", lit(61)), ("copy", 60, 11, 1), ("copy", 3, 9, 2)]),
        framing.snappy_block([("lit", lit(300)), ("copy", 300, 64, 4), ("lit", lit(5)), ("copy", 369, 4, 1)]),
        framing.snappy_block([("lit", lit(17000))], force_len_bytes=3),
        framing.snappy_block([("lit", lit(10))], force_len_bytes=4),
        framing.snappy_block([("lit", lit(8000)), ("copy", 8000, 64, 2)] + [("copy", 4200, 33, 2), ("lit", lit(2))] * 50),
    ]
    try:
        import pyarrow as pa

        d = corpus.chunk_corpus(3, 32768, rng)
        good.append(pa.Codec("snappy").compress(d.tobytes()).to_pybytes())
    except Exception:  # pragma: no cover
        pass
    want = b"".join(framing.snappy_decode_py(b) for b in good)
    rc, got = _decode_guarded(gpu_codec, SNAPPY, framing.snappy_stream(good), len(want) + 64, 5)
    assert rc == 0 and got == want
    z = b"abcdefgh"
    bad = [
        framing.snappy_block([("lit", z), ("copy", 9, 4, 2)]),           # offset before the block
        framing.snappy_block([("lit", z), ("copy", 0, 4, 2)]),           # offset 0
        framing.snappy_block([("lit", z), ("copy", 4, 8, 2)], ulen=12),  # copy past the declared length
        framing.snappy_block([("lit", z)], ulen=9),                      # shorter than declared
        framing.snappy_block([("lit", z * 10)])[:-5],                    # literal past the input
        framing.snappy_block([("lit", z), ("copy", 4, 8, 4)])[:-2],      # truncated copy
        framing.snappy_block([("lit", z), ("copy", 4, 64, 2)] * 3, ulen=100),
    ]
    for i, blk in enumerate(bad):
        rc, _ = _decode_guarded(gpu_codec, SNAPPY, framing.snappy_stream([good[0], blk]), 4096)
        assert rc == -3, (i, rc)


def _lzf_chunk(block: bytes, ulen: int) -> bytes:
    import struct

    return b"ZV\x01" + struct.pack(">HH", len(block), ulen) + block


def test_lzf_mutated_streams_agree_with_the_oracle_and_stay_inside(gpu_codec, oracle, decode_variant):
    """round 4: LZF chunk streams (liblzf-written blocks from tests/golden/lzf_liblzf.npz and the oracle's encoder) under random
    damage - flipped bits in chunk headers and block payloads, truncations, spliced chunks: the GPU decoder gives the oracle's
    verdict (same bytes when both accept; S3S_E_BAD_FRAME when the oracle refuses) and never writes outside the destination.
    (LZF has the batch decoder only: the decode-variant option does not change the path.)"""
    import os

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lzf_liblzf.npz"))
    rng = np.random.default_rng(77)
    seeds = []
    for n in ("records", "runs", "mixed", "far"):
        seeds.append((_lzf_chunk(g["lzf_" + n].tobytes(), g["raw_" + n].size), g["raw_" + n].size))
    d = corpus.chunk_corpus(7, 90_000, rng)
    s = oracle.compress_map_output(4, 0, d, [0, d.size])[0].tobytes()
    seeds.append((s, d.size))
    agree_ok = agree_bad = 0
    for it in range(120):
        stream, ulen = seeds[it % len(seeds)]
        b = bytearray(stream)
        how = it % 4
        if how == 0:
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
        elif how == 1:
            b = b[: int(rng.integers(1, len(b)))]
        elif how == 2:
            at = int(rng.integers(0, len(b)))
            b[at:at] = bytes(rng.integers(0, 256, int(rng.integers(1, 9)), dtype=np.uint8))
        else:
            other = seeds[(it + 1) % len(seeds)][0]
            cut = int(rng.integers(5, min(len(b), len(other))))
            b = b[:cut] + bytearray(other[cut:])
        b = bytes(b)
        cap = ulen + 300
        ref_rc, ref_out, _ = oracle.decompress_range(4, 0, np.frombuffer(b, np.uint8), [0, len(b)], None, cap)
        rc, out = _decode_guarded(gpu_codec, 4, b, cap, misalign=it % 5)
        if ref_rc == 0:
            assert rc == 0 and out == ref_out.tobytes(), (it, how)
            agree_ok += 1
        else:
            assert rc in (-3, -2), (it, how, rc, ref_rc)
            agree_bad += 1
    assert agree_bad > 30 and agree_ok > 5, (agree_ok, agree_bad)


def test_big_lz4_frames_malformed_payloads(gpu_codec, decode_variant):
    """round 4: LZ4Block frames of 256 KiB blocks (liblz4's byU32 parse) with damaged payloads - flipped bytes, an offset that points in
    front of the block 100 KB in, a literal run that overruns the block - are S3S_E_BAD_FRAME, valid ones decode exactly, and nothing outside
    the destination changes (the frame check catches what still decodes)."""
    import struct

    rng = np.random.default_rng(78)
    from s3shuffle import datagen

    data = datagen.terasort_map_output(600_000, 1, seed=9)[0]
    bs = 262144

    def stream(payload_edit=None):
        out = bytearray()
        for p in range(0, data.size, bs):
            chunk = np.ascontiguousarray(data[p:p + bs])
            pay = bytearray(framing.lz4_fast(chunk))
            if payload_edit and p == 0:
                payload_edit(pay)
            out += b"LZ4Block" + bytes([0x20 | 8]) + struct.pack("<iiI", len(pay), chunk.size, framing.xxh32(chunk.tobytes()) & 0x0FFFFFFF) + bytes(pay)
        out += b"LZ4Block" + bytes([0x10 | 8]) + struct.pack("<iii", 0, 0, 0)
        return bytes(out)

    rc, out = _decode_guarded(gpu_codec, LZ4, stream(), data.size, misalign=3)
    assert rc == 0 and out == data.tobytes()

    def flip(pay):
        for _ in range(3):
            pay[int(rng.integers(100, len(pay)))] ^= 0x5A

    def cut_tail(pay):
        del pay[-7:]

    for edit in (flip, cut_tail):
        for _ in range(3):
            rc, out = _decode_guarded(gpu_codec, LZ4, stream(edit), data.size)
            assert rc == -3, edit.__name__
    # hand-made: 100 000 literals, then a match whose offset reaches in front of the block
    z = rng.integers(0, 256, 100_000, dtype=np.uint8).tobytes()
    bad = framing.lz4_block([(z, 100_001, 8)], b"tail!")
    good = framing.lz4_block([(z, 100_000, 8)], b"tail!")
    orig = framing.lz4_decode_py(good, max_out=1 << 20)
    for blk, want in ((good, 0), (bad, -3)):
        s = b"LZ4Block" + bytes([0x20 | 8]) + struct.pack("<iiI", len(blk), len(orig), framing.xxh32(orig) & 0x0FFFFFFF) + blk + \
            b"LZ4Block" + bytes([0x10 | 8]) + struct.pack("<iii", 0, 0, 0)
        rc, out = _decode_guarded(gpu_codec, LZ4, s, len(orig), misalign=1)
        assert rc == want and (want != 0 or out == orig)
