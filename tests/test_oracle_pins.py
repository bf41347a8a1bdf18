"""Pins the CPU oracle (oracle/, test infrastructure) before anything trusts it.

The reference (IBM/spark-s3-shuffle) holds no golden vectors for this path: its
six tests are round trips.  The arithmetic lives in third-party code reached through Spark's
CompressionCodec, so the oracle is pinned against
  * the native libraries in this image that ARE that third-party code: liblz4 1.9.3 (what
    lz4-java's JNI instance binds), zlib (the definition java.util.zip.{CRC32,Adler32} wrap),
    python-xxhash (xxHash reference implementation), libsnappy 1.1.8;
  * independent decoders (pyarrow lz4_raw / snappy);
  * published known-answer vectors and hand-assembled frames of the lz4-java LZ4Block format;
  * the committed fixtures under tests/golden/ (made by tests/golden/make_golden.py).
Snappy byte-exactness vs the JVM's bundled snappy 1.1.10 stays "parity unpinned" (DESIGN.md).
"""
import ctypes
import os
import struct
import zlib

import numpy as np
import pytest

import corpus

LZ4, SNAPPY, NONE = 1, 2, 0
ADLER, CRC = 1, 2
SEED = 0x9747B28C


def _liblz4():
    try:
        L = ctypes.CDLL("liblz4.so.1")
    except OSError:
        pytest.skip("liblz4.so.1 not present")
    L.LZ4_versionString.restype = ctypes.c_char_p
    return L


def _lz4_native(L, data: np.ndarray) -> bytes:
    cap = L.LZ4_compressBound(data.size)
    out = ctypes.create_string_buffer(cap)
    n = L.LZ4_compress_default(data.ctypes.data_as(ctypes.c_char_p), out, data.size, cap)
    return out.raw[:n]


# ---- hashes / checksums -------------------------------------------------------------------------
def test_xxh32_matches_reference_implementation(oracle):
    xxhash = pytest.importorskip("xxhash")
    rng = np.random.default_rng(1)
    for n in list(range(0, 70)) + [255, 256, 1000, 32767, 32768, 100_003]:
        d = rng.integers(0, 256, n, dtype=np.uint8)
        for seed in (0, 1, SEED, 0xFFFFFFFF):
            assert oracle.xxh32(d, seed) == xxhash.xxh32(d.tobytes(), seed=seed).intdigest(), (n, seed)
    # published xxHash32 sanity vectors (xxhash repo: sanity buffer is generated, so use well-known ones)
    assert oracle.xxh32(np.zeros(0, np.uint8), 0) == 0x02CC5D05
    assert oracle.xxh32(np.zeros(0, np.uint8), SEED) == 0x8D3B42D8  # SURVEY §8c probe


def test_crc32_adler32_known_answers_and_zlib(oracle):
    assert oracle.checksum(CRC, np.frombuffer(b"123456789", np.uint8)) == 0xCBF43926
    assert oracle.checksum(ADLER, np.frombuffer(b"Wikipedia", np.uint8)) == 0x11E60398
    assert oracle.checksum(CRC, np.zeros(0, np.uint8)) == 0      # java.util.zip.CRC32 of no bytes
    assert oracle.checksum(ADLER, np.zeros(0, np.uint8)) == 1    # java.util.zip.Adler32 of no bytes
    rng = np.random.default_rng(2)
    for n in [1, 2, 3, 7, 8, 9, 15, 16, 17, 63, 64, 65, 5551, 5552, 5553, 65521, 1 << 20, 3_000_001]:
        d = rng.integers(0, 256, n, dtype=np.uint8)
        assert oracle.checksum(CRC, d) == zlib.crc32(d.tobytes()), n
        assert oracle.checksum(ADLER, d) == zlib.adler32(d.tobytes()), n
    ff = np.full(1 << 20, 255, np.uint8)  # Adler32 worst case for deferred modulo
    assert oracle.checksum(ADLER, ff) == zlib.adler32(ff.tobytes())


def test_crc32c_rfc3720_vectors_and_the_processor_instruction(oracle):
    """java.util.zip.CRC32C (Castagnoli; ABI 8's S3S_CHECKSUM_CRC32C).  No library of this image computes it from Python, so
    the pins are: the check value, the four vectors of RFC 3720 B.4 (iSCSI), and the x86 crc32 instruction - the polynomial
    in silicon, which is also what HotSpot's CRC32C intrinsic executes - on lengths around every boundary of both forms;
    the running form (init = previous value) must chain."""
    u8 = lambda b: np.frombuffer(b, np.uint8)  # noqa: E731
    assert oracle.crc32c(u8(b"123456789")) == 0xE3069283
    assert oracle.crc32c(np.zeros(32, np.uint8)) == 0x8A9136AA
    assert oracle.crc32c(np.full(32, 255, np.uint8)) == 0x62A8AB43
    assert oracle.crc32c(np.arange(32, dtype=np.uint8)) == 0x46DD794E
    assert oracle.crc32c(np.arange(31, -1, -1, dtype=np.uint8)) == 0x113FDB5C
    assert oracle.checksum(3, np.zeros(0, np.uint8)) == 0
    if not oracle.crc32c_hw_available():
        pytest.skip("no SSE4.2 on this host: the instruction pin cannot run")
    rng = np.random.default_rng(12)
    for n in [1, 2, 3, 7, 8, 9, 15, 16, 17, 63, 64, 65, 4095, 4096, 4097, 65521, 1 << 20, 3_000_001]:
        d = rng.integers(0, 256, n, dtype=np.uint8)
        want = oracle.crc32c(d, hw=True)
        assert oracle.crc32c(d) == want and oracle.checksum(3, d) == want and oracle.checksum_fast(3, d) == want, n
        cut = int(rng.integers(0, n + 1))
        assert oracle.crc32c(d[cut:], init=oracle.crc32c(d[:cut])) == want


# ---- raw LZ4 block: byte-exact with liblz4 1.9.3 -------------------------------------------------
@pytest.mark.parametrize("kind", range(corpus.N_KINDS))
def test_lz4_block_bit_exact_with_liblz4(oracle, kind):
    L = _liblz4()
    assert L.LZ4_versionString() == b"1.9.3", "the pin is only meaningful against liblz4 1.9.3"
    rng = np.random.default_rng(40 + kind)
    lengths = [n for n in corpus.EDGE_LENGTHS if 0 < n <= 65536] + [12345, 20000, 32000]
    for n in lengths:
        if kind == 6 and n > 6000:
            continue
        d = corpus.chunk_corpus(kind, n, rng)
        assert oracle.lz4_compress_block(d).tobytes() == _lz4_native(L, d), (kind, n)


def test_lz4_block_hypothesis_style_random_structures(oracle):
    """Seeded structured fuzz: mixtures of literals, short/long matches, overlapping copies."""
    L = _liblz4()
    rng = np.random.default_rng(99)
    for it in range(300):
        n = int(rng.integers(1, 40000))
        alphabet = int(rng.choice([2, 4, 16, 64, 256]))
        d = rng.integers(0, alphabet, n, dtype=np.uint8)
        for _ in range(int(rng.integers(0, 60))):
            if n < 16:
                break
            dst = int(rng.integers(1, n - 4))
            ln = int(rng.integers(4, min(400, n - dst) + 1))
            off = int(rng.integers(1, dst + 1))
            for k in range(ln):  # overlapping copy semantics
                d[dst + k] = d[dst + k - off]
        assert oracle.lz4_compress_block(d).tobytes() == _lz4_native(L, d), it


def test_lz4_block_decodes_with_independent_decoders(oracle):
    pa = pytest.importorskip("pyarrow")
    L = _liblz4()
    rng = np.random.default_rng(5)
    codec = pa.Codec("lz4_raw")
    for kind in range(corpus.N_KINDS):
        d = corpus.chunk_corpus(kind, 5000 if kind == 6 else 32768, rng)
        c = oracle.lz4_compress_block(d)
        back = codec.decompress(c.tobytes(), decompressed_size=d.size)
        assert back.to_pybytes() == d.tobytes()
        out = ctypes.create_string_buffer(d.size)
        n = L.LZ4_decompress_safe(c.ctypes.data_as(ctypes.c_char_p), out, c.size, d.size)
        assert n == d.size and out.raw == d.tobytes()
        # and the oracle's own decoder on the library's output
        lib = oracle.lib()
        nat = np.frombuffer(_lz4_native(L, d), np.uint8)
        dec = np.empty(d.size, np.uint8)
        r = lib.s3o_lz4_decompress_block(nat.ctypes.data, nat.size, dec.ctypes.data, d.size, None)
        assert r == d.size and np.array_equal(dec, d)


# ---- LZ4Block stream framing (lz4-java 1.8.0 LZ4BlockOutputStream) -------------------------------
MAGIC = b"LZ4Block"


def test_lz4block_one_byte_partition_hand_assembled(oracle):
    xxhash = pytest.importorskip("xxhash")
    b = b"\x7f"
    check = xxhash.xxh32(b, seed=SEED).intdigest() & 0x0FFFFFFF
    want = (MAGIC + bytes([0x15]) + struct.pack("<iiI", 1, 1, check) + b +
            MAGIC + bytes([0x15]) + struct.pack("<iii", 0, 0, 0))
    got = oracle.compress_stream(LZ4, np.frombuffer(b, np.uint8))
    assert got.tobytes() == want


def test_lz4block_stream_structure(oracle):
    xxhash = pytest.importorskip("xxhash")
    L = _liblz4()
    rng = np.random.default_rng(8)
    d = np.concatenate([corpus.chunk_corpus(7, 70_000, rng), rng.integers(0, 256, 40_000, dtype=np.uint8)])
    s = oracle.compress_stream(LZ4, d).tobytes()
    pos, upos = 0, 0
    while True:
        assert s[pos:pos + 8] == MAGIC
        token = s[pos + 8]
        clen, olen, check = struct.unpack("<iiI", s[pos + 9:pos + 21])
        assert token & 0x0F == 5  # level = log2(32 KiB) - 10
        if olen == 0:
            assert token == 0x15 and clen == 0 and check == 0 and pos + 21 == len(s)
            break
        chunk = d[upos:upos + olen]
        assert olen == min(32768, d.size - upos)
        assert check == xxhash.xxh32(chunk.tobytes(), seed=SEED).intdigest() & 0x0FFFFFFF
        native = _lz4_native(L, chunk)
        if len(native) >= olen:  # LZ4BlockOutputStream: compressedLength >= len -> RAW
            assert token == 0x15 and clen == olen and s[pos + 21:pos + 21 + clen] == chunk.tobytes()
        else:
            assert token == 0x25 and s[pos + 21:pos + 21 + clen] == native
        pos += 21 + clen
        upos += olen
    assert upos == d.size
    assert oracle.compress_stream(LZ4, np.zeros(0, np.uint8)).size == 0  # never-opened stream


def test_liblz4_backed_baseline_stream_equals_restatement(oracle):
    lib = oracle.lib()
    if not lib.s3o_mt_have_liblz4():
        pytest.skip("liblz4 not loadable")
    rng = np.random.default_rng(12)
    for kind in (0, 1, 3, 7):
        d = corpus.chunk_corpus(kind, 100_000, rng)
        want = oracle.compress_stream(LZ4, d)
        out = np.empty(want.size + 64, np.uint8)
        n = lib.s3o_mt_stream_liblz4(d.ctypes.data, d.size, 32768, out.ctypes.data)
        assert n == want.size and np.array_equal(out[:n], want)


def test_libsnappy_backed_baseline_stream_equals_restatement(oracle):
    """round 4: the Snappy legs of the cpu_baseline run libsnappy itself (as the LZ4 legs run liblz4): same image as the
    restatement, and concatenated streams (a batch range) decode back through the library's decompressor"""
    lib = oracle.lib()
    if not lib.s3o_mt_have_libsnappy():
        pytest.skip("libsnappy not loadable")
    rng = np.random.default_rng(13)
    images, srcs = [], []
    for kind in (0, 1, 3, 7):
        d = corpus.chunk_corpus(kind, 100_000, rng)
        want = oracle.compress_stream(SNAPPY, d)
        out = np.empty(want.size + 64, np.uint8)
        n = lib.s3o_mt_stream_libsnappy(d.ctypes.data, d.size, 32768, out.ctypes.data)
        assert n == want.size and np.array_equal(out[:n], want)
        images.append(out[:n].copy())
        srcs.append(d)
    cat = np.concatenate(images)
    back = np.empty(sum(d.size for d in srcs), np.uint8)
    assert lib.s3o_mt_decode_libsnappy(cat.ctypes.data, cat.size, back.ctypes.data, back.size) == back.size
    assert np.array_equal(back, np.concatenate(srcs))
    cat[40] ^= 0x55
    assert lib.s3o_mt_decode_libsnappy(cat.ctypes.data, cat.size, back.ctypes.data, back.size) < 0 or not np.array_equal(back, np.concatenate(srcs))


def test_simd_checksums_of_the_baseline_equal_zlib(oracle):
    """round 4 (the CPU leg's CRC32 must run at hardware speed): the carry-less-multiply CRC32 and the SSSE3
    Adler32 the cpu_baseline uses give zlib's values for every length, alignment and running value; the folding constants
    they derive from the polynomial are the ones Intel's paper tabulates"""
    import zlib

    if oracle.lib().s3o_simd_available() & 1:
        assert oracle.simd_crc_constants() == [0x154442bd4, 0x1c6e41596, 0x1751997d0, 0x0ccaa009e, 0x163cd6124, 0x1db710641, 0x1f7011641]
    rng = np.random.default_rng(14)
    buf = rng.integers(0, 256, 1 << 19, dtype=np.uint8)
    buf[1000:40000] = 0xFF  # the largest byte sums between two modulo steps
    for t in range(1500):
        n = int(rng.integers(0, 700)) if t % 3 else int(rng.integers(0, 1 << 19))
        off = int(rng.integers(0, 67))
        b = buf[off:min(off + n, buf.size)]
        init = int(rng.integers(0, 1 << 32)) if t % 2 else 0
        assert oracle.checksum_fast(CRC, b, init) == zlib.crc32(b.tobytes(), init)
        ainit = 1 if not t % 2 else (init % 65521) | (((init >> 16) % 65521) << 16)
        assert oracle.checksum_fast(ADLER, b, ainit) == zlib.adler32(b.tobytes(), ainit)
    assert oracle.checksum_fast(CRC, np.frombuffer(b"123456789" * 20, np.uint8)) == zlib.crc32(b"123456789" * 20)
    assert oracle.checksum_fast(ADLER, np.zeros(0, np.uint8)) == 1 and oracle.checksum_fast(CRC, np.zeros(0, np.uint8)) == 0


# ---- Snappy (oracle restates libsnappy 1.1.8; JVM bundles 1.1.10: parity unpinned) ----------------
def _libsnappy():
    for name in ("libsnappy.so.1", "/opt/conda/lib/libsnappy.so.1"):
        try:
            return ctypes.CDLL(name)
        except OSError:
            continue
    pytest.skip("libsnappy not present")


def test_snappy_block_vs_libsnappy_and_pyarrow(oracle):
    S = _libsnappy()
    pa = pytest.importorskip("pyarrow")
    rng = np.random.default_rng(21)
    for kind in range(corpus.N_KINDS):
        for n in (1, 15, 16, 17, 100, 4096, 32768):
            if kind == 6 and n > 6000:
                continue
            d = corpus.chunk_corpus(kind, n, rng)
            c = oracle.snappy_compress_block(d)
            cap = ctypes.c_size_t(S.snappy_max_compressed_length(ctypes.c_size_t(n)))
            S.snappy_max_compressed_length.restype = ctypes.c_size_t
            out = ctypes.create_string_buffer(32 + n + n // 6)
            olen = ctypes.c_size_t(len(out))
            assert S.snappy_compress(d.ctypes.data_as(ctypes.c_char_p), ctypes.c_size_t(n), out, ctypes.byref(olen)) == 0
            assert c.tobytes() == out.raw[:olen.value], (kind, n)
            assert pa.Codec("snappy").decompress(c.tobytes(), decompressed_size=n).to_pybytes() == d.tobytes()


def test_snappy_stream_framing(oracle):
    rng = np.random.default_rng(22)
    d = corpus.chunk_corpus(7, 80_000, rng)
    s = oracle.compress_stream(SNAPPY, d).tobytes()
    assert s[:8] == b"\x82SNAPPY\x00" and struct.unpack(">ii", s[8:16]) == (1, 1)
    pos, upos = 16, 0
    while pos < len(s):
        (clen,) = struct.unpack(">i", s[pos:pos + 4])
        chunk = d[upos:upos + 32768]
        assert s[pos + 4:pos + 4 + clen] == oracle.snappy_compress_block(chunk).tobytes()
        pos += 4 + clen
        upos += chunk.size
    assert upos == d.size and pos == len(s)
    assert oracle.decompress_stream(SNAPPY, np.frombuffer(s, np.uint8), d.size).tobytes() == d.tobytes()


# ---- LZF (round 4: decode only on the GPU; liblzf 3.6 is the pin for the block format) --------------------
LZF = 4
HERE = os.path.dirname(os.path.abspath(__file__))
_CONDA39 = "/opt/conda/bin/python3.9"


def test_lzf_block_decoder_against_liblzf_fixtures(oracle):
    """tests/golden/lzf_liblzf.npz was written by liblzf itself (imagecodecs of the conda python3.9,
    tests/golden/make_lzf_golden.py): the oracle's block decoder gives the raw bytes back for every case, and its own
    encoder's blocks decode to the source as well (a valid stream, not compress-lzf's bytes)"""
    g = np.load(os.path.join(HERE, "golden", "lzf_liblzf.npz"))
    names = sorted(k[4:] for k in g.files if k.startswith("raw_"))
    assert len(names) >= 8
    for n in names:
        raw, enc = g["raw_" + n], g["lzf_" + n]
        got = oracle.lzf_decompress_block(enc, raw.size)
        assert not isinstance(got, int) and np.array_equal(got, raw), n
        mine = oracle.lzf_compress_block(raw)
        back = oracle.lzf_decompress_block(mine, raw.size)
        assert not isinstance(back, int) and np.array_equal(back, raw), n
    # malformed blocks: a reference in front of the block, a literal run / a reference cut off by the end
    assert oracle.lzf_decompress_block(np.array([0x20, 0x05], np.uint8), 16) == -3
    assert oracle.lzf_decompress_block(np.array([0x03, 1, 2], np.uint8), 16) == -3
    assert oracle.lzf_decompress_block(np.array([0x00, 7, 0xE0], np.uint8), 16) == -3
    assert oracle.lzf_decompress_block(np.array([0x00, 7, 0x20, 0x00], np.uint8), 2) == -2


def test_lzf_live_against_liblzf(oracle):
    """both directions against the C library, when the image's conda python3.9 (imagecodecs) is there: liblzf decodes what the
    oracle's encoder writes, the oracle decodes what liblzf writes, on fresh corpora"""
    import subprocess

    if not os.path.exists(_CONDA39):
        pytest.skip("no conda python3.9 with imagecodecs")
    script = os.path.join(HERE, "golden", "make_lzf_golden.py")
    probe = subprocess.run([_CONDA39, "-c", "import imagecodecs"], capture_output=True)
    if probe.returncode != 0:
        pytest.skip("imagecodecs not importable")
    rng = np.random.default_rng(44)
    for kind in (0, 2, 3, 6, 7):
        d = corpus.chunk_corpus(kind, 40_000 if kind != 6 else 6000, rng)
        enc = subprocess.run([_CONDA39, script, "--encode"], input=d.tobytes(), capture_output=True, check=True).stdout
        got = oracle.lzf_decompress_block(np.frombuffer(enc, np.uint8), d.size)
        assert not isinstance(got, int) and np.array_equal(got, d), kind
        mine = oracle.lzf_compress_block(d)
        back = subprocess.run([_CONDA39, script, "--decode", str(d.size)], input=mine.tobytes(), capture_output=True, check=True).stdout
        assert back == d.tobytes(), kind


def test_lzf_chunk_stream_framing(oracle):
    """compress-lzf chunks: 'Z' 'V' type | len BE [| ulen BE]; chunks of at most 65 535 bytes, stored when they do not shrink;
    concatenated streams are more chunks; the layout functions take the codec like the others"""
    rng = np.random.default_rng(45)
    d = np.concatenate([corpus.chunk_corpus(7, 100_000, rng), rng.integers(0, 256, 70_000, dtype=np.uint8)])
    offs = np.array([0, 90_000, 90_000, 150_000, d.size], np.int64)
    img, index, sums = oracle.compress_map_output(LZF, ADLER, d, offs)
    pos, seen, kinds = 0, 0, set()
    first = img[:index[1]].tobytes()
    while pos < len(first):
        assert first[pos:pos + 2] == b"ZV" and first[pos + 2] in (0, 1)
        kinds.add(first[pos + 2])
        n = struct.unpack(">H", first[pos + 3:pos + 5])[0]
        if first[pos + 2] == 1:
            u = struct.unpack(">H", first[pos + 5:pos + 7])[0]
            blk = np.frombuffer(first[pos + 7:pos + 7 + n], np.uint8)
            assert np.array_equal(oracle.lzf_decompress_block(blk, u), d[seen:seen + u])
            pos += 7 + n
            seen += u
        else:
            assert first[pos + 5:pos + 5 + n] == d[seen:seen + n].tobytes()
            pos += 5 + n
            seen += n
    assert seen == 90_000 and index[2] == index[1]
    rc, back, bad = oracle.decompress_range(LZF, ADLER, img, index, sums, d.size)
    assert rc == 0 and np.array_equal(back, d)
    # the random partition is stored chunks
    third = img[index[3]:index[4]].tobytes()
    assert third[2] == 0
    bad_img = img.copy()
    bad_img[1] = ord("X")
    assert oracle.decompress_range(LZF, 0, bad_img, index, None, d.size)[0] == -3


# ---- map-output layout (.data / .index / .checksum) ------------------------------------------------
@pytest.mark.parametrize("codec", [LZ4, SNAPPY, NONE])
@pytest.mark.parametrize("algo", [ADLER, CRC])
def test_map_output_layout(oracle, codec, algo):
    rng = np.random.default_rng(31)
    data, offsets = corpus.ragged_map_output(rng, 23, 90_000)
    img, index, sums = oracle.compress_map_output(codec, algo, data, offsets)
    n = len(offsets) - 1
    assert index[0] == 0 and index[-1] == img.size and np.all(np.diff(index) >= 0)
    for p in range(n):
        part = data[offsets[p]:offsets[p + 1]]
        stream = img[index[p]:index[p + 1]]
        if codec == NONE:
            want = part
        else:
            want = oracle.compress_stream(codec, part)
        assert np.array_equal(stream, want), p
        ref = zlib.adler32(stream.tobytes()) if algo == ADLER else zlib.crc32(stream.tobytes())
        assert sums[p] == ref
        if part.size == 0:
            assert stream.size == 0 and sums[p] == (1 if algo == ADLER else 0)
    # .index / .checksum images: big-endian longs (S3ShuffleHelper.writeArrayAsBlock)
    assert oracle.longs_to_be(index) == struct.pack(f">{n + 1}q", *index.tolist())
    assert oracle.longs_to_be(sums) == struct.pack(f">{n}q", *sums.tolist())


def test_reduce_side_verify_and_batch_ranges(oracle):
    rng = np.random.default_rng(32)
    data, offsets = corpus.ragged_map_output(rng, 12, 80_000)
    img, index, sums = oracle.compress_map_output(LZ4, ADLER, data, offsets)
    rc, out, bad = oracle.decompress_range(LZ4, ADLER, img, index, sums, data.size)
    assert rc == 0 and bad == -1 and np.array_equal(out, data)
    r0, r1 = 3, 9
    sub = img[index[r0]:index[r1]]
    rc, out, bad = oracle.decompress_range(LZ4, ADLER, sub, index[r0:r1 + 1] - index[r0], sums[r0:r1],
                                           int(offsets[r1] - offsets[r0]))
    assert rc == 0 and np.array_equal(out, data[offsets[r0]:offsets[r1]])
    nonempty = [p for p in range(r0, r1) if index[p + 1] > index[p]]
    corrupt = sub.copy()
    victim = nonempty[len(nonempty) // 2]
    corrupt[index[victim] - index[r0] + 25] ^= 0x40
    rc, out, bad = oracle.decompress_range(LZ4, ADLER, corrupt, index[r0:r1 + 1] - index[r0], sums[r0:r1],
                                           int(offsets[r1] - offsets[r0]))
    assert rc == oracle.E_CHECKSUM and bad == victim - r0


# ---- committed golden fixtures ---------------------------------------------------------------------
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_golden_fixtures_reproduce(oracle):
    import json

    import golden.make_golden as mg

    manifest = json.load(open(os.path.join(GOLDEN, "manifest.json")))
    assert manifest["cases"], "no golden cases committed"
    for case in manifest["cases"]:
        data, offsets = mg.case_input(case)
        img, index, sums = oracle.compress_map_output(case["codec"], case["checksum"], data, offsets)
        blob = open(os.path.join(GOLDEN, case["name"] + ".bin"), "rb").read()
        want_img, want_index, want_sums = mg.split_blob(blob, len(offsets) - 1)
        assert oracle.longs_to_be(index) == want_index, case["name"]
        assert oracle.longs_to_be(sums) == want_sums, case["name"]
        assert img.tobytes() == want_img, case["name"]
        assert case["sha256"] == mg.sha256(blob)
