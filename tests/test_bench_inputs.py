"""The inputs bench.py BUILDS for its decode-only lines (host side, no GPU): images written by the third-party libraries the
JVM writers wrap — liblz4 for LZ4Block frames of a larger block size, libzstd for `ZStdCompressionCodec` streams, liblzf
(through the image's conda interpreter) for `LZFOutputStream` chunks — must be what the oracle's decoders and the libraries'
own decoders turn back into the source, with per-partition checksums over the COMPRESSED bytes and a consistent index.
A wrong benchmark input would make a decode line measure something else than it says."""
import os
import sys
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "spark-s3-shuffle_amd"))


@pytest.fixture(scope="module")
def map_output():
    from s3shuffle import datagen

    return datagen.terasort_map_output(3 << 20, 23, seed=2, map_id=1)


def _check_layout(img, index, sums, nparts, fn):
    assert index[0] == 0 and index[-1] == img.size and len(index) == nparts + 1 and (np.diff(index) >= 0).all()
    for p in range(nparts):
        assert int(sums[p]) == fn(img[index[p]:index[p + 1]].tobytes()) & 0xFFFFFFFF


def test_lzf_image_is_what_compress_lzf_writes(map_output, oracle):
    import bench

    if not os.path.exists("/opt/conda/bin/python3.9"):
        pytest.skip("no conda python3.9 (liblzf is reachable only through its imagecodecs)")
    data, offs = map_output
    img, index, sums = bench.lzf_map_output_image(data, offs, "adler32")
    _check_layout(img, index, sums, len(offs) - 1, zlib.adler32)
    assert bytes(img[:2]) == b"ZV"
    # the oracle's chunk walk + liblzf block decoder (pinned against liblzf both ways, tests/test_oracle_pins.py)
    s, n = oracle.mt_decompress_bench(oracle.CODEC_LZF, oracle.CHECKSUM_ADLER32, img, index, sums, data.size, 1, reps=1, use_liblz4=False)
    assert s >= 0 and n == data.size


def test_zstd_image_decodes_with_libzstd(map_output):
    import bench

    data, offs = map_output
    img, index, sums = bench.zstd_map_output_image(data, offs, "crc32")
    _check_layout(img, index, sums, len(offs) - 1, zlib.crc32)
    z = bench._libzstd()
    out = np.empty(data.size, np.uint8)
    for p in range(len(offs) - 1):
        a, b = int(index[p]), int(index[p + 1])
        if b > a:
            r = z.ZSTD_decompress(out.ctypes.data + int(offs[p]), int(offs[p + 1] - offs[p]), img.ctypes.data + a, b - a)
            assert not z.ZSTD_isError(r) and r == offs[p + 1] - offs[p]
    assert np.array_equal(out, data)


def test_jvm_lz4_image_of_256k_blocks_decodes_with_the_oracle(map_output, oracle):
    import bench

    data, offs = map_output
    img, index, sums = bench.jvm_lz4_map_output_image(data, offs, "adler32", 262144)
    _check_layout(img, index, sums, len(offs) - 1, zlib.adler32)
    assert bytes(img[:8]) == b"LZ4Block" and img[8] == (0x20 | 8) or img[8] == (0x10 | 8)  # level = log2(256 KiB) - 10
    rc, back, bad = oracle.decompress_range(oracle.CODEC_LZ4, oracle.CHECKSUM_ADLER32, img, index, sums, data.size)
    if rc == 0:  # (the oracle's reader takes any LZ4Block block size: the frame header carries it)
        assert np.array_equal(back, data)
    else:        # ... or refuses frames above its own 32 KiB blocks: then liblz4 itself decodes the payloads
        import ctypes
        import struct

        L = ctypes.CDLL("liblz4.so.1")
        L.LZ4_decompress_safe.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        raw, pos, out = img.tobytes(), 0, bytearray()
        while pos < len(raw):
            assert raw[pos:pos + 8] == b"LZ4Block"
            token = raw[pos + 8]
            clen, olen, _chk = struct.unpack_from("<iiI", raw, pos + 9)
            body = raw[pos + 21:pos + 21 + clen]
            if olen:
                if token & 0xF0 == 0x10:
                    out += body
                else:
                    buf = ctypes.create_string_buffer(olen)
                    assert L.LZ4_decompress_safe(body, buf, clen, olen) == olen
                    out += buf.raw
            pos += 21 + clen
        assert bytes(out) == data.tobytes()


def test_every_secondary_line_names_a_defined_workload():
    import bench

    for label, workload, direction, mib, maps in bench.SECONDARY:
        assert workload in bench.WORKLOADS and direction in ("compress", "decompress") and mib > 0 and maps > 0
        gen, nparts, codec, algo = bench.WORKLOADS[workload]
        if codec in ("zstd", "lzf") or workload in bench.JVM_LZ4_BLOCK:
            assert direction == "decompress", label  # decode-only codecs / block sizes
