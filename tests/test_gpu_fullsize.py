"""BASELINE.json sizes on the GPU: the WHOLE `.data` image, the index and the checksums of every full-size
configuration are compared with the oracle (it needs 0.16 s per 128 MiB map task, 1.3 s for the 1 GiB block), and
the size-independent properties ride along — compress -> verify+decompress round trip, index consistency,
checksum-of-output re-derived by the checksum kernel, idempotence (same bytes on a second context)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LZ4, SNAPPY = 1, 2
ADLER, CRC = 1, 2


def _roundtrip(gpu_codec, oracle, codec, algo, data, offsets):
    """Host-buffer entry points on purpose (no torch in the test process: the library brings its own
    HIP runtime binding and must be the only one initialised)."""
    import s3shuffle

    img, index, sums = gpu_codec.compress_map_output(codec, algo, data, offsets)
    total = img.size
    # the oracle on the same input: every byte of the image, every index entry, every checksum
    want_img, want_index, want_sums = oracle.compress_map_output(codec, algo, data, offsets)
    assert np.array_equal(index, want_index), "index differs from the oracle"
    assert np.array_equal(sums, want_sums), "checksums differ from the oracle"
    assert img.size == want_img.size and np.array_equal(img, want_img), ".data image differs from the oracle"
    del want_img
    assert index[0] == 0 and index[-1] == total and (np.diff(index) >= 0).all()
    empty = np.diff(np.asarray(offsets)) == 0
    assert (np.diff(index)[empty] == 0).all()                      # empty partition = 0 bytes
    # checksums over the compressed bytes, recomputed by the checksum-only entry point
    again = gpu_codec.checksum_ranges(algo, img, index)
    assert np.array_equal(again, sums)
    # idempotence: a second context produces the same image
    with s3shuffle.Codec(0) as other:
        img2, index2, sums2 = other.compress_map_output(codec, algo, data, offsets)
        assert np.array_equal(index2, index) and np.array_equal(sums2, sums) and np.array_equal(img2, img)
        del img2
    # reduce side: whole range (batch fetch of every partition) round-trips
    out = gpu_codec.decompress_range(codec, algo, img, index, sums, dst_capacity=data.size)
    assert out.size == data.size and np.array_equal(out, data)
    del out
    return total


def test_terasort_128mib_200_partitions_lz4(gpu_codec, oracle):
    from s3shuffle import datagen

    data, offs = datagen.terasort_map_output(128 << 20, 200, seed=2, map_id=1)
    total = _roundtrip(gpu_codec, oracle, LZ4, ADLER, data, offs)
    assert 3.5 < data.size / total < 5.5


def test_terasort_128mib_2000_partitions_lz4_crc32(gpu_codec, oracle):
    from s3shuffle import datagen

    data, offs = datagen.terasort_map_output(128 << 20, 2000, seed=4, map_id=5)
    _roundtrip(gpu_codec, oracle, LZ4, CRC, data, offs)


def test_skew_1gib_single_partition_lz4(gpu_codec, oracle):
    from s3shuffle import datagen

    data, offs = datagen.skew_block(1 << 30, "terasort", seed=5)
    _roundtrip(gpu_codec, oracle, LZ4, ADLER, data, offs)


def test_skew_extremes_256mib(gpu_codec, oracle):
    from s3shuffle import datagen

    for kind in ("zeros", "random"):
        data, offs = datagen.skew_block(256 << 20, kind, seed=5)
        total = _roundtrip(gpu_codec, oracle, LZ4, CRC, data, offs)
        if kind == "random":
            assert total == data.size + 21 * (data.size // 32768) + 21   # every frame stored RAW
        else:
            assert total < data.size // 200


def test_tpcds_wide_128mib_snappy(gpu_codec, oracle):
    from s3shuffle import datagen

    data, offs = datagen.tpcds_wide_map_output(128 << 20, 200, seed=3)
    _roundtrip(gpu_codec, oracle, SNAPPY, ADLER, data, offs)


def test_tpcds_wide_128mib_lz4(gpu_codec, oracle):
    """configs[2] rows under Spark's default codec (match-dense: ~6.5 sequences per 64-byte window)."""
    from s3shuffle import datagen

    data, offs = datagen.tpcds_wide_map_output(128 << 20, 200, seed=3, map_id=2)
    _roundtrip(gpu_codec, oracle, LZ4, CRC, data, offs)


# ---- reduce side only codecs at BASELINE sizes (round 5, VERDICT r4 "what's weak" 2): the configurations bench.py times —
# 128 MiB map outputs of 200 and of 2 000 partition frames — had a byte compare only at <= 12 MiB.  The images are written
# by the third-party libraries (libzstd as zstd-jni drives it; liblzf through oracle's chunk writer), decoded by the batched
# entry point bench.py uses, and compared with the source byte for byte.
ZSTD, LZF = 3, 4


@pytest.mark.parametrize("nparts", [200, 2000])
def test_terasort_128mib_zstd_frames_decode_to_the_source(gpu_codec, nparts):
    from oracle import zstd_ref as z
    from s3shuffle import datagen

    data, offs = datagen.terasort_map_output(128 << 20, nparts, seed=2, map_id=3)
    img, index, sums = z.compress_map_output(ADLER, data, offs, 1)
    out = gpu_codec.decompress_range(ZSTD, ADLER, img, index, sums, dst_capacity=data.size)
    assert out.size == data.size and np.array_equal(out, data)
    del out
    # two map outputs in one batched call (what the driver's bench line does), the second one a sub-range
    r0, r1 = nparts // 4, nparts - 3
    whole, sub = np.ascontiguousarray(img), np.ascontiguousarray(img[index[r0]:index[r1]])
    out_a, out_b = np.zeros(data.size, np.uint8), np.zeros(int(offs[r1] - offs[r0]), np.uint8)
    res = gpu_codec.decompress_ranges_batch(ZSTD, ADLER, [
        (whole.ctypes.data, whole.size, index, sums, out_a.ctypes.data, out_a.size),
        (sub.ctypes.data, sub.size, index[r0:r1 + 1] - index[r0], sums[r0:r1], out_b.ctypes.data, out_b.size)])
    assert [(st, n) for st, n, _ in res] == [(0, out_a.size), (0, out_b.size)]
    assert np.array_equal(out_a, data) and np.array_equal(out_b, data[offs[r0]:offs[r1]])


def test_wide_rows_128mib_zstd_frames_decode_to_the_source(gpu_codec):
    from oracle import zstd_ref as z
    from s3shuffle import datagen

    data, offs = datagen.tpcds_wide_map_output(128 << 20, 200, seed=3, map_id=2)
    img, index, sums = z.compress_map_output(CRC, data, offs, 1)
    out = gpu_codec.decompress_range(ZSTD, CRC, img, index, sums, dst_capacity=data.size)
    assert out.size == data.size and np.array_equal(out, data)


def test_skew_1gib_single_zstd_frame_takes_the_two_pass_form(gpu_codec):
    """BASELINE config 5 under zstd: ONE 1 GiB partition = one frame of 8 192 blocks.  The single-pass form would guess 8 x its
    compressed size of scratch: over the budget, so the call takes the two-pass form (sizes, then bytes straight into the
    caller's buffer) — the documented fallback; slow (one wavefront walks the frame) but correct."""
    from oracle import zstd_ref as z
    from s3shuffle import datagen

    data, offs = datagen.skew_block(1 << 30, "terasort", seed=5)
    img, index, sums = z.compress_map_output(ADLER, data, offs, 1)
    out = gpu_codec.decompress_range(ZSTD, ADLER, img, index, sums, dst_capacity=data.size)
    assert out.size == data.size and np.array_equal(out, data)


@pytest.mark.parametrize("nparts", [200, 2000])
def test_terasort_128mib_lzf_streams_decode_to_the_source(gpu_codec, oracle, nparts):
    from s3shuffle import datagen

    data, offs = datagen.terasort_map_output(128 << 20, nparts, seed=2, map_id=4)
    img, index, sums = oracle.compress_map_output(LZF, ADLER, data, offs)
    out = gpu_codec.decompress_range(LZF, ADLER, img, index, sums, dst_capacity=data.size)
    assert out.size == data.size and np.array_equal(out, data)


def test_terasort_128mib_lzf_streams_written_by_liblzf(gpu_codec):
    """the same with liblzf 3.6 itself as the writer (conda python3.9 + imagecodecs, one interpreter start per map output)"""
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from s3shuffle import datagen

    if not os.path.exists("/opt/conda/bin/python3.9"):
        pytest.skip("no conda python3.9 (liblzf is reachable only through its imagecodecs)")
    data, offs = datagen.terasort_map_output(128 << 20, 200, seed=2, map_id=6)
    img, index, sums = bench.lzf_map_output_image(data, offs, "adler32")
    out = gpu_codec.decompress_range(LZF, ADLER, img, index, sums, dst_capacity=data.size)
    assert out.size == data.size and np.array_equal(out, data)
