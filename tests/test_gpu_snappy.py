"""GPU parity, Snappy leg: s3s_compress_map_output / s3s_decompress_range with
S3S_CODEC_SNAPPY against the oracle's restatement of SnappyOutputStream + snappy 1.1.8 (byte-exact
vs libsnappy 1.1.8; "parity unpinned" vs the JVM's bundled 1.1.10)."""
import numpy as np
import pytest

import corpus

pytestmark = pytest.mark.gpu

SNAPPY = 2
ADLER, CRC = 1, 2


@pytest.fixture(params=[(3, 0), (4, 1)], ids=["general+ring-valu", "window+batch-decoder"], autouse=True)
def variants(request, gpu_codec):
    """Every test runs against both Snappy decoders (S3S_OPT_LZ4_DECODE_VARIANT: 3 ring / 4 batch) and
    both compressor paths (S3S_OPT_SNAPPY_VARIANT: 0 general batch only / 1 exact windows first)."""
    d_dec, d_cmp = gpu_codec.get_option(5), gpu_codec.get_option(6)
    gpu_codec.set_option(5, request.param[0])
    gpu_codec.set_option(6, request.param[1])
    yield request.param
    gpu_codec.set_option(5, d_dec)
    gpu_codec.set_option(6, d_cmp)


def _check(gpu_codec, oracle, algo, data, offsets, block_size=32768):
    img, index, sums = gpu_codec.compress_map_output(SNAPPY, algo, data, offsets)
    r_img, r_index, r_sums = oracle.compress_map_output(SNAPPY, algo, data, offsets, block_size)
    assert np.array_equal(index, r_index), (index[:8], r_index[:8])
    if algo:
        assert np.array_equal(sums, r_sums)
    assert img.size == r_img.size
    if not np.array_equal(img, r_img):
        first = int(np.nonzero(img != r_img)[0][0])
        part = int(np.searchsorted(index, first, side="right") - 1)
        raise AssertionError(f"image differs at byte {first} (partition {part}, +{first - index[part]})")
    return img, index, sums


@pytest.mark.parametrize("kind", range(corpus.N_KINDS))
def test_snappy_single_partition_edge_lengths(gpu_codec, oracle, kind):
    rng = np.random.default_rng(400 + kind)
    for n in corpus.EDGE_LENGTHS:
        if kind == 6 and n > 6000:
            continue
        data = corpus.chunk_corpus(kind, n, rng)
        img, index, sums = _check(gpu_codec, oracle, ADLER, data, [0, n])
        back = gpu_codec.decompress_range(SNAPPY, ADLER, img, index, sums)
        assert np.array_equal(back, data), (kind, n)


@pytest.mark.parametrize("algo", [ADLER, CRC, 0])
def test_snappy_ragged_partitions_and_batch_ranges(gpu_codec, oracle, algo):
    rng = np.random.default_rng(17 + algo)
    for it in range(5):
        data, offsets = corpus.ragged_map_output(rng, n_parts=int(rng.integers(1, 30)), max_len=120_000)
        img, index, sums = _check(gpu_codec, oracle, algo, data, offsets)
        n = len(offsets) - 1
        r0 = int(rng.integers(0, n))
        r1 = int(rng.integers(r0 + 1, n + 1))
        sub = img[index[r0]:index[r1]]
        out = gpu_codec.decompress_range(SNAPPY, algo, sub, index[r0:r1 + 1] - index[r0],
                                         None if algo == 0 else sums[r0:r1])
        assert np.array_equal(out, data[offsets[r0]:offsets[r1]])


def test_snappy_workloads_and_golden(gpu_codec, oracle):
    import json
    import os

    import golden.make_golden as mg
    from s3shuffle import datagen

    d, o = datagen.tpcds_wide_map_output(4 << 20, 200, seed=3)
    img, index, sums = _check(gpu_codec, oracle, ADLER, d, o)
    assert np.array_equal(gpu_codec.decompress_range(SNAPPY, ADLER, img, index, sums), d)
    d, o = datagen.terasort_map_output(2 << 20, 50, seed=2)
    _check(gpu_codec, oracle, CRC, d, o)
    for kind in ("zeros", "random"):
        d, o = datagen.skew_block(1 << 20, kind, seed=5)
        img, index, sums = _check(gpu_codec, oracle, CRC, d, o)
        assert np.array_equal(gpu_codec.decompress_range(SNAPPY, CRC, img, index, sums), d)
    gdir = os.path.dirname(mg.__file__)
    case = [c for c in json.load(open(os.path.join(gdir, "manifest.json")))["cases"] if c["codec"] == SNAPPY][0]
    data, offsets = mg.case_input(case)
    img, index, sums = gpu_codec.compress_map_output(SNAPPY, case["checksum"], data, offsets)
    want_img, want_index, want_sums = mg.split_blob(open(os.path.join(gdir, case["name"] + ".bin"), "rb").read(), len(offsets) - 1)
    assert img.tobytes() == want_img and oracle.longs_to_be(index) == want_index and oracle.longs_to_be(sums) == want_sums


def test_snappy_decode_errors_and_concatenated_streams(gpu_codec, oracle):
    import s3shuffle

    rng = np.random.default_rng(5)
    a = corpus.chunk_corpus(7, 70_000, rng)
    b = corpus.chunk_corpus(3, 40_000, rng)
    sa, sb = oracle.compress_stream(SNAPPY, a), oracle.compress_stream(SNAPPY, b)
    # one partition made of two complete streams (multi-spill merge)
    cat = np.concatenate([sa, sb])
    out = gpu_codec.decompress_range(SNAPPY, 0, cat, [0, cat.size], None)
    assert np.array_equal(out, np.concatenate([a, b]))
    assert gpu_codec.decompressed_size(SNAPPY, cat) == a.size + b.size
    bad = sa.copy()
    bad[3] ^= 0xFF  # stream header
    with pytest.raises(s3shuffle.CodecError) as ei:
        gpu_codec.decompress_range(SNAPPY, 0, bad, [0, bad.size], None, dst_capacity=a.size)
    assert ei.value.code == -3
    bad = sa.copy()
    bad[16 + 4 + 2] ^= 0x02  # varint preamble of the first block says 0 bytes: length mismatch
    with pytest.raises(s3shuffle.CodecError) as ei:
        gpu_codec.decompress_range(SNAPPY, 0, bad, [0, bad.size], None, dst_capacity=a.size + 70_000)
    assert ei.value.code == -3
    sums = np.array([oracle.checksum(ADLER, sa)], np.int64)
    with pytest.raises(s3shuffle.CodecError) as ei:
        gpu_codec.decompress_range(SNAPPY, ADLER, sa, [0, sa.size], sums + 1)
    assert ei.value.code == -4 and ei.value.partition == 0


def test_snappy_block_size_option(gpu_codec, oracle):
    rng = np.random.default_rng(23)
    data, offsets = corpus.ragged_map_output(rng, 7, 40_000)
    for bs in (512, 1024, 4096, 16384):
        gpu_codec.set_option(2, bs)
        try:
            eff = max(bs, 1024)  # snappy-java: Math.max(MIN_BLOCK_SIZE, blockSize)
            _check(gpu_codec, oracle, ADLER, data, offsets, block_size=eff)
        finally:
            gpu_codec.set_option(2, 32768)


def test_snappy_every_element_shape_of_the_deferred_flush(gpu_codec, oracle):
    """literal runs 0 .. 20 (up to 12 wait as records, longer ones go out at once behind a flush) x two- and three-byte copies:
    every byte count / tail position a lane of the Snappy window block's flush writes (snappy_window_engine.inc .Ls_flush)"""
    rng = np.random.default_rng(54)
    n = 3 * 32768 + 777
    data = corpus.planted_sequence_shapes(rng, n, range(0, 21), (4, 6, 11, 12, 13, 30, 64, 65, 90))
    _check(gpu_codec, oracle, 1, data, [0, 33333, 33333, n])
