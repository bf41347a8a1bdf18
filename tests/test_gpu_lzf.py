"""GPU parity of the LZF reduce side (S3S_CODEC_LZF; round 4): map outputs whose partitions are LZFCompressionCodec
streams - compress-lzf chunks ('Z' 'V' type | len ...) around liblzf blocks, written here by the oracle's encoder and, for the
block format's pin, by liblzf itself (tests/golden/lzf_liblzf.npz) - come back byte for byte through s3s_decompress_range and the
batched entry points, per-partition checksums verified first; damaged streams are refused inside their buffers; compression
with this codec is refused (it stays on the JVM: compress-lzf's output is not a function of the partition's bytes)."""
import os
import struct

import numpy as np
import pytest

import corpus

pytestmark = pytest.mark.gpu

LZF = 4
ADLER, CRC = 1, 2
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("algo", [ADLER, CRC, 0])
def test_lzf_map_outputs_decode_to_the_source(gpu_codec, oracle, algo):
    from s3shuffle import datagen

    for data, offs in (datagen.terasort_map_output(6 << 20, 60, seed=2), datagen.tpcds_wide_map_output(3 << 20, 17, seed=3),
                       datagen.kv_int_map_output(200_000, 5, seed=1), datagen.skew_block(1 << 20, "zeros", seed=5),
                       datagen.skew_block(300_000, "random", seed=5)):
        img, index, sums = oracle.compress_map_output(LZF, algo, data, offs)
        assert gpu_codec.decompressed_size(LZF, img) == data.size
        out = gpu_codec.decompress_range(LZF, algo, img, index, sums if algo else None, dst_capacity=data.size)
        assert np.array_equal(out, data)
        n = len(offs) - 1
        if n > 3:  # a ShuffleBlockBatchId-style sub-range
            r0, r1 = n // 3, n - 1
            sub = img[index[r0]:index[r1]]
            out = gpu_codec.decompress_range(LZF, algo, sub, index[r0:r1 + 1] - index[r0], None if algo == 0 else sums[r0:r1],
                                             dst_capacity=int(offs[r1] - offs[r0]))
            assert np.array_equal(out, data[offs[r0]:offs[r1]])


def test_chunks_written_by_liblzf(gpu_codec):
    """blocks encoded by liblzf 3.6 (the C library; fixtures made by tests/golden/make_lzf_golden.py) inside chunk headers
    assembled by hand: a partition per case, compressed chunks and one stored chunk"""
    import zlib

    g = np.load(os.path.join(HERE, "golden", "lzf_liblzf.npz"))
    names = sorted(k[4:] for k in g.files if k.startswith("raw_"))
    parts, raws = [], []
    for n in names:
        raw, enc = g["raw_" + n], g["lzf_" + n]
        parts.append(b"ZV\x01" + struct.pack(">HH", enc.size, raw.size) + enc.tobytes())
        raws.append(raw)
    stored = np.random.default_rng(5).integers(0, 256, 4000, dtype=np.uint8)
    parts.append(b"ZV\x00" + struct.pack(">H", stored.size) + stored.tobytes() + parts[0])  # two chunks in one partition
    raws.append(np.concatenate([stored, raws[0]]))
    img = np.frombuffer(b"".join(parts), np.uint8)
    index = np.concatenate([[0], np.cumsum([len(p) for p in parts])]).astype(np.int64)
    sums = np.array([zlib.crc32(p) for p in parts], np.int64)
    want = np.concatenate(raws)
    out = gpu_codec.decompress_range(LZF, CRC, img, index, sums, dst_capacity=want.size)
    assert np.array_equal(out, want)


def test_lzf_batched_ranges_device_and_host(gpu_codec, oracle):
    from hipdev import Dev
    from s3shuffle import datagen

    tasks = [datagen.terasort_map_output(2 << 20, 30, seed=7, map_id=m) for m in range(3)] + [datagen.tpcds_wide_map_output(1 << 20, 9, seed=8)]
    imgs = [oracle.compress_map_output(LZF, CRC, d, o) for d, o in tasks]
    dev = Dev()
    try:
        args, outs = [], []
        for (d, o), (img, index, sums) in zip(tasks, imgs):
            d_out = dev.upload(np.full(d.size + 16, 0xA5, np.uint8))
            outs.append(d_out)
            args.append((dev.upload(img), img.size, index, sums, d_out, d.size))
        res = gpu_codec.decompress_ranges_batch_device(LZF, CRC, args)
        for (d, o), (st, n, bad), d_out in zip(tasks, res, outs):
            back = dev.download(d_out, d.size + 16)
            assert st == 0 and n == d.size and np.array_equal(back[:n], d) and np.all(back[n:] == 0xA5)
    finally:
        dev.free()
    houts = [np.zeros(d.size, np.uint8) for d, _ in tasks]
    keep = [np.ascontiguousarray(i[0]) for i in imgs]
    res = gpu_codec.decompress_ranges_batch(LZF, CRC, [(k.ctypes.data, k.size, i[1], i[2], o.ctypes.data, o.size) for k, i, o in zip(keep, imgs, houts)])
    for (d, o), (st, n, bad), out in zip(tasks, res, houts):
        assert st == 0 and n == d.size and np.array_equal(out, d)


def test_lzf_damage_capacity_and_refused_compression(gpu_codec, oracle):
    import s3shuffle
    from s3shuffle import datagen

    data, offs = datagen.terasort_map_output(1 << 20, 8, seed=4)
    img, index, sums = oracle.compress_map_output(LZF, ADLER, data, offs)
    bad = img.copy()
    bad[index[3] + 40] ^= 0x10
    with pytest.raises(s3shuffle.CodecError) as e:
        gpu_codec.decompress_range(LZF, ADLER, bad, index, sums, dst_capacity=data.size)
    assert e.value.code == -4 and e.value.partition == 3
    # checksums off: the chunk chain or the block decoder has to object (or the bytes differ - LZF has no block hash)
    broken = img.copy()
    broken[index[2] + 1] = ord("X")  # the chunk magic
    with pytest.raises(s3shuffle.CodecError) as e:
        gpu_codec.decompress_range(LZF, 0, broken, index, None, dst_capacity=data.size)
    assert e.value.code == -3
    cut = img[:index[-1] - 5]
    idx = index.copy()
    idx[-1] -= 5
    with pytest.raises(s3shuffle.CodecError) as e:
        gpu_codec.decompress_range(LZF, 0, cut, idx, None, dst_capacity=data.size)
    assert e.value.code == -3
    with pytest.raises(s3shuffle.CodecError) as e:
        gpu_codec.decompress_range(LZF, ADLER, img, index, sums, dst_capacity=data.size - 1)
    assert e.value.code == -2
    with pytest.raises(s3shuffle.CodecError) as e:
        gpu_codec.compress_map_output(LZF, ADLER, data, offs)
    assert e.value.code in (-6, -1)  # (the sizing helper refuses first: S3S_E_INVALID; the entry points answer S3S_E_UNSUPPORTED)
