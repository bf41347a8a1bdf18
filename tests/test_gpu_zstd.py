"""GPU parity of the Zstandard reduce side (S3S_CODEC_ZSTD): map outputs written the way Spark's
ZStdCompressionCodec writes them (libzstd 1.4.8 streaming frames, one per non-empty partition — oracle/zstd_ref.py)
must come back byte for byte through s3s_decompress_range / the batched entry points, with the per-partition checksum
validation of S3ChecksumValidationStream in front; other levels and frame shapes widen the coverage; damaged streams
are refused without leaving the destination; compression with this codec is refused (it stays on the JVM)."""
import numpy as np
import pytest

import corpus

pytestmark = pytest.mark.gpu

ZSTD, LZ4 = 3, 1
ADLER, CRC = 1, 2


def _image(algo, data, offs, level=1):
    from oracle import zstd_ref as z

    return z.compress_map_output(algo, data, offs, level)


@pytest.mark.parametrize("algo", [ADLER, CRC, 0])
def test_spark_style_map_outputs_decode_to_the_source(gpu_codec, algo):
    from s3shuffle import datagen

    for data, offs in (datagen.terasort_map_output(12 << 20, 200, seed=2), datagen.tpcds_wide_map_output(6 << 20, 64, seed=3),
                       datagen.kv_int_map_output(300_000, 7, seed=1), datagen.skew_block(3 << 20, "zeros", seed=5),
                       datagen.skew_block(2 << 20, "random", seed=5)):
        img, index, sums = _image(algo, data, offs)
        assert gpu_codec.decompressed_size(ZSTD, img) == data.size
        out = gpu_codec.decompress_range(ZSTD, algo, img, index, sums, dst_capacity=data.size)
        assert np.array_equal(out, data)
        n = len(offs) - 1
        if n > 3:  # a ShuffleBlockBatchId-style sub-range
            r0, r1 = n // 3, n - 2
            sub = img[index[r0]:index[r1]]
            out = gpu_codec.decompress_range(ZSTD, algo, sub, index[r0:r1 + 1] - index[r0], None if algo == 0 else sums[r0:r1],
                                             dst_capacity=int(offs[r1] - offs[r0]))
            assert np.array_equal(out, data[offs[r0]:offs[r1]])


def test_other_levels_frame_shapes_and_concatenated_streams(gpu_codec):
    from oracle import zstd_ref as z
    from s3shuffle import datagen

    rng = np.random.default_rng(3)
    a = datagen.terasort_map_output(900_000, 1, seed=5)[0]
    b = datagen.tpcds_wide_map_output(700_000, 1, seed=6)[0]
    c = corpus.chunk_corpus(2, 60_000, rng)
    skip = np.frombuffer(b"\x50\x2a\x4d\x18\x03\x00\x00\x00abc", np.uint8)
    parts = [np.concatenate([z.compress_stream(a, 1), z.compress_stream(b, 3)]),      # two spill pieces = two frames
             z.compress(c, 19), np.zeros(0, np.uint8), np.concatenate([skip, z.compress_stream(b, 9, checksum=True)]),
             z.compress_stream(a, -3), z.compress_stream(c, 5, chunk=1000, window_log=10)]
    want = [np.concatenate([a, b]), c, np.zeros(0, np.uint8), b, a, c]
    index = np.zeros(len(parts) + 1, np.int64)
    np.cumsum([p.size for p in parts], out=index[1:])
    img = np.concatenate(parts)
    out = gpu_codec.decompress_range(ZSTD, 0, img, index, None, dst_capacity=sum(w.size for w in want))
    assert np.array_equal(out, np.concatenate(want))


def test_batched_ranges_device_and_host(gpu_codec):
    import s3shuffle
    from hipdev import Dev
    from s3shuffle import datagen

    tasks = [datagen.terasort_map_output(3 << 20, 50, seed=7, map_id=m) for m in range(3)] + \
            [datagen.tpcds_wide_map_output(2 << 20, 20, seed=8, map_id=9)]
    imgs = [_image(CRC, d, o) for d, o in tasks]
    dev = Dev()
    try:
        args, outs = [], []
        for (d, o), (img, index, sums) in zip(tasks, imgs):
            d_out = dev.upload(np.full(d.size + 16, 0xA5, np.uint8))
            outs.append(d_out)
            args.append((dev.upload(img), img.size, index, sums, d_out, d.size))
        res = gpu_codec.decompress_ranges_batch_device(ZSTD, CRC, args)
        for (d, o), (st, n, bad), d_out in zip(tasks, res, outs):
            back = dev.download(d_out, d.size + 16)
            assert st == 0 and n == d.size and np.array_equal(back[:n], d) and np.all(back[n:] == 0xA5)
    finally:
        dev.free()
    hargs, houts = [], []
    for (d, o), (img, index, sums) in zip(tasks, imgs):
        out = np.zeros(d.size, np.uint8)
        houts.append(out)
        hargs.append((np.ascontiguousarray(img).ctypes.data, img.size, index, sums, out.ctypes.data, d.size))
        hargs[-1] = hargs[-1] + ()
    keep = [np.ascontiguousarray(i[0]) for i in imgs]
    hargs = [(k.ctypes.data, k.size, i[1], i[2], o.ctypes.data, o.size) for k, i, o in zip(keep, imgs, houts)]
    res = gpu_codec.decompress_ranges_batch(ZSTD, CRC, hargs)
    for (d, o), (st, n, bad), out in zip(tasks, res, houts):
        assert st == 0 and n == d.size and np.array_equal(out, d)


def test_checksum_mismatch_corruption_and_capacity(gpu_codec):
    import s3shuffle
    from s3shuffle import datagen

    data, offs = datagen.terasort_map_output(2 << 20, 12, seed=4)
    img, index, sums = _image(ADLER, data, offs)
    bad = img.copy()
    victim = 7
    bad[index[victim] + 40] ^= 0x10
    with pytest.raises(s3shuffle.CodecError) as ei:
        gpu_codec.decompress_range(ZSTD, ADLER, bad, index, sums, dst_capacity=data.size)
    assert ei.value.code == -4 and ei.value.partition == victim
    # without the partition checksums the decoder itself has to notice (or decode what libzstd decodes)
    from oracle import zstd_ref as z

    rng = np.random.default_rng(9)
    refused = 0
    for _ in range(40):
        m = img.copy()
        m[int(rng.integers(0, m.size))] ^= 1 << int(rng.integers(0, 8))
        ref = z.decompress(m[index[0]:index[1]], data.size)  # (first partition only: enough to classify)
        try:
            out = gpu_codec.decompress_range(ZSTD, 0, m, index, None, dst_capacity=data.size)
            assert out.size <= data.size
        except s3shuffle.CodecError as e:
            assert e.code in (-3, -2, -6)
            refused += 1
    assert refused > 10
    with pytest.raises(s3shuffle.CodecError) as ei:
        gpu_codec.decompress_range(ZSTD, ADLER, img, index, sums, dst_capacity=data.size - 1)
    assert ei.value.code == -2
    # truncated last frame
    with pytest.raises(s3shuffle.CodecError) as ei:
        gpu_codec.decompress_range(ZSTD, 0, img[:-3], np.append(index[:-1], index[-1] - 3), None, dst_capacity=data.size)
    assert ei.value.code == -3
    # a frame WITH a content checksum (not what Spark writes; foreign writers do): damage that still decodes - a byte of a stored
    # block - is caught by the XXH64 the decoder computes over what it decoded, as libzstd catches it
    rnd = np.random.default_rng(5).integers(0, 256, 300_000, dtype=np.uint8)
    comp = z.compress_stream(rnd, 1, checksum=True)
    idx = np.array([0, comp.size], np.int64)
    assert np.array_equal(gpu_codec.decompress_range(ZSTD, 0, comp, idx, None, dst_capacity=rnd.size), rnd)
    bad = comp.copy()
    bad[comp.size // 2] ^= 0x40
    assert z.decompress(bad, rnd.size + 64) is None
    with pytest.raises(s3shuffle.CodecError) as ei:
        gpu_codec.decompress_range(ZSTD, 0, bad, idx, None, dst_capacity=rnd.size)
    assert ei.value.code == -3


def test_compression_with_zstd_is_refused(gpu_codec):
    import s3shuffle

    data = np.zeros(1000, np.uint8)
    with pytest.raises(s3shuffle.CodecError) as ei:
        gpu_codec.compress_map_output(ZSTD, ADLER, data, [0, 1000])
    assert ei.value.code in (-6, -1)


def test_single_pass_and_its_fallback(gpu_codec):
    """round 4: partitions are decoded ONCE into a scratch area at a guessed capacity (8 x the compressed size) and moved
    back to back by a copy kernel; one partition that outgrows the guess (zeros: 1000 : 1) sends the call to the two-pass
    form.  Both forms give the same bytes, single range and batched ranges, ragged partition sizes incl. empty ones, a
    destination painted behind its end."""
    from hipdev import Dev
    from s3shuffle import datagen

    rng = np.random.default_rng(9)
    plain, poffs = corpus.ragged_map_output(rng, 23, 600_000)
    z_parts = [np.zeros(300_000, np.uint8), corpus.chunk_corpus(7, 50_000, rng), np.zeros(0, np.uint8), np.full(70_000, 65, np.uint8)]
    zdata = np.concatenate(z_parts)
    zoffs = np.concatenate([[0], np.cumsum([p.size for p in z_parts])]).astype(np.int64)
    cases = [(plain, poffs), (zdata, zoffs)]  # the second one cannot stay inside its guesses
    imgs = [_image(CRC, d, o) for d, o in cases]
    for (d, o), (img, index, sums) in zip(cases, imgs):
        assert np.array_equal(gpu_codec.decompress_range(ZSTD, CRC, img, index, sums, dst_capacity=d.size), d)
    dev = Dev()
    try:
        args, outs = [], []
        for (d, o), (img, index, sums) in zip(cases + cases[:1], imgs + imgs[:1]):
            d_out = dev.upload(np.full(d.size + 32, 0x5A, np.uint8))
            outs.append((d_out, d))
            args.append((dev.upload(img), img.size, index, sums, d_out, d.size))
        for sel in ([0, 2], [0, 1, 2]):  # all inside their guesses / one range outgrows -> whole call in two passes
            res = gpu_codec.decompress_ranges_batch_device(ZSTD, CRC, [args[i] for i in sel])
            for i, (st, n, bad) in zip(sel, res):
                back = dev.download(outs[i][0], outs[i][1].size + 32)
                assert st == 0 and n == outs[i][1].size and np.array_equal(back[:n], outs[i][1]) and np.all(back[n:] == 0x5A)
        # a destination one byte short is S3S_E_CAPACITY in the single pass too, with the size it needs
        a = args[0]
        res = gpu_codec.decompress_ranges_batch_device(ZSTD, CRC, [(a[0], a[1], a[2], a[3], a[4], plain.size - 1), args[2]], raise_on_error=False)
        assert res[0][0] == -2 and res[0][1] == plain.size and res[1][0] == 0
    finally:
        dev.free()



def test_many_huffman_blocks_per_partition(gpu_codec):
    """The literal wavefront works one block ahead of its two sequence wavefronts through two literal buffers per partition
    (zstd_partitions_kernel, LitPipe): partitions of ten and more Huffman-coded blocks exercise the hand-over and the buffer
    reuse; an odd partition count leaves the last workgroup with one partition; an empty partition and a partition of stored
    blocks (no Huffman literals at all) share workgroups with them.  Once in the single-pass form, once - a partition of
    zeros outgrows its guess - through the size pass and the decode pass."""
    from s3shuffle import datagen

    tera, toffs = datagen.terasort_map_output(6 << 20, 5, seed=21)      # 1.2 MiB = 10 blocks per partition
    wide, woffs = datagen.tpcds_wide_map_output(5 << 20, 3, seed=22)    # treeless blocks among them
    rnd = np.random.default_rng(23).integers(0, 256, 400_000, dtype=np.uint8)
    pieces = [tera[toffs[k]:toffs[k + 1]] for k in range(5)] + [wide[woffs[k]:woffs[k + 1]] for k in range(3)]
    pieces += [np.zeros(0, np.uint8), rnd, pieces[0][:700_000]]  # 11 partitions
    for extra in ([], [np.zeros(2_000_000, np.uint8)]):  # (the zeros: 2 MB from ~200 bytes, far beyond 8 x)
        parts = pieces + extra
        data = np.concatenate(parts)
        offs = np.concatenate([[0], np.cumsum([p.size for p in parts])]).astype(np.int64)
        for algo in (ADLER, 0):
            img, index, sums = _image(algo, data, offs)
            assert gpu_codec.decompressed_size(ZSTD, img) == data.size
            out = gpu_codec.decompress_range(ZSTD, algo, img, index, sums, dst_capacity=data.size)
            assert np.array_equal(out, data), (len(parts), algo)
        # the same without its first partition: the other parity of partition pairs
        sub = img[index[1]:]
        out = gpu_codec.decompress_range(ZSTD, 0, sub, index[1:] - index[1], None, dst_capacity=int(data.size - offs[1]))
        assert np.array_equal(out, data[offs[1]:])


def test_damaged_multi_block_partitions_end_on_both_sides(gpu_codec):
    """Damage anywhere in partitions of ten Huffman-coded blocks: in a literals section the LITERAL wavefront meets it first
    (its `err` reaches a sequence side that is waiting), in a sequence section the sequence side does (its `quit` releases the
    literal side).  Every call must come back: refused with "bad frame" (or, rarely, a capacity / size verdict), or decoded
    to exactly what libzstd makes of the same bytes (a Spark writer's frames carry no content checksum, so most single-bit
    damage decodes - to other bytes - in both); and its undamaged neighbours in the same workgroups decode."""
    import s3shuffle
    from oracle import zstd_ref as z
    from s3shuffle import datagen

    data, offs = datagen.terasort_map_output(7 << 20, 6, seed=41)  # 1.2 MiB = 10 blocks per partition
    img, index, _ = _image(0, data, offs)
    rng = np.random.default_rng(42)
    refused = same = 0
    for it in range(48):
        m = img.copy()
        victim = int(rng.integers(0, 6))
        lo, hi = int(index[victim]), int(index[victim + 1])
        at = lo + int(rng.integers(16 if it % 3 else 100_000, hi - lo))  # (it % 3 == 0: well behind the first block)
        if it % 4 == 3:  # eight random bytes: tables and headers rarely survive that
            m[at:at + 8] = rng.integers(0, 256, min(8, hi - at), dtype=np.uint8)
        else:
            m[at] ^= 1 << int(rng.integers(0, 8))
        ref = z.decompress(m[lo:hi], int(offs[victim + 1] - offs[victim]) + 4096)
        try:
            out = gpu_codec.decompress_range(ZSTD, 0, m, index, None, dst_capacity=data.size + 4096)
        except s3shuffle.CodecError as e:
            assert e.code in (-3, -2), (it, e.code)
            refused += 1
            continue
        assert ref is not None, it  # what the GPU decodes, libzstd decodes
        want = np.concatenate([data[:offs[victim]], ref, data[offs[victim + 1]:]])
        assert np.array_equal(out, want), it
        same += 1
    assert refused >= 5 and same >= 10, (refused, same)  # (a flipped bit in a Huffman stream is just other literals: no frame checksum)
    # and the call after all that is a clean one
    assert np.array_equal(gpu_codec.decompress_range(ZSTD, 0, img, index, None, dst_capacity=data.size), data)


def test_small_partition_declaring_128k_of_huffman_literals_is_refused_and_harmless(gpu_codec):
    """advisor r5 (high): the single pass sizes a partition's literal scratch from its COMPRESSED size.  A crafted partition
    of a few hundred bytes whose 4-stream Huffman section claims regen = 131 072 (tests/corpus.py) must come back as a bad
    frame — and the valid partitions decoded by the SAME call (its neighbours in the literal scratch) must be intact."""
    from hipdev import Dev
    from s3shuffle import datagen

    data, offs = datagen.terasort_map_output(3 << 20, 24, seed=17)  # ~128 KiB partitions: Huffman-coded literals in every block
    img, index, sums = _image(0, data, offs)
    for regen, sb in ((131072, 40), (131072, 300), (30000, 64)):
        crafted = corpus.zstd_frame_with_oversized_huffman_literals(regen, sb)
        dev = Dev()
        try:
            d_good_out = dev.upload(np.full(data.size + 16, 0xA5, np.uint8))
            d_bad_out = dev.upload(np.full(regen + 8192, 0xA5, np.uint8))
            # the crafted frame sits BETWEEN valid partitions of one range as well: [first 12 partitions | crafted | the rest]
            cut = int(index[12])
            mixed = np.concatenate([img[:cut], crafted, img[cut:]])
            m_index = np.concatenate([index[:13], index[12:] + crafted.size])
            d_mixed_out = dev.upload(np.full(data.size + regen + 8192, 0xA5, np.uint8))
            args = [(dev.upload(img), img.size, index, None, d_good_out, data.size),
                    (dev.upload(crafted), crafted.size, np.array([0, crafted.size], np.int64), None, d_bad_out, regen + 4096),
                    (dev.upload(mixed), mixed.size, m_index, None, d_mixed_out, data.size + regen + 4096)]
            res = gpu_codec.decompress_ranges_batch_device(ZSTD, 0, args, raise_on_error=False)
            assert res[0][0] == 0 and res[0][1] == data.size
            back = dev.download(d_good_out, data.size + 16)
            assert np.array_equal(back[:data.size], data) and np.all(back[data.size:] == 0xA5)
            assert res[1][0] == -3 and res[2][0] == -3
            assert np.all(dev.download(d_bad_out, regen + 8192)[regen + 4096:] == 0xA5)
        finally:
            dev.free()
    # and the context is fine afterwards
    assert np.array_equal(gpu_codec.decompress_range(ZSTD, 0, img, index, None, dst_capacity=data.size), data)
