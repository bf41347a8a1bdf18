"""GPU parity of the HOST-buffer batch entry points (s3s_compress_map_outputs_batch / s3s_decompress_ranges_batch —
what the JNI shim binds): several tasks per call, page-locked and pageable buffers, more than one pipeline group,
every task compared with the oracle byte for byte; error statuses per range."""
import numpy as np
import pytest

import corpus

pytestmark = pytest.mark.gpu

LZ4, SNAPPY, NONE = 1, 2, 0
ADLER, CRC = 1, 2


def _tasks(sizes_parts, seed):
    from s3shuffle import datagen

    out = []
    for i, (n_bytes, parts) in enumerate(sizes_parts):
        if n_bytes == 0:
            out.append((np.zeros(0, np.uint8), np.zeros(parts + 1, np.int64)))
        elif i % 2:
            out.append(datagen.tpcds_wide_map_output(n_bytes, parts, seed=seed + i, map_id=i))
        else:
            out.append(datagen.terasort_map_output(n_bytes, parts, seed=seed + i, map_id=i))
    return out


def _run_compress(codec_obj, codec, algo, tasks, pinned):
    import s3shuffle

    keep, args = [], []
    for data, offs in tasks:
        cap = codec_obj.max_compressed_size(codec, offs)
        if pinned:
            src = s3shuffle.PinnedBuffer(max(data.size, 1))
            src.array[:data.size] = data
            dst = s3shuffle.PinnedBuffer(max(cap, 1))
            keep.append((src, dst))
            args.append((src.ptr, offs, dst.ptr, cap))
            views = dst.array
        else:
            src = np.ascontiguousarray(data)
            dst = np.empty(max(cap, 1), np.uint8)
            keep.append((src, dst))
            args.append((src.ctypes.data, offs, dst.ctypes.data, cap))
    res = codec_obj.compress_map_outputs_batch(codec, algo, args)
    out = []
    for (total, index, sums), (src, dst) in zip(res, keep):
        arr = dst.array if pinned else dst
        out.append((arr[:total].copy(), index, sums))
    for src, dst in keep:
        if pinned:
            src.free()
            dst.free()
    return out


@pytest.mark.parametrize("pinned", [True, False], ids=["page-locked", "pageable"])
@pytest.mark.parametrize("codec,algo", [(LZ4, ADLER), (LZ4, CRC), (SNAPPY, ADLER), (NONE, CRC)])
def test_host_batch_compress_matches_the_oracle(gpu_codec, oracle, codec, algo, pinned):
    # six tasks, three pipeline groups (the group size is ~128 MiB of source), an empty task and a tiny one in between
    spec = [(70 << 20, 200), (50 << 20, 64), (0, 5), (90 << 20, 200), (4096, 3), (60 << 20, 17)]
    tasks = _tasks(spec, seed=40)
    got = _run_compress(gpu_codec, codec, algo, tasks, pinned)
    for (data, offs), (img, index, sums) in zip(tasks, got):
        want = oracle.compress_map_output(codec, algo, data, offs)
        assert np.array_equal(index, want[1])
        assert np.array_equal(sums, want[2])
        assert np.array_equal(img, want[0])


def test_host_batch_compress_single_task_and_capacity(gpu_codec, oracle):
    import s3shuffle

    (data, offs), = _tasks([(3 << 20, 9)], seed=50)
    (img, index, sums), = _run_compress(gpu_codec, LZ4, ADLER, [(data, offs)], True)
    want = oracle.compress_map_output(LZ4, ADLER, data, offs)
    assert np.array_equal(img, want[0]) and np.array_equal(index, want[1]) and np.array_equal(sums, want[2])
    # one task of two with a destination that is too small: its status says so, the other one is complete
    t2 = _tasks([(2 << 20, 4), (2 << 20, 4)], seed=51)
    src = [np.ascontiguousarray(d) for d, _ in t2]
    caps = [gpu_codec.max_compressed_size(LZ4, t2[0][1]), 1000]
    dst = [np.empty(c, np.uint8) for c in caps]
    with pytest.raises(s3shuffle.CodecError) as ei:
        gpu_codec.compress_map_outputs_batch(LZ4, ADLER, [(s.ctypes.data, o, d.ctypes.data, c)
                                                          for s, (_, o), d, c in zip(src, t2, dst, caps)])
    assert ei.value.code == -2


@pytest.mark.parametrize("pinned", [True, False], ids=["page-locked", "pageable"])
@pytest.mark.parametrize("codec,algo", [(LZ4, ADLER), (SNAPPY, CRC)])
def test_host_batch_decompress_matches_the_source(gpu_codec, oracle, codec, algo, pinned):
    import s3shuffle

    spec = [(120 << 20, 200), (100 << 20, 64), (0, 5), (150 << 20, 200), (4096, 3)]
    tasks = _tasks(spec, seed=60)
    keep, args = [], []
    for data, offs in tasks:
        img, index, sums = oracle.compress_map_output(codec, algo, data, offs)
        if pinned:
            comp = s3shuffle.PinnedBuffer(max(img.size, 1))
            comp.array[:img.size] = img
            dst = s3shuffle.PinnedBuffer(max(data.size, 1))
            dst.array[:] = 0xA5
            args.append((comp.ptr, img.size, index, sums, dst.ptr, data.size))
        else:
            comp = np.ascontiguousarray(img)
            dst = np.full(max(data.size, 1), 0xA5, np.uint8)
            args.append((comp.ctypes.data, img.size, index, sums, dst.ctypes.data, data.size))
        keep.append((comp, dst))
    res = gpu_codec.decompress_ranges_batch(codec, algo, args)
    for (data, _), (st, n, bad), (comp, dst) in zip(tasks, res, keep):
        assert st == 0 and n == data.size and bad == -1
        arr = dst.array if pinned else dst
        assert np.array_equal(arr[:n], data)
    if pinned:
        for comp, dst in keep:
            comp.free()
            dst.free()


def test_host_batch_decompress_reports_the_bad_range(gpu_codec, oracle):
    rng = np.random.default_rng(70)
    args, keep, victims = [], [], {}
    for r in range(4):
        data, offs = corpus.ragged_map_output(rng, 6, 300_000)
        img, index, sums = oracle.compress_map_output(LZ4, ADLER, data, offs)
        img = img.copy()
        if r == 2:
            nonempty = [p for p in range(6) if index[p + 1] > index[p]]
            victims[r] = nonempty[-1]
            img[index[victims[r]] + 25] ^= 0x20
        dst = np.zeros(data.size, np.uint8)
        keep.append((img, dst, data))
        args.append((img.ctypes.data, img.size, index, sums, dst.ctypes.data, data.size))
    res = gpu_codec.decompress_ranges_batch(LZ4, ADLER, args, raise_on_error=False)
    for r, (st, n, bad) in enumerate(res):
        if r in victims:
            assert st == -4 and bad == victims[r]
        else:
            assert st == 0 and np.array_equal(keep[r][1][:n], keep[r][2])


def test_call_level_failure_marks_every_unfinished_entry_not_run(gpu_codec, oracle):
    """ABI 6 (advisor r3): a batch call that fails as a CALL (here: an argument error the validation meets at the third
    range) returns that code and leaves S3S_STATUS_NOT_RUN in every entry — the caller applies the return code to exactly
    those — while a failure of ONE range (previous test) keeps its own verdict and its neighbours decode."""
    from s3shuffle import codec as sc

    rng = np.random.default_rng(71)
    args, keep = [], []
    for r in range(4):
        data, offs = corpus.ragged_map_output(rng, 5, 200_000)
        img, index, sums = oracle.compress_map_output(LZ4, ADLER, data, offs)
        dst = np.zeros(data.size, np.uint8)
        keep.append((img, dst, data))
        if r == 2:
            index = index.copy()
            index[-1] += 1  # part_offsets must span [0, comp_len]: S3S_E_INVALID for the call
        args.append((img.ctypes.data, img.size, index, sums, dst.ctypes.data, data.size))
    res = gpu_codec.decompress_ranges_batch(LZ4, ADLER, args, raise_on_error=False)
    assert [st for st, _, _ in res] == [sc.STATUS_NOT_RUN] * 4, res
    assert gpu_codec._lib.s3s_last_error(gpu_codec._h)  # the message names the range
    # the device form of the same call
    import hipdev

    dev = hipdev.Dev()
    try:
        dargs = [(dev.upload(np.asarray(k[0])), a[1], a[2], a[3], dev.alloc(max(a[5], 1)), a[5]) for k, a in zip(keep, args)]
        res = gpu_codec.decompress_ranges_batch_device(LZ4, ADLER, dargs, raise_on_error=False)
        assert [st for st, _, _ in res] == [sc.STATUS_NOT_RUN] * 4, res
    finally:
        dev.free()
    # map side: the library refuses a zstd compress call as a whole
    tasks = [(k[2].ctypes.data, np.array([0, k[2].size], np.int64), k[1].ctypes.data, k[1].size) for k in keep]
    arr = (sc.MapTask * len(tasks))()
    offs_keep = []
    for i, (src, offs, dst, cap) in enumerate(tasks):
        offs_keep.append(offs)
        arr[i].d_src, arr[i].src_offsets, arr[i].num_partitions = src, offs.ctypes.data_as(sc.ctypes.POINTER(sc.ctypes.c_int64)), 1
        arr[i].d_dst, arr[i].dst_capacity = dst, cap
        arr[i].out_index = np.zeros(2, np.int64).ctypes.data_as(sc.ctypes.POINTER(sc.ctypes.c_int64))
    rc = gpu_codec._lib.s3s_compress_map_outputs_batch(gpu_codec._h, 3, 0, arr, len(tasks))
    assert rc in (-1, -6) and [arr[i].status for i in range(len(tasks))] == [sc.STATUS_NOT_RUN] * len(tasks)
