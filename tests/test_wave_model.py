"""The lock-step CPU model of the wave64 LZ4 compressor (tests/model/lz4_wave_model.cpp — the
algorithm lz4_compress.hip runs on one wavefront) must be byte-identical with the oracle under
every same-address LDS store order the hardware might choose.  Runs on the CPU-only box."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import corpus

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "model", "lz4_wave_model.cpp")
SO = os.path.join(HERE, "model", "liblz4_wave_model.so")


@pytest.fixture(scope="module")
def model():
    if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(SRC):
        subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", SO, SRC], check=True)
    L = ctypes.CDLL(SO)
    L.lz4_wave_model_compress.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                          ctypes.c_uint64, ctypes.c_void_p]
    return L


def _run(model, d, mode, seed=1, stats=None):
    out = np.empty(d.size + 64, np.uint8)
    n = model.lz4_wave_model_compress(d.ctypes.data, d.size, out.ctypes.data, mode, seed,
                                      stats.ctypes.data if stats is not None else None)
    return None if n < 0 else out[:n]


@pytest.mark.parametrize("kind", range(corpus.N_KINDS))
def test_model_matches_oracle(model, oracle, kind):
    rng = np.random.default_rng(300 + kind)
    for n in [13, 14, 20, 64, 65, 130, 1000, 4096, 20000, 32768]:
        if kind == 6 and n > 6000:
            continue
        d = corpus.chunk_corpus(kind, n, rng)
        want = oracle.lz4_compress_block(d)
        for mode in (0, 1, 2):
            got = _run(model, d, mode, seed=n)
            if got is None:  # would not fit in len bytes -> the frame layer stores RAW
                assert want.size > d.size - 0 or want.size >= d.size, (kind, n, mode)
                continue
            assert np.array_equal(got, want), (kind, n, mode)


def test_model_stats_on_terasort(model, oracle):
    from s3shuffle import datagen

    d, _ = datagen.skew_block(32768, "terasort", seed=5)
    stats = np.zeros(8, np.int64)
    got = _run(model, d, 2, 7, stats)
    assert np.array_equal(got, oracle.lz4_compress_block(d))
    assert stats[0] > 0 and stats[2] > 0


# ---- Snappy -------------------------------------------------------------------------------------
SN_SRC = os.path.join(HERE, "model", "snappy_wave_model.cpp")
SN_SO = os.path.join(HERE, "model", "libsnappy_wave_model.so")


@pytest.fixture(scope="module")
def sn_model():
    if not os.path.exists(SN_SO) or os.path.getmtime(SN_SO) < os.path.getmtime(SN_SRC):
        subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", SN_SO, SN_SRC], check=True)
    L = ctypes.CDLL(SN_SO)
    L.snappy_wave_model_compress.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                             ctypes.c_uint64, ctypes.c_void_p]
    return L


@pytest.mark.parametrize("kind", range(corpus.N_KINDS))
def test_snappy_model_matches_oracle(sn_model, oracle, kind):
    rng = np.random.default_rng(700 + kind)
    for n in [1, 14, 15, 16, 20, 64, 65, 130, 1000, 4096, 18442, 20000, 32767, 32768]:
        if kind == 6 and n > 6000:
            continue
        d = corpus.chunk_corpus(kind, n, rng)
        want = oracle.snappy_compress_block(d)
        for mode in (0, 1, 2):
            out = np.empty(d.size + d.size // 6 + 64, np.uint8)
            m = sn_model.snappy_wave_model_compress(d.ctypes.data, d.size, out.ctypes.data, mode, n, None)
            assert np.array_equal(out[:m], want), (kind, n, mode)


def test_window_model_matches_oracle(oracle):
    """Exact-window parse (tests/model/lz4_window_model.cpp): fast windows + general batches, with
    candidate snapshots taken 0..2 windows early, must equal the oracle byte for byte."""
    src = os.path.join(HERE, "model", "lz4_window_model.cpp")
    so = os.path.join(HERE, "model", "liblz4_window_model.so")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so, src], check=True)
    L = ctypes.CDLL(so)
    L.lz4_window_model_compress.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                            ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                            ctypes.c_int, ctypes.c_void_p]
    for kind in range(corpus.N_KINDS):
        rng = np.random.default_rng(500 + kind)
        for n in [13, 64, 65, 129, 1000, 4096, 32768]:
            if kind == 6 and n > 6000:
                continue
            d = corpus.chunk_corpus(kind, n, rng)
            want = oracle.lz4_compress_block(d)
            for la in (0, 2):
                out = np.empty(d.size + 64, np.uint8)
                m = L.lz4_window_model_compress(d.ctypes.data, d.size, out.ctypes.data, 2, n, 1, 64, 8, la, None)
                if m < 0:
                    assert want.size >= d.size
                else:
                    assert np.array_equal(out[:m], want), (kind, n, la)


def test_snappy_window_model_matches_oracle(oracle):
    """Exact-window Snappy parse (tests/model/snappy_window_model.cpp): windows + general batches,
    under adversarial same-address store winners, against the oracle; and the window path must carry
    nearly all copies on the wide-row workload (that is what it is for)."""
    from s3shuffle import datagen

    src = os.path.join(HERE, "model", "snappy_window_model.cpp")
    so = os.path.join(HERE, "model", "libsnappy_window_model.so")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so, src], check=True)
    L = ctypes.CDLL(so)
    L.snappy_window_model_compress.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                               ctypes.c_uint64, ctypes.c_int, ctypes.c_void_p]

    def run(d, mode, seed, stats=None):
        out = np.empty(32 + d.size + d.size // 6 + 64, np.uint8)
        n = L.snappy_window_model_compress(d.ctypes.data, d.size, out.ctypes.data, mode, seed, 1,
                                           stats.ctypes.data if stats is not None else None)
        return out[:n]

    for kind in range(corpus.N_KINDS):
        rng = np.random.default_rng(500 + kind)
        for n in [0, 1, 14, 15, 16, 64, 65, 130, 191, 192, 193, 255, 256, 257, 300, 1000, 4096, 20000, 32768]:
            if kind == 6 and n > 6000:
                continue
            d = corpus.chunk_corpus(kind, n, rng)
            want = oracle.snappy_compress_block(d)
            for mode in (0, 1, 2):
                assert np.array_equal(run(d, mode, n), want), (kind, n, mode)
    d, _ = datagen.tpcds_wide_map_output(1 << 19, 8, 5)
    st = np.zeros(8, np.int64)
    for k in range(0, d.size - 32768, 32768):
        blk = np.ascontiguousarray(d[k:k + 32768])
        assert np.array_equal(run(blk, 2, k, st), oracle.snappy_compress_block(blk))
    assert st[2] > 20 * st[3] and st[1] > 0  # copies found in windows >> copies found by batches
