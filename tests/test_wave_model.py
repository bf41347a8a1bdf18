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
