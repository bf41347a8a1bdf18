"""Pre-check for the speculative next-window candidate gather: with the table snapshot of
window W+1 taken BEFORE window W is resolved (lookahead 1), in how many windows is the snapshot still the
table's answer (a) for every live lane, (b) for the lanes the sequential code visits; and how many windows have
a duplicate-hash group among their live lanes (the second duplicate pass is needed only then).
TEST INFRASTRUCTURE: runs tests/model/lz4_window_model.cpp, checks its bytes against the oracle."""
import ctypes
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "spark-s3-shuffle_amd"))
sys.path.insert(0, ROOT)
from s3shuffle import datagen  # noqa: E402
from oracle import binding as oracle  # noqa: E402

SRC = os.path.join(ROOT, "tests", "model", "lz4_window_model.cpp")
SO = os.path.join(ROOT, "tests", "model", "liblz4_window_model.so")
if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(SRC):
    subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", SO, SRC], check=True)
L = ctypes.CDLL(SO)
L.lz4_window_model_compress.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64,
                                        ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]


def run(name, data, nblocks=64):
    st = np.zeros(16, np.int64)
    for b in range(nblocks):
        d = np.ascontiguousarray(data[b * 32768:(b + 1) * 32768])
        if d.size < 32768:
            break
        out = np.empty(d.size + 64, np.uint8)
        n = L.lz4_window_model_compress(d.ctypes.data, d.size, out.ctypes.data, 0, 1, 1, 8, 4, 1, st.ctypes.data)
        assert n > 0 and np.array_equal(out[:n], oracle.lz4_compress_block(d)), (name, b)
    w = st[3]
    print(f"{name}: windows {w}, sequences {st[2]} ({st[2] / w:.2f}/window), general batches {st[0]}")
    print(f"   snapshot stale for a LIVE lane      : {st[12]} ({100 * st[12] / w:.1f} % of windows)")
    print(f"   snapshot stale for a VISITED lane   : {st[11]} ({100 * st[11] / w:.1f} %)")
    print(f"   ... not all of them candidates in the previous window: {st[14]} ({100 * st[14] / w:.1f} %)")
    print(f"   duplicate-hash group among live lanes: {st[13]} ({100 * st[13] / w:.1f} %)")


if __name__ == "__main__":
    d, _ = datagen.terasort_map_output(8 << 20, 4, seed=1)
    run("terasort", d)
    d, _ = datagen.tpcds_wide_map_output(8 << 20, 4, seed=1)
    run("wide rows", d)
