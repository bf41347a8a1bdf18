"""Latency of one host-buffer call as a function of the map output size (the break-even a shim needs
for `spark.shuffle.s3.gpu.minBytes`): GPU call vs the oracle's single-thread liblz4-equivalent on the
same bytes.  usage: python tests/tools/small_call_latency.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "spark-s3-shuffle_amd")); sys.path.insert(0, ROOT)
import numpy as np
import s3shuffle
from s3shuffle import datagen
from oracle import binding as oracle

c = s3shuffle.Codec(0)
for size in (4 << 10, 64 << 10, 512 << 10, 2 << 20, 8 << 20, 32 << 20):
    data, offs = datagen.terasort_map_output(size, 20, seed=2)
    src = s3shuffle.PinnedBuffer(data.size); src.array[:] = data
    cap = c.max_compressed_size(1, offs)
    dst = s3shuffle.PinnedBuffer(cap); back = s3shuffle.PinnedBuffer(data.size)
    for _ in range(3):
        img, index, sums = c.compress_map_output(1, 1, src.array, offs, out=dst.array)
    reps = 20
    t = time.perf_counter()
    for _ in range(reps):
        img, index, sums = c.compress_map_output(1, 1, src.array, offs, out=dst.array)
    t_c = (time.perf_counter() - t) / reps
    comp = img.copy()
    for _ in range(3):
        c.decompress_range(1, 1, comp, index, sums, out=back.array)
    t = time.perf_counter()
    for _ in range(reps):
        c.decompress_range(1, 1, comp, index, sums, out=back.array)
    t_d = (time.perf_counter() - t) / reps
    t = time.perf_counter()
    for _ in range(3):
        oracle.compress_map_output(1, 1, data, offs)
    t_cpu = (time.perf_counter() - t) / 3
    print(f"{data.size:>10d} B  gpu compress+checksum {t_c*1e6:8.0f} us ({data.size/t_c/1e9:6.2f} GB/s)   "
          f"gpu verify+decompress {t_d*1e6:8.0f} us ({data.size/t_d/1e9:6.2f} GB/s)   "
          f"cpu 1 thread compress+checksum {t_cpu*1e6:8.0f} us ({data.size/t_cpu/1e9:5.2f} GB/s)", flush=True)
    for b in (src, dst, back):
        b.free()
