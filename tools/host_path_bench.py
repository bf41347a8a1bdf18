"""PCIe-inclusive rate of the host-buffer entry points (never the bench `value`)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "spark-s3-shuffle_amd")); sys.path.insert(0, ROOT)
import numpy as np
import s3shuffle
from s3shuffle import datagen
data, offs = datagen.terasort_map_output(128 << 20, 200, seed=2)
c = s3shuffle.Codec(0)
for it in range(4):
    t = time.perf_counter(); img, index, sums = c.compress_map_output(1, 1, data, offs); dt = time.perf_counter() - t
    print(f"host->host compress+checksum: {data.size/dt/1e9:.2f} GB/s ({dt*1e3:.1f} ms for {data.size>>20} MiB)")
for it in range(4):
    t = time.perf_counter(); out = c.decompress_range(1, 1, img, index, sums, dst_capacity=data.size); dt = time.perf_counter() - t
    print(f"host->host verify+decompress: {data.size/dt/1e9:.2f} GB/s ({dt*1e3:.1f} ms)")
