"""PCIe-inclusive rate of the host-buffer entry points (never the bench `value`):
pageable numpy buffers vs page-locked buffers from s3s_host_alloc, 1..N task threads (one
context each).  usage: python tools/host_path_bench.py [threads=1,2,3] [reps=6]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "spark-s3-shuffle_amd")); sys.path.insert(0, ROOT)
import numpy as np
import s3shuffle
from s3shuffle import datagen

threads = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,2,3").split(",")]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
data, offs = datagen.terasort_map_output(128 << 20, 200, seed=2)
c0 = s3shuffle.Codec(0)
img, index, sums = c0.compress_map_output(1, 1, data, offs)
cap = c0.max_compressed_size(1, offs)


def run(kind, direction, nthreads):
    ctxs = [s3shuffle.Codec(0) for _ in range(nthreads)]
    bufs = []
    for _ in range(nthreads):
        if kind == "pinned":
            a, b, o = s3shuffle.PinnedBuffer(data.size), s3shuffle.PinnedBuffer(cap), s3shuffle.PinnedBuffer(data.size)
            a.array[:] = data
            b.array[: img.size] = img
            bufs.append((a, b, o))
        else:
            bufs.append(None)

    def work(i):
        c = ctxs[i]
        for _ in range(reps):
            if kind == "pinned":
                a, b, o = bufs[i]
                if direction == "compress":
                    c.compress_map_output(1, 1, a.array, offs, out=b.array)
                else:
                    c.decompress_range(1, 1, b.array[: img.size], index, sums, out=o.array)
            else:
                if direction == "compress":
                    c.compress_map_output(1, 1, data, offs)
                else:
                    c.decompress_range(1, 1, img, index, sums, dst_capacity=data.size)

    work_warm = [threading.Thread(target=work, args=(i,)) for i in range(nthreads)]
    [t.start() for t in work_warm]; [t.join() for t in work_warm]
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(nthreads)]
    [t.start() for t in th]; [t.join() for t in th]
    dt = time.perf_counter() - t0
    for b in bufs:
        if b:
            [x.free() for x in b]
    for c in ctxs:
        c.close()
    return data.size * reps * nthreads / dt / 1e9


for direction in ("compress", "decompress"):
    for kind in ("pageable", "pinned"):
        for n in threads:
            print(f"host->host {direction:10s} {kind:8s} threads={n}: {run(kind, direction, n):6.2f} GB/s", flush=True)
