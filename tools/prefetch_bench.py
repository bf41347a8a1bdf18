"""End-to-end host path through the C++ mirror: map tasks write a shuffle through
S3ShuffleMapOutputWriter (pinned staging -> GPU compress+checksum -> store), a reduce task reads it
back through (a) the sequential one-context reader with pageable buffers and (b) the
S3BufferedPrefetchIterator pipeline (fetch threads -> pinned -> decode contexts -> pinned).  The store
is a tmpfs / page-cache directory, so the numbers are PCIe + kernel + memcpy, not disk.
usage: python tools/prefetch_bench.py [n_maps=16] [map_mib=128] [root=/dev/shm/s3s_bench]"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "spark-s3-shuffle_amd")); sys.path.insert(0, ROOT)
import numpy as np
import s3shuffle
from s3shuffle import datagen, host

n_maps = int(sys.argv[1]) if len(sys.argv) > 1 else 16
map_mib = int(sys.argv[2]) if len(sys.argv) > 2 else 128
root = sys.argv[3] if len(sys.argv) > 3 else ("/dev/shm/s3s_bench" if os.path.isdir("/dev/shm") else "/tmp/s3s_bench")
n_parts = 200
d = host.Dispatcher("file://" + root)
d.remove_root()
data, offs = datagen.terasort_map_output(map_mib << 20, n_parts, seed=2)
total = data.size * n_maps


def write_maps(ids):
    for m in ids:
        w = host.MapOutputWriter(d, 0, m, n_parts)
        for p in range(n_parts):
            w.get_partition_writer(p)
            w.write(data[offs[p]:offs[p + 1]])
        w.commit_all_partitions()
        w.close()


for nthreads in (1, 2, 4, 8):
    best = 1e9
    for rep in range(3):  # first pass warms the page-locked cache and the context cache
        d.remove_shuffle(0)
        t0 = time.perf_counter()
        th = [threading.Thread(target=write_maps, args=(range(i, n_maps, nthreads),)) for i in range(nthreads)]
        [t.start() for t in th]; [t.join() for t in th]
        best = min(best, time.perf_counter() - t0)
    dt = best
    print(f"map side   {nthreads} task threads: {total/dt/1e9:6.2f} GB/s uncompressed ({n_maps} x {map_mib} MiB incl. staging copy + file write)", flush=True)

best = 1e9
for rep in range(3):
    t0 = time.perf_counter(); st = host.consume_sequential(d, 0, 0, n_parts, True); best = min(best, time.perf_counter() - t0)
assert st["decoded_bytes"] == total
print(f"reduce side sequential reader (1 context, pageable buffers): {total/best/1e9:6.2f} GB/s decoded", flush=True)
for (fetchers, decoders) in ((4, 1), (8, 2), (10, 2), (10, 3), (16, 4)):
    d.set_prefetch(1 << 30, fetchers, decoders, 2 << 30)
    best = 0
    for _ in range(3):
        t0 = time.perf_counter(); st = host.consume_prefetched(d, 0, 0, n_parts, True); dt = time.perf_counter() - t0
        assert st["decoded_bytes"] == total
        best = max(best, total / dt / 1e9)
    print(f"reduce side prefetch pipeline fetch={fetchers:2d} decode={decoders}: {best:6.2f} GB/s decoded "
          f"(pinned high water {st['pinned_high_water_compressed']>>20} + {st['pinned_high_water_decoded']>>20} MiB, "
          f"consumer waited {st['seconds_waiting']*1e3:.0f} ms)", flush=True)
d.set_prefetch(128 << 20, 10, 2, 512 << 20)
t0 = time.perf_counter(); st = host.consume_prefetched(d, 0, 0, n_parts, True); dt = time.perf_counter() - t0
print(f"reduce side prefetch pipeline, reference budgets (128 MiB compressed, 10 threads) + 512 MiB decoded: {total/dt/1e9:6.2f} GB/s", flush=True)
d.remove_root()
d.close()
