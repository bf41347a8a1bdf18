#!/usr/bin/env python3
"""(GPU) Where does a SMALL batched call spend its time?  N single-partition TeraSort blocks of B MiB resident in HBM, one
batched device call per step (s3s_compress_map_outputs_batch_device / s3s_decompress_ranges_batch_device), one context:
wall clock per call against the library's own stage events (hash / codec / assemble / checksum, discover on the reduce side).
usage: python tools/small_blocks_probe.py [block MiB] [blocks] [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "spark-s3-shuffle_amd")); sys.path.insert(0, ROOT)
import numpy as np
import torch
import s3shuffle
from s3shuffle import datagen

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 8
nblk = int(sys.argv[2]) if len(sys.argv) > 2 else 8
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
dev = torch.device("cuda", 0)
c = s3shuffle.Codec(0)
c.set_option(s3shuffle.codec.OPT_PROFILE, 1)
LZ4, ADLER = s3shuffle.CODEC_LZ4, s3shuffle.CHECKSUM_ADLER32
big = datagen.skew_block(mib << 20, "terasort", seed=5, map_id=0)[0]
tasks = []
for k in range(nblk):
    data = np.roll(big, 100 * k)
    offs = np.array([0, data.size], np.int64)
    cap = c.max_compressed_size(LZ4, offs)
    tasks.append({"src": torch.from_numpy(data.copy()).to(dev), "offs": offs, "cap": cap, "dst": torch.empty(cap, dtype=torch.uint8, device=dev),
                  "out": torch.empty(data.size, dtype=torch.uint8, device=dev), "u": data.size})
torch.cuda.synchronize()
stages = ("hash", "codec", "assemble", "checksum", "discover", "total")
ids = {"hash": s3shuffle.codec.STAGE_HASH, "codec": s3shuffle.codec.STAGE_CODEC, "assemble": s3shuffle.codec.STAGE_ASSEMBLE,
       "checksum": s3shuffle.codec.STAGE_CHECKSUM, "discover": getattr(s3shuffle.codec, "STAGE_DISCOVER", 4), "total": s3shuffle.codec.STAGE_TOTAL}


def run(fn, label):
    for _ in range(3):
        fn()
    acc = {k: 0.0 for k in stages}
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
        for k in stages:
            acc[k] += c.stage_ms(ids[k])
    dt = (time.perf_counter() - t0) / reps
    u = sum(t["u"] for t in tasks)
    print(f"{label}: {nblk} x {mib} MiB per call: wall {dt*1e3:.3f} ms = {u/dt/1e9:.1f} GB/s; events (ms): " +
          ", ".join(f"{k} {acc[k]/reps:.3f}" for k in stages) + f"; outside the events {dt*1e3 - acc['total']/reps:.3f}", flush=True)


res = [None]
def comp():
    res[0] = c.compress_map_outputs_batch_device(LZ4, ADLER, [(t["src"].data_ptr(), t["offs"], t["dst"].data_ptr(), t["cap"]) for t in tasks])
run(comp, "compress+checksum")
for t, r in zip(tasks, res[0]):
    t["total"], t["index"], t["sums"] = r
def dec():
    r = c.decompress_ranges_batch_device(LZ4, ADLER, [(t["dst"].data_ptr(), t["total"], t["index"], t["sums"], t["out"].data_ptr(), t["u"]) for t in tasks])
    assert all(x[0] == 0 for x in r)
run(dec, "verify+decompress")
assert all(bool(torch.equal(t["out"], t["src"])) for t in tasks)
