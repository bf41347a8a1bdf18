"""s3s_checksum_ranges_device alone: Adler32 / CRC32 over N MiB resident in HBM (default 1 GiB: past the 256 MiB MALL),
HIP-event time of the call.  usage: python tools/checksum_bench.py [MiB]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "spark-s3-shuffle_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, zlib
import s3shuffle
n = (int(sys.argv[1]) if len(sys.argv) > 1 else 1024) << 20
rng = np.random.default_rng(1)
data = rng.integers(0, 256, n, dtype=np.uint8)
print('bytes', n, flush=True)
d = torch.from_numpy(data).cuda()
c = s3shuffle.Codec(0); c.set_option(3, 1)
for nranges in (1, 200, 2000):
    offs = np.linspace(0, n, nranges + 1).astype(np.int64)
    for algo, name in ((1, "adler32"), (2, "crc32")):
        best = 1e9
        for it in range(6):
            t = time.perf_counter(); out = c.checksum_ranges_device(algo, d.data_ptr(), offs); dt = time.perf_counter() - t
            best = min(best, dt)
        ref = (zlib.adler32 if algo == 1 else zlib.crc32)(data[offs[0]:offs[1]].tobytes())
        assert int(out[0]) == ref, (name, nranges)
        ms = max(c.stage_ms(3), c.stage_ms(0), 1e-6)
        print(f"{name:8s} {nranges:5d} ranges: wall {best*1e3:.3f} ms -> {n/best/1e12:.2f} TB/s | stage total {ms:.3f} ms -> {n/(ms*1e-3)/1e12:.2f} TB/s", flush=True)
