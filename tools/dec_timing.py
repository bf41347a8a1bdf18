"""Phase accounting of the batch LZ4 decoder (instrumented build: make -C spark-s3-shuffle_amd/csrc dbg).
s_memtime ticks of lane 0 per phase, summed over the frames of one call; reading the clock waits for the LDS
queue, so the split is approximate — the shares are what matters."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["S3S_CODEC_LIB"] = os.path.join(ROOT, "spark-s3-shuffle_amd", "lib", "libs3shuffle_codec_dbg.so")
sys.path.insert(0, os.path.join(ROOT, "spark-s3-shuffle_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import s3shuffle
from s3shuffle import datagen
which = sys.argv[1] if len(sys.argv) > 1 else "terasort"
n_bytes = (int(sys.argv[2]) if len(sys.argv) > 2 else 128) << 20
data, offs = (datagen.terasort_map_output(n_bytes, 200, seed=2) if which == "terasort"
              else datagen.tpcds_wide_map_output(n_bytes, 200, seed=2))
lib = s3shuffle.load_library()
lib.s3s_debug_read_bdec.argtypes = [ctypes.c_void_p, ctypes.c_int]
c = s3shuffle.Codec(0); c.set_option(3, 1)
d_src = torch.from_numpy(data).cuda()
cap = c.max_compressed_size(1, offs)
d_dst = torch.empty(cap, dtype=torch.uint8, device="cuda"); d_out = torch.empty(data.size, dtype=torch.uint8, device="cuda")
total, index, sums = c.compress_map_output_device(1, 1, d_src.data_ptr(), offs, d_dst.data_ptr(), cap)
buf = (ctypes.c_ulonglong * 16)()
for it in range(3):
    lib.s3s_debug_read_bdec(buf, 1)
    n = c.decompress_range_device(1, 1, d_dst.data_ptr(), total, index, sums, d_out.data_ptr(), data.size)
    torch.cuda.synchronize()
    lib.s3s_debug_read_bdec(buf, 0)
assert n == data.size and torch.equal(d_out, d_src)
fr = max(buf[12], 1)
names = ["parse", "batch setup", "literals", "rounds lane", "rounds quarter", "singles", "slide+flush", "slow/big"]
tot = buf[8] / fr
print(f"{which}: frames {fr}, ticks per frame {tot:.0f}; rounds per frame: lane {buf[9]/fr:.1f}, quarter {buf[10]/fr:.1f}, single {buf[11]/fr:.1f}; call {c.stage_ms(0):.3f} ms total")
for k, nm in enumerate(names):
    print(f"  {nm:15s} {buf[k]/fr:9.0f} ticks/frame  {100.0*buf[k]/max(buf[8],1):5.1f} %")
