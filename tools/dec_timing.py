"""Phase accounting of the LZ4 decode kernel (instrumented build)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["S3S_CODEC_LIB"] = os.path.join(ROOT, "spark-s3-shuffle_amd", "lib", "libs3shuffle_codec_dbg.so")
sys.path.insert(0, os.path.join(ROOT, "spark-s3-shuffle_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import s3shuffle
from s3shuffle import datagen
data, offs = datagen.terasort_map_output(64 << 20, 100, seed=2)
lib = s3shuffle.load_library()
lib.s3s_debug_read_dec.argtypes = [ctypes.c_void_p, ctypes.c_int]
c = s3shuffle.Codec(0); c.set_option(3, 1)
d_src = torch.from_numpy(data).cuda()
cap = c.max_compressed_size(1, offs)
d_dst = torch.empty(cap, dtype=torch.uint8, device="cuda"); d_out = torch.empty(data.size, dtype=torch.uint8, device="cuda")
total, index, sums = c.compress_map_output_device(1, 1, d_src.data_ptr(), offs, d_dst.data_ptr(), cap)
buf = (ctypes.c_ulonglong * 16)()
for it in range(2):
    lib.s3s_debug_read_dec(buf, 1)
    n = c.decompress_range_device(1, 1, d_dst.data_ptr(), total, index, sums, d_out.data_ptr(), data.size)
    lib.s3s_debug_read_dec(buf, 1)
assert n == data.size and torch.equal(d_out, d_src)
if buf[10]:
    g = buf[10]
    print(f"GLOBAL decoder: frames {g}, decode ticks/frame {buf[8]/g:.0f}, hash ticks/frame {buf[9]/g:.0f}, seqs/frame {buf[11]/g:.0f}, flushes/frame {buf[12]/g:.1f}, extra drains/frame {buf[13]/g:.1f}; kernel {c.stage_ms(1):.3f} ms")
fr = max(buf[6], 1)
print(f"decode kernel {c.stage_ms(1):.3f} ms, frames decoded {buf[6]}, seqs/frame {buf[4]/fr:.0f}, slow-parse/frame {buf[5]/fr:.1f}")
print(f"per frame ticks: parse {buf[0]/fr:.0f}, match-copy {buf[1]/fr:.0f}, decode-wave total {buf[2]/fr:.0f}, workgroup total {buf[3]/fr:.0f}")
print(f"per sequence ticks: parse {buf[0]/max(buf[4],1):.0f}, match-copy {buf[1]/max(buf[4],1):.0f}")
