"""Phase accounting of the LZ4 compress kernel (instrumented build, `make -C csrc dbg`)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["S3S_CODEC_LIB"] = os.path.join(ROOT, "spark-s3-shuffle_amd", "lib", "libs3shuffle_codec_dbg.so")
sys.path.insert(0, os.path.join(ROOT, "spark-s3-shuffle_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import s3shuffle
from s3shuffle import datagen
kind = sys.argv[1] if len(sys.argv) > 1 else "terasort"
size = int(sys.argv[2]) if len(sys.argv) > 2 else (64 << 20)
if kind == "terasort":
    data, offs = datagen.terasort_map_output(size, 100, seed=2)
else:
    data, offs = datagen.tpcds_wide_map_output(size, 100, seed=3)
lib = s3shuffle.load_library()
lib.s3s_debug_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
c = s3shuffle.Codec(0)
c.set_option(3, 1)
d_src = torch.from_numpy(data).cuda()
cap = c.max_compressed_size(1, offs)
d_dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
names = ["top2Tread", "dup_passes", "G_wait", "runs", "slow_ext", "commit", "prep_valu", "", "windows", "seqs", "slow", "general", "ev_find", "ev_emit", "ev_update", "suspects", "nonmatch"]
variants = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [10]
for variant in variants:
    c.set_option(4, variant)
    for it in range(2):
        buf = (ctypes.c_ulonglong * 32)()
        lib.s3s_debug_read(buf, 1)
        total, index, sums = c.compress_map_output_device(1, 1, d_src.data_ptr(), offs, d_dst.data_ptr(), cap)
        lib.s3s_debug_read(buf, 1)
    nchunks = sum(-(-int(offs[i + 1] - offs[i]) // 32768) for i in range(len(offs) - 1))
    print(f"{kind} variant {variant}: codec {c.stage_ms(1):.3f} ms, chunks {nchunks}, ratio {data.size/total:.2f}")
    for i, n in enumerate(names):
        if n:
            print(f"  {n:10s} {buf[i] / nchunks:12.1f} per chunk")
    cyc = sum(buf[i] for i in (0, 1, 2, 3, 5, 6)) / nchunks
    print(f"  accounted cycles/chunk {cyc:.0f}; per window: " + ", ".join(f"{names[i]}={buf[i]/max(buf[8],1):.0f}" for i in (0, 1, 2, 6, 3, 5)) + f"; per slow ext {buf[4]/max(buf[10],1):.0f}")
