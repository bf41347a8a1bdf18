"""LZ4 compress variants on match-dense wide rows vs TeraSort records (single stream)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "spark-s3-shuffle_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import s3shuffle
from s3shuffle import datagen

def main():
    size = int(sys.argv[1]) if len(sys.argv) > 1 else (128 << 20)
    variants = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 10]
    dev = torch.device("cuda:0")
    torch.zeros(1, device=dev)
    c = s3shuffle.Codec(0)
    c.set_option(3, 1)
    for kind in ("tpcds", "terasort", "kvint"):
        if kind == "tpcds":
            data, offs = datagen.tpcds_wide_map_output(size, 200, seed=5)
        elif kind == "terasort":
            data, offs = datagen.terasort_map_output(size, 200, seed=2)
        else:
            data, offs = datagen.kv_int_map_output(size // 4, 200, seed=7)
        d_src = torch.from_numpy(data).to(dev)
        cap = c.max_compressed_size(1, offs)
        d_dst = torch.empty(cap, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        ref = None
        for variant in variants:
            c.set_option(4, variant)
            for it in range(3):
                t = time.perf_counter()
                total, index, sums = c.compress_map_output_device(1, 1, d_src.data_ptr(), offs, d_dst.data_ptr(), cap)
                dt = time.perf_counter() - t
            img = d_dst[:total].cpu().numpy().copy()
            same = "" if ref is None else f" identical_to_first={bool(np.array_equal(img, ref))}"
            if ref is None:
                ref = img
            print(f"{kind} lz4 variant={variant}: U={data.size} C={total} ratio={data.size/total:.2f} wall={dt*1e3:.2f} ms "
                  f"-> {data.size/dt/1e9:.1f} GB/s | codec={c.stage_ms(1):.2f} ms{same}", flush=True)

if __name__ == "__main__":
    main()
