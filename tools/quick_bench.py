"""Early single-GPU probe: device-resident compress+checksum throughput and stage times."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "spark-s3-shuffle_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import s3shuffle
from s3shuffle import datagen

def main():
    size = int(sys.argv[1]) if len(sys.argv) > 1 else (256 << 20)
    nparts = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    kind = sys.argv[3] if len(sys.argv) > 3 else "terasort"
    if kind == "terasort":
        data, offs = datagen.terasort_map_output(size, nparts, seed=2)
    else:
        data, offs = datagen.skew_block(size, kind, seed=5)
    dev = torch.device("cuda:0")
    d_src = torch.from_numpy(data).to(dev)
    c = s3shuffle.Codec(0)
    cap = c.max_compressed_size(1, offs)
    d_dst = torch.empty(cap, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    c.set_option(3, 1)
    for variant, algo in ((1, 1), (1, 2)):
        c.set_option(4, variant)
        for it in range(3):
            t = time.perf_counter()
            total, index, sums = c.compress_map_output_device(1, algo, d_src.data_ptr(), offs, d_dst.data_ptr(), cap)
            dt = time.perf_counter() - t
            print(f"{kind} variant={variant} algo={algo} it={it}: U={data.size} C={total} ratio={data.size/total:.2f} wall={dt*1e3:.2f} ms "
                  f"-> {data.size/dt/1e9:.1f} GB/s | stages ms total={c.stage_ms(0):.2f} codec={c.stage_ms(1):.2f} "
                  f"assemble={c.stage_ms(2):.3f} checksum={c.stage_ms(3):.3f}")
    # checksum-only roofline probe on the uncompressed data
    for algo in (1, 2):
        for it in range(3):
            t = time.perf_counter()
            c.checksum_ranges_device(algo, d_src.data_ptr(), offs)
            dt = time.perf_counter() - t
            print(f"checksum-only algo={algo}: wall={dt*1e3:.2f} ms, kernel={c.stage_ms(3):.3f} ms -> {data.size/(c.stage_ms(3)*1e-3)/1e9:.0f} GB/s")

if __name__ == "__main__":
    main()
