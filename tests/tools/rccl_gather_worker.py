"""Single-process worker of tests/test_gpu_multidevice.py: reduce-side gather over the `nccl` (= RCCL) backend.

Runs under RANK / WORLD_SIZE from the environment (world 1 on the 1-GPU box; the same file works under
torch.distributed.run with more ranks).  Every rank compresses + decodes ITS map outputs with the product library on
its own device and hands views of the decode buffer to gather_reduce_partitions; the result is compared with the
generator's bytes.  Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "spark-s3-shuffle_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import s3shuffle  # noqa: E402
from s3shuffle import datagen, gather, sharding  # noqa: E402


def main():
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29571")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    n_maps, n_reduce = 3 * world, 7
    mine = sharding.partition_maps(range(n_maps), world)[rank]
    decoded, keep = {}, []
    with s3shuffle.Codec(local) as c:
        for m in mine:
            data, offs = datagen.terasort_map_output(1 << 20, n_reduce, seed=11, map_id=m)
            d_src = torch.from_numpy(data).to(dev)
            cap = c.max_compressed_size(s3shuffle.CODEC_LZ4, offs)
            d_img = torch.empty(cap, dtype=torch.uint8, device=dev)
            total, index, sums = c.compress_map_output_device(s3shuffle.CODEC_LZ4, s3shuffle.CHECKSUM_CRC32, d_src.data_ptr(), offs,
                                                              d_img.data_ptr(), cap)
            d_out = torch.empty(data.size, dtype=torch.uint8, device=dev)
            n = c.decompress_range_device(s3shuffle.CODEC_LZ4, s3shuffle.CHECKSUM_CRC32, d_img.data_ptr(), total, index, sums,
                                          d_out.data_ptr(), data.size)
            assert n == data.size
            keep.append(d_out)
            for r in range(n_reduce):  # views into the decode buffer: nothing is staged
                decoded[(m, r)] = d_out[int(offs[r]):int(offs[r + 1])]
    torch.cuda.synchronize()
    mode = "self send/recv through RCCL"
    try:
        got = gather.gather_reduce_partitions(decoded, n_reduce, local_through_collective=True)
    except RuntimeError as e:  # a build without self point-to-point: the header all_to_all still ran over RCCL
        mode = f"local partitions aliased ({type(e).__name__})"
        got = gather.gather_reduce_partitions(decoded, n_reduce)
    torch.cuda.synchronize()
    want_keys = {(m, r) for m in range(n_maps) for r in range(n_reduce) if r % world == rank}
    ok = set(got) == want_keys
    for (m, r) in sorted(want_keys):
        data, offs = datagen.terasort_map_output(1 << 20, n_reduce, seed=11, map_id=m)
        ok = ok and np.array_equal(got[(m, r)].cpu().numpy(), data[int(offs[r]):int(offs[r + 1])])
    dist.barrier()
    dist.destroy_process_group()
    print(json.dumps({"rank": rank, "world": world, "ok": bool(ok), "keys": len(got), "mode": mode, "backend": "nccl"}))
    if not ok:
        raise SystemExit(2)


if __name__ == "__main__":
    main()
