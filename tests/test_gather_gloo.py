"""The optional reduce-side gather (all-to-all-v of decoded partition ranges), world_size 2 on gloo:
every reducer must end up with its partition of EVERY map output, byte for byte."""
import os
import socket

import numpy as np


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _part(m, r):
    rng = np.random.default_rng(1000 * m + r)
    return rng.integers(0, 256, int(rng.integers(0, 5000)), dtype=np.uint8)


def _worker(rank, world, port, n_maps, n_reduce, q):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "spark-s3-shuffle_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    from s3shuffle import gather, sharding

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = sharding.partition_maps(range(n_maps), world)[rank]
    decoded = {(m, r): torch.from_numpy(_part(m, r)) for m in mine for r in range(n_reduce)}
    got = gather.gather_reduce_partitions(decoded, n_reduce)
    q.put((rank, {k: v.numpy().tobytes() for k, v in got.items()}))
    dist.destroy_process_group()


def test_two_rank_reduce_side_gather():
    import torch.multiprocessing as mp

    world, n_maps, n_reduce = 2, 5, 7
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_maps, n_reduce, q)) for r in range(world)]
    [p.start() for p in procs]
    res = dict(q.get(timeout=180) for _ in range(world))
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    for rank in range(world):
        want_keys = {(m, r) for m in range(n_maps) for r in range(n_reduce) if r % world == rank}
        assert set(res[rank]) == want_keys
        for (m, r) in want_keys:
            assert res[rank][(m, r)] == _part(m, r).tobytes()
