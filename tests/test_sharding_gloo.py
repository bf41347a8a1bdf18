"""N>1 path on CPU: world_size-2 gloo run of the sharding + timing/aggregation logic bench.py uses
(mapId % nGPU ownership, barrier, MAX over ranks of the elapsed time, SUM of bytes), with the
ORACLE standing in for the device codec so that the merged result can be checked bit for bit
against a single-process run.  (The oracle is only the checker/stand-in inside this test.)"""
import os
import socket
import struct

import numpy as np
import pytest

import corpus


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_maps, q):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "spark-s3-shuffle_amd"), os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    from oracle import binding as oracle
    from s3shuffle import sharding

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = sharding.partition_maps(range(n_maps), world)[rank]
    results = {}
    u = c = 0
    dist.barrier()
    for m in mine:
        data, offs = corpus.ragged_map_output(np.random.default_rng(1000 + m), 6, 50_000)
        img, index, sums = oracle.compress_map_output(1, 1, data, offs)
        results[m] = (img.tobytes(), oracle.longs_to_be(index), oracle.longs_to_be(sums))
        u += data.size
        c += img.size
    elapsed = torch.tensor([0.25 * (rank + 1)], dtype=torch.float64)
    dist.barrier()
    dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    agg = torch.tensor([u, c], dtype=torch.int64)
    dist.all_reduce(agg, op=dist.ReduceOp.SUM)
    q.put((rank, mine, results, float(elapsed.item()), agg.tolist()))
    dist.destroy_process_group()


def test_two_rank_sharded_compress_matches_single_process(oracle):
    import torch.multiprocessing as mp

    from s3shuffle import sharding

    world, n_maps = 2, 7
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_maps, q)) for r in range(world)]
    [p.start() for p in procs]
    got = [q.get(timeout=180) for _ in range(world)]
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    merged, owners = {}, {}
    for rank, mine, results, elapsed, agg in got:
        assert elapsed == pytest.approx(0.5)          # MAX over ranks
        for m in mine:
            assert sharding.device_for_map(m, world) == rank
            owners[m] = rank
        merged.update(results)
    assert sorted(merged) == list(range(n_maps)) and sorted(owners) == list(range(n_maps))
    u = c = 0
    for m in range(n_maps):
        data, offs = corpus.ragged_map_output(np.random.default_rng(1000 + m), 6, 50_000)
        img, index, sums = oracle.compress_map_output(1, 1, data, offs)
        assert merged[m] == (img.tobytes(), oracle.longs_to_be(index), oracle.longs_to_be(sums)), m
        u += data.size
        c += img.size
    assert got[0][4] == [u, c] and got[1][4] == [u, c]  # SUM over ranks


def test_sharding_helpers():
    from s3shuffle import sharding

    assert [sharding.device_for_map(m, 8) for m in range(10)] == [0, 1, 2, 3, 4, 5, 6, 7, 0, 1]
    assert sharding.map_ids_for_rank(3, 8, 3) == [3, 11, 19]
    assert sharding.partition_maps([5, 2, 9, 4], 2) == [[2, 4], [5, 9]]
    with pytest.raises(ValueError):
        sharding.device_for_map(1, 0)
    # S3ShuffleDispatcher.getPath: ${rootDir}${mapId % folderPrefixes}/${appId}/${shuffleId}/${name}
    assert sharding.block_path("file:///tmp/spark-s3-shuffle", "app-1", 4, 23, "data") == \
        "file:///tmp/spark-s3-shuffle/3/app-1/4/shuffle_4_23_0.data"
    assert sharding.block_name(0, 7, "checksum") == "shuffle_0_7_0.checksum"  # no .ADLER32 suffix
    assert struct.pack(">q", 1) == b"\x00" * 7 + b"\x01"


def test_bench_dry_run_two_ranks_gloo():
    """bench.py --gpus 2 --dry-run under torch.distributed.run on the CPU box (gloo): the launch line the driver
    uses for the scaling runs must come up, agree on mapId % nGPU and cover every map task exactly once."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29541", os.path.join(root, "bench.py"),
                          "--gpus", "2", "--dry-run", "--maps-per-gpu", "3"],
                         capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["ok"] and d["world"] == 2 and d["backend"] == "gloo"
    assert sorted(m for r in d["ranks"] for m in r["map_ids"]) == list(range(6))
    assert [r["map_ids"] for r in sorted(d["ranks"], key=lambda r: r["rank"])] == [[0, 2, 4], [1, 3, 5]]
