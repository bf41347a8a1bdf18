"""Multi-device readiness on whatever the box has: a context on EVERY visible device ordinal runs
the parity check against the oracle (one device on the 1-GPU box; all eight on a node), `mapId % nGPU` picks among
them, and the optional reduce-side gather executes over the `nccl` backend (RCCL) from / into the decode buffers."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import corpus

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LZ4, ADLER, CRC = 1, 1, 2


def test_context_on_every_device_matches_the_oracle(codec_lib, oracle):
    import s3shuffle
    from s3shuffle import datagen, sharding

    n_dev = s3shuffle.device_count()
    assert n_dev >= 1
    seen = set()
    for map_id in range(2 * n_dev):
        ordinal = sharding.device_for_map(map_id, n_dev)  # the rule bench.py and the shim use
        seen.add(ordinal)
        data, offs = datagen.terasort_map_output(2 << 20, 16, seed=21, map_id=map_id)
        want = oracle.compress_map_output(LZ4, CRC, data, offs)
        with s3shuffle.Codec(ordinal) as c:
            img, index, sums = c.compress_map_output(LZ4, CRC, data, offs)
            assert np.array_equal(img, want[0]) and np.array_equal(index, want[1]) and np.array_equal(sums, want[2]), ordinal
            assert np.array_equal(c.decompress_range(LZ4, CRC, img, index, sums), data), ordinal
    assert seen == set(range(n_dev))


def test_two_contexts_on_two_devices_at_once(codec_lib, oracle):
    """Contexts on different ordinals used alternately from one thread (the executor case: one JVM, several GPUs)."""
    import s3shuffle
    from s3shuffle import datagen

    n_dev = s3shuffle.device_count()
    if n_dev < 2:
        pytest.skip("one visible device: covered by test_context_on_every_device_matches_the_oracle")
    ctxs = [s3shuffle.Codec(d) for d in range(n_dev)]
    try:
        for it in range(2):
            for d, c in enumerate(ctxs):
                data, offs = datagen.terasort_map_output(1 << 20, 8, seed=22 + it, map_id=d)
                want = oracle.compress_map_output(LZ4, ADLER, data, offs)
                img, index, sums = c.compress_map_output(LZ4, ADLER, data, offs)
                assert np.array_equal(img, want[0]) and np.array_equal(index, want[1]) and np.array_equal(sums, want[2])
    finally:
        [c.close() for c in ctxs]


def test_reduce_side_gather_over_rccl():
    """gather_reduce_partitions on the nccl backend, one rank per visible GPU (one rank on the 1-GPU box)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29571", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "rccl_gather_worker.py")],
                         capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["ok"] and d["backend"] == "nccl" and d["keys"] == 3 * 7
    print(d["mode"])


def test_host_mirror_devices_key(codec_lib, oracle, tmp_path):
    """the C++ host mirror's `spark.shuffle.s3.gpu.devices` (Conf::numGpus) — map task m commits on device
    m % nGPU, every map output equals the oracle's image whatever device wrote it, and a reduce task reads them all back.
    On a one-GPU box the key is clamped to 1 and the same job runs on device 0; with several GPUs every ordinal is used
    (`devices` = 2 limits a bigger node to two of them)."""
    import s3shuffle
    from s3shuffle import host

    n_dev = s3shuffle.device_count()
    for want_devices in sorted({1, min(2, n_dev), n_dev}):
        root = "file://" + str(tmp_path / ("store%d" % want_devices))
        d = host.Dispatcher(root, num_gpus=want_devices)
        try:
            n_maps, R = 2 * max(want_devices, 1) + 1, 5
            used, images = set(), {}
            for m in range(n_maps):
                assert d.device_for_map(m) == m % want_devices
                used.add(d.device_for_map(m))
                rng = np.random.default_rng(100 + m)
                parts = [corpus.chunk_corpus(int(rng.integers(0, 8)), int(rng.integers(1, 60_000)), rng).tobytes() for _ in range(R)]
                w = host.MapOutputWriter(d, 0, m, R)
                for p in range(R):
                    w.get_partition_writer(p)
                    w.write(parts[p])
                    w.close_partition()
                w.commit_all_partitions()
                w.close()
                offs = np.concatenate([[0], np.cumsum([len(x) for x in parts])]).astype(np.int64)
                img, index, sums = oracle.compress_map_output(LZ4, ADLER, np.frombuffer(b"".join(parts), np.uint8), offs)
                assert open(d.get_path(host.KIND_DATA, 0, m), "rb").read() == img.tobytes(), (want_devices, m)
                images[m] = parts
            assert used == set(range(want_devices))
            for p in range(R):
                blocks = host.read_shuffle(d, 0, p, p + 1, True)
                got = sorted(bytes(b[4]) for b in blocks)
                assert got == sorted(images[m][p] for m in range(n_maps)), (want_devices, p)
        finally:
            d.remove_root()
            d.close()
