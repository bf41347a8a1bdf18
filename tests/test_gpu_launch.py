"""The N-GPU launch line, on one GPU: bench.py under `python -m torch.distributed.run
--nproc-per-node 1` creates the RCCL process group (backend "nccl"), runs the barrier + MAX / SUM all_reduce the
scaling runs use, and prints the contract's JSON line; --dry-run validates rank <-> device <-> mapId."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(extra, port):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1"] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])


def test_dry_run_over_rccl():
    d = _launch(["--dry-run", "--maps-per-gpu", "4"], 29551)
    assert d["ok"] and d["backend"] == "nccl" and d["world"] == 1
    assert d["ranks"][0]["map_ids"] == [0, 1, 2, 3] and d["ranks"][0]["device"] == "cuda:0"


def test_timed_run_through_the_process_group():
    d = _launch(["--steps", "2", "--warmup", "1", "--maps-per-gpu", "2", "--map-mib", "16", "--no-cpu-baseline"], 29552)
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["metric"] == "shuffle_block_compress_checksum_throughput" and d["unit"] == "GB/s" and d["value"] > 0
    assert d["config"]["workload"] == "terasort-10g-200p-lz4" and "roofline" in d
    assert d["image_verified"] is True and d["rccl_ranks"] == 1


def test_plain_command_last_line_is_the_compact_headline():
    """The driver's N=1 form (`python bench.py --gpus 1 ...`, no launcher): the LAST stdout line alone must round-trip as the
    contract's JSON object, under 4 KB, with roofline + cpu_baseline + image_verified."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--maps-per-gpu", "2",
           "--map-mib", "16", "--cpu-seconds", "1", "--full-line"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = out.stdout.strip().splitlines()
    last = lines[-1]
    assert len(last) < 4096
    d = json.loads(last)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["image_verified"] is True and d["cpu_baseline"]["cores"] >= 1 and d["roofline"]["bound"] == "hbm"
    assert d["roofline"]["achieved"] > 0 and d["rccl_ranks"] == 0
    full = json.loads(lines[-2])["full_record"]
    assert full["value"] == d["value"] and len(full["cpu_baseline"]["sample"]) > len(d["cpu_baseline"]["sample"])
