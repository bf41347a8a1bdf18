"""The driver's contract for bench.py's stdout: the LAST line is one compact JSON object (the round-5
line had grown to 22 KB and the driver could not parse it), and `python bench.py --gpus N` launches its own ranks."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "spark-s3-shuffle_amd"))

CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")


def _canned(n_ranks=8):
    """A full record as run_workload builds it, with the longest strings the code can produce."""
    long = "x" * 700
    return {
        "metric": "shuffle_block_compress_checksum_throughput", "value": 112.034, "unit": "GB/s", "n_gpus": n_ranks,
        "rccl_ranks": n_ranks, "per_rank": [{"rank": r, "device": r, "elapsed_s": 0.19168, "GBps": 112.034, "map_tasks": 8}
                                            for r in range(n_ranks)],
        "steps": 20, "warmup": 5, "ms_per_step": 9.5841, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8", "data": "synthetic",
        "config": {"workload": "terasort-100g-2000p-lz4-crc32", "generator": long, "direction": "compress", "codec": "lz4 (" + long + ")",
                   "checksum": "adler32", "partitions_per_map_task": 2000, "map_task_bytes": 134217700, "map_tasks_per_gpu": 8,
                   "uncompressed_bytes_per_step": 8 * 1073741600, "compressed_bytes_per_step": 8 * 256694371,
                   "compression_ratio": 4.183, "sharding": long, "task_threads_per_gpu": 4, "map_tasks_per_library_call": 2,
                   "lz4_parse_variant": {"setting": "auto", "ran_in_last_call": [7]}, "inputs": long},
        "roofline": {"bound": "hbm", "kernel": long, "achieved": 64.297, "peak": 8000.0, "unit": "GB/s", "frac": 0.008037,
                     "traffic": 722134993, "avg_launch_ms": 5.1702, "concurrent_launches": 4, "launch_shape": long,
                     "achieved_all_streams": 257.19, "algorithmic_bytes_per_launch": 332428560, "frac_of_copy_ceiling": 0.010222,
                     "whole_path_read_frac": 0.014004, "traffic_over_algorithmic": 2.172, "traffic_source": long},
        "stages_ms_per_library_call": {"hash": 0.5114, "codec": 5.1702, "assemble": 2.0134, "checksum": 0.1411, "total": 7.9101},
        "image_verified": True,
        "cpu_baseline": {"value": 34.23, "unit": "GB/s", "cores": 16, "kind": "reference-lib", "sample": long,
                         "sample_short": long, "single_thread_GBps": 2.181, "wall_s": 11.92},
        "speedup_vs_cpu_all_cores": 3.273, "speedup_vs_cpu_1_core": 51.37, "kernel_sources_sha256": "b" * 64,
    }


def test_headline_is_compact_and_complete():
    import bench

    line = bench.compact_headline(_canned(), bench.SECONDARY_FILE)
    blob = json.dumps(line)
    assert len(blob) < bench.HEADLINE_MAX_BYTES < 8192, len(blob)
    back = json.loads(blob)
    for k in CONTRACT_KEYS:
        assert k in back, k
    assert back["roofline"]["traffic"] == 722134993 and back["roofline"]["frac"] == 0.008037 and back["roofline"]["bound"] == "hbm"
    assert set(("achieved", "peak", "unit", "frac", "traffic")) <= set(back["roofline"])
    assert back["cpu_baseline"]["cores"] == 16 and back["cpu_baseline"]["kind"] == "reference-lib" and back["cpu_baseline"]["value"] == 34.23
    assert back["config"]["workload"] == "terasort-100g-2000p-lz4-crc32" and "model" not in back["config"]
    assert back["image_verified"] is True and back["secondary_file"] == "bench_secondary.json"
    assert back["rccl_ranks"] == 8 and len(back["per_rank"]) == 8
    assert max(len(v) for v in back["config"].values() if isinstance(v, str)) <= 48


def test_secondary_summary_stays_short():
    import bench

    leg = {k: v for k, v in _canned(1).items()}
    leg["roofline"] = dict(leg["roofline"])
    sec = {label: dict(leg) for label, *_ in bench.SECONDARY}
    sec["broken:leg"] = {"error": "RuntimeError(" + "y" * 500 + ")"}
    sec["block_size_sweep"] = {"points": [{"block_MiB": m, "blocks_per_step": n, "compress": 101.234, "decompress": 391.123}
                                          for m, n in bench.SWEEP], "points_more_blocks_in_flight": [{"block_MiB": 8, "blocks_per_step": 32,
                                                                                                       "compress": 73.0, "decompress": 225.1}]}
    sec["hbm_bound_stages"] = {f"checksum-only-1gib:{a}:{r}": {"roofline": {"achieved": 5560.1, "frac": 0.695}, "matches_zlib": True}
                               for a in ("adler32", "crc32", "crc32c") for r in ("1-range", "2000-ranges")}
    sec["host_path"] = {"compress_by_task_threads": {"1": 50.1, "2": 53.6, "4": 52.3}, "verify_decompress_by_task_threads": {"1": 45.0, "2": 53.1},
                        "round_trip_bit_exact": True}
    blob = json.dumps(bench.secondary_summary(sec))
    # headline + summary together must sit inside the ~8.5 KB stdout tail the driver keeps
    assert len(blob) < 4096, len(blob)
    back = json.loads(blob)
    assert len(back["secondary_summary"]) == len(bench.SECONDARY) + 1 and back["host_path_GBps"]["bit_exact"] is True


def test_plain_gpus_2_launches_its_own_ranks():
    """`python bench.py --gpus 2 --dry-run` with no launcher and no WORLD_SIZE: two gloo ranks on the CPU box, mapId % 2."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "TORCHELASTIC_RUN_ID", "GROUP_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--maps-per-gpu", "3"],
                         capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["ok"] and d["world"] == 2 and d["backend"] == "gloo"
    assert [r["map_ids"] for r in sorted(d["ranks"], key=lambda r: r["rank"])] == [[0, 2, 4], [1, 3, 5]]
