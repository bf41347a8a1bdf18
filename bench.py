#!/usr/bin/env python3
"""bench.py — shuffle-block compress+checksum throughput on MI355X (BASELINE.json's metric).

One "step" = one pass of the map-side hot path (LZ4Block compress + per-partition checksum +
.data/.index assembly, s3s_compress_map_output_device) over one batch of synthetic map outputs
that is ALREADY RESIDENT in HBM.  Workload at N=1 = BASELINE.json configs[1] ("TeraSort 10 GB,
200 partitions, LZ4"): a batch is `--maps-per-gpu` TeraSort map tasks of one 128 MiB input
split each, range-partitioned into 200 reduce partitions.  With N GPUs map task
m runs on GPU m % N (the reference's mapId % folderPrefixes sharding,
S3ShuffleDispatcher.scala:142-143); per-GPU work is fixed -> weak scaling; the compress path
has no exchange step, so there is no data-path collective (only the timing barrier).

Launch:  python bench.py --gpus 1 --steps K --warmup W
         python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
                --master-port P bench.py --gpus N --steps K --warmup W
         python bench.py --gpus N ...   (no launcher: re-executes itself under torch.distributed.run)
Rank 0's LAST stdout line is the ONE compact JSON line of the contract (< 4 KB) with `roofline`
(dominant kernel: the LZ4 block-compress kernel, HIP-event timed on the library's own stream) and
`cpu_baseline` (the CPU oracle driving liblz4 1.9.3 on the host cores; checker code, timed here only
as the baseline).  The plain N=1 command also runs the other BASELINE configurations: their full
records go to bench_secondary.json and to an EARLIER stdout line, a short summary to the line
before the headline.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "spark-s3-shuffle_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

# executor environment of the GPU path: one hardware queue per task-thread stream instead of
# the HIP default of 4 shared ones; read by libamdhip64 when it is loaded, i.e. before `import torch`
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec (/opt/skills/guides/MI355X_MICROARCH.md)
HBM_COPY_CEILING_GBPS = 6290.0  # measured float4-copy ceiling, same guide

WORKLOADS = {
    # name: (generator, partitions, codec, checksum)
    "terasort-10g-200p-lz4": ("terasort", 200, "lz4", "adler32"),     # configs[1]  (N=1 headline)
    "tpcds-wide-100g-200p-snappy": ("tpcds", 200, "snappy", "adler32"),   # configs[2]
    "tpcds-wide-100g-200p-lz4": ("tpcds", 200, "lz4", "adler32"),         # configs[2] rows under the default codec
    "terasort-100g-2000p-lz4-crc32": ("terasort", 2000, "lz4", "crc32"),  # configs[3]
    "skew-1part-lz4": ("skew", 1, "lz4", "adler32"),                  # configs[4] (use --direction decompress)
    "terasort-10g-200p-zstd": ("terasort", 200, "zstd", "adler32"),   # reduce side only (--direction decompress)
    "tpcds-wide-100g-200p-zstd": ("tpcds", 200, "zstd", "adler32"),   # the same for wide rows (more sequences per byte, treeless blocks)
    "terasort-100g-2000p-zstd": ("terasort", 2000, "zstd", "adler32"),  # 64 KiB frames: one block each (the literal wavefront cannot run ahead)
    # objects of a JVM writer with spark.io.compression.lz4.blockSize=256k (S3ShuffleReader.scala:57-59): reduce side only
    "terasort-10g-200p-lz4-256k": ("terasort", 200, "lz4", "adler32"),
}
WORKLOADS["terasort-10g-200p-lzf"] = ("terasort", 200, "lzf", "adler32")  # spark.io.compression.codec=lzf: reduce side only
JVM_LZ4_BLOCK = {"terasort-10g-200p-lz4-256k": 262144}  # workloads whose inputs are LZ4Block images written by liblz4 on the host


def jvm_lz4_map_output_image(data, offs, algo_name: str, block_size: int):
    """What LZ4BlockOutputStream(blockSize) writes per partition, built with liblz4 (third-party library of the image, as libzstd for
    the zstd inputs): LZ4_compress_default per block (>= 64 KiB: its byU32 parse), token level = log2(blockSize) - 10, stored block when
    compression does not help, end frame; per-partition checksums by zlib.  The product only decodes these."""
    import ctypes
    import struct
    import zlib

    import xxhash

    L = ctypes.CDLL("liblz4.so.1")
    L.LZ4_compress_default.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    level = max(0, int(block_size).bit_length() - 1 - 10)
    buf = np.empty(block_size + block_size // 200 + 64, np.uint8)
    streams = []
    for p in range(len(offs) - 1):
        part = np.ascontiguousarray(data[offs[p]:offs[p + 1]])
        out = bytearray()
        for q in range(0, part.size, block_size):
            chunk = part[q:q + block_size]
            n = L.LZ4_compress_default(chunk.ctypes.data, buf.ctypes.data, chunk.size, buf.size)
            raw = n <= 0 or n >= chunk.size
            body = chunk.tobytes() if raw else buf[:n].tobytes()
            out += b"LZ4Block" + bytes([(0x10 if raw else 0x20) | level]) + struct.pack(
                "<iiI", len(body), chunk.size, xxhash.xxh32(chunk.tobytes(), seed=0x9747B28C).intdigest() & 0x0FFFFFFF) + body
        if part.size:
            out += b"LZ4Block" + bytes([0x10 | level]) + struct.pack("<iii", 0, 0, 0)
        streams.append(bytes(out))
    img = np.frombuffer(b"".join(streams), np.uint8)
    index = np.concatenate([[0], np.cumsum([len(x) for x in streams])]).astype(np.int64)
    f = zlib.adler32 if algo_name == "adler32" else zlib.crc32
    return img, index, np.array([f(x) for x in streams], np.int64)


def lzf_map_output_image(data, offs, algo_name: str):
    """(.data image, index, checksums) of one map task under LZFCompressionCodec: what compress-lzf's LZFOutputStream writes per
    partition (chunks of <= 65 535 bytes around liblzf blocks), built by liblzf 3.6 itself through the image's conda python3.9
    (tests/golden/make_lzf_golden.py --streams) — third-party library as input generator, like libzstd above; decode only."""
    import struct
    import subprocess
    import zlib

    blob = b"".join(struct.pack("<Q", int(offs[p + 1] - offs[p])) + data[offs[p]:offs[p + 1]].tobytes() for p in range(len(offs) - 1))
    r = subprocess.run(["/opt/conda/bin/python3.9", os.path.join(ROOT, "tests", "golden", "make_lzf_golden.py"), "--streams"],
                       input=blob, capture_output=True, check=True).stdout
    streams, pos = [], 0
    while pos < len(r):
        (n,) = struct.unpack_from("<Q", r, pos)
        streams.append(r[pos + 8:pos + 8 + n])
        pos += 8 + n
    assert len(streams) == len(offs) - 1
    img = np.frombuffer(b"".join(streams), np.uint8)
    index = np.concatenate([[0], np.cumsum([len(x) for x in streams])]).astype(np.int64)
    f = zlib.adler32 if algo_name == "adler32" else zlib.crc32
    return img, index, np.array([f(x) for x in streams], np.int64)


def cpu_baseline_lzf_decompress(workload: str, target_s: float, map_mib: int):
    """The host beside the LZF line: per-partition Adler32 validation + the chunk walk + a plain C restatement of liblzf's
    lzf_decompress (oracle/s3s_oracle_lzf.c: liblzf itself is reachable only through a python module here), one fetched
    range per thread — kind "port"."""
    from oracle import binding as oracle

    gen, nparts, codec, algo = WORKLOADS[workload]
    cores = usable_cores()
    data, offs = make_map_output(workload, 0, min(map_mib, 256) << 20)
    img, index, sums = lzf_map_output_image(data, offs, algo)
    s1, n = oracle.mt_decompress_bench(oracle.CODEC_LZF, oracle.CHECKSUM_ADLER32, img, index, sums, data.size, cores, reps=1, use_liblz4=False)
    if s1 < 0 or n != data.size:
        return None
    reps = max(1, min(2000, int(target_s / max(s1, 1e-3))))
    s, _ = oracle.mt_decompress_bench(oracle.CODEC_LZF, oracle.CHECKSUM_ADLER32, img, index, sums, data.size, cores, reps=reps, use_liblz4=False)
    t1, _ = oracle.mt_decompress_bench(oracle.CODEC_LZF, oracle.CHECKSUM_ADLER32, img, index, sums, data.size, 1, reps=2, use_liblz4=False)
    return {"value": round(data.size * cores * reps / s / 1e9, 3), "unit": "GB/s", "cores": cores, "kind": "port",
            "sample": f"{cores} threads x {reps} reps x one whole {min(map_mib, 256)} MiB map task ({nparts} partitions), Adler32 validation + "
                      f"compress-lzf chunk walk + C restatement of liblzf 3.6 lzf_decompress; os.cpu_count()={os.cpu_count()}, limit={cores}",
            "single_thread_GBps": round(data.size * 2 / t1 / 1e9, 3), "wall_s": round(s, 2)}


# ---- Zstandard inputs (reduce side only): the map outputs a JVM writer produces with spark.io.compression.codec=zstd ----
# libzstd is a third-party library of the image (the one zstd-jni wraps); it is used here to BUILD the benchmark's input
# and as the CPU baseline, exactly like numpy builds the uncompressed inputs — the product never links it.
class _ZBuf(__import__("ctypes").Structure):
    _fields_ = [("p", __import__("ctypes").c_void_p), ("size", __import__("ctypes").c_size_t), ("pos", __import__("ctypes").c_size_t)]


def _libzstd():
    import ctypes

    z = ctypes.CDLL("libzstd.so.1")
    z.ZSTD_compressBound.restype = ctypes.c_size_t
    z.ZSTD_compressBound.argtypes = [ctypes.c_size_t]
    z.ZSTD_isError.restype = ctypes.c_uint
    z.ZSTD_isError.argtypes = [ctypes.c_size_t]
    z.ZSTD_createCCtx.restype = ctypes.c_void_p
    z.ZSTD_freeCCtx.argtypes = [ctypes.c_void_p]
    z.ZSTD_CCtx_setParameter.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    z.ZSTD_compressStream2.restype = ctypes.c_size_t
    z.ZSTD_compressStream2.argtypes = [ctypes.c_void_p, ctypes.POINTER(_ZBuf), ctypes.POINTER(_ZBuf), ctypes.c_int]
    z.ZSTD_decompress.restype = ctypes.c_size_t
    z.ZSTD_decompress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
    return z


def zstd_map_output_image(data, offs, algo_name: str):
    """(.data image, index, checksums) of one map task under ZStdCompressionCodec: one streaming frame (level 1, 32 KiB
    writes, no content size) per non-empty partition; checksums over the compressed bytes (zlib's Adler32 / CRC32)."""
    import ctypes
    import zlib

    z = _libzstd()
    n = len(offs) - 1
    cctx = z.ZSTD_createCCtx()
    parts = []
    try:
        for p in range(n):
            a, b = int(offs[p]), int(offs[p + 1])
            if b == a:
                parts.append(np.zeros(0, np.uint8))
                continue
            z.ZSTD_CCtx_setParameter(cctx, 100, 1)  # ZSTD_c_compressionLevel = spark.io.compression.zstd.level default
            cap = int(z.ZSTD_compressBound(b - a)) + 1024
            out = np.empty(cap, np.uint8)
            ob = _ZBuf(out.ctypes.data, cap, 0)
            pos = a
            while pos < b:
                k = min(32768, b - pos)
                ib = _ZBuf(data.ctypes.data + pos, k, 0)
                while ib.pos < ib.size:
                    assert not z.ZSTD_isError(z.ZSTD_compressStream2(cctx, ctypes.byref(ob), ctypes.byref(ib), 0))
                pos += k
            ib = _ZBuf(data.ctypes.data, 0, 0)
            while z.ZSTD_compressStream2(cctx, ctypes.byref(ob), ctypes.byref(ib), 2) != 0:
                pass
            parts.append(out[: ob.pos].copy())
    finally:
        z.ZSTD_freeCCtx(cctx)
    index = np.zeros(n + 1, np.int64)
    np.cumsum([q.size for q in parts], out=index[1:])
    fn = zlib.adler32 if algo_name == "adler32" else zlib.crc32
    sums = np.array([fn(q) & 0xFFFFFFFF for q in parts], np.int64)
    return (np.concatenate(parts) if parts else np.zeros(0, np.uint8)), index, sums


def cpu_baseline_zstd_decompress(workload: str, target_s: float, map_mib: int):
    """libzstd's own decoder on the host cores, one partition frame per call, all usable cores (ctypes drops the GIL);
    per-partition checksum validation with zlib in front — what the JVM reader does per task, without JVM / JNI overheads."""
    import zlib
    from concurrent.futures import ThreadPoolExecutor

    gen, nparts, codec, algo = WORKLOADS[workload]
    cores = usable_cores()
    data, offs = make_map_output(workload, 0, min(map_mib, 256) << 20)
    img, index, sums = zstd_map_output_image(data, offs, algo)
    z = _libzstd()
    fn = zlib.adler32 if algo == "adler32" else zlib.crc32

    def one_task(_):
        out = np.empty(data.size, np.uint8)
        for p in range(len(offs) - 1):
            a, b = int(index[p]), int(index[p + 1])
            if b == a:
                continue
            assert fn(img[a:b]) & 0xFFFFFFFF == sums[p]  # (buffer protocol: no copy, zlib drops the GIL)
            z.ZSTD_decompress(out.ctypes.data + int(offs[p]), int(offs[p + 1] - offs[p]), img.ctypes.data + a, b - a)
        return out

    t0 = time.perf_counter()
    one_task(0)
    s1 = time.perf_counter() - t0
    reps = max(1, min(50, int(target_s / max(s1, 1e-3))))
    with ThreadPoolExecutor(cores) as ex:
        t0 = time.perf_counter()
        list(ex.map(one_task, range(cores * reps)))
        s = time.perf_counter() - t0
    return {
        "value": round(data.size * cores * reps / s / 1e9, 3), "unit": "GB/s", "cores": cores, "kind": "reference",
        "sample": f"{cores} threads x {reps} reps x one whole {min(map_mib, 256)} MiB map task ({nparts} partitions = {nparts} zstd frames), "
                  f"zlib {algo} validation + libzstd 1.4.8 ZSTD_decompress per frame (the library zstd-jni wraps); JVM/JNI overheads "
                  f"not included; os.cpu_count()={os.cpu_count()}, cgroup/affinity limit={cores}",
        "single_thread_GBps": round(data.size / s1 / 1e9, 3), "wall_s": round(s, 2),
    }


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="terasort-10g-200p-lz4", choices=sorted(WORKLOADS))
    ap.add_argument("--map-mib", type=int, default=128, help="uncompressed MiB per map task (input split)")
    ap.add_argument("--maps-per-gpu", type=int, default=8)
    ap.add_argument("--task-threads", type=int, default=0,
                    help="concurrent task threads per GPU, one s3s_ctx (HIP stream) each — an executor runs "
                         "several tasks at once (spark.executor.cores = 4 in the reference's examples); 0 = 4 for compress "
                         "(measured 78.6 GB/s against 76.8 with two and 75.9 with one: the hash / assemble / checksum "
                         "stages of one call overlap the other calls' codec kernels), 2 for decompress (round 6: 464 vs 416 GB/s "
                         "with one, 428 with four)")
    ap.add_argument("--batch", type=int, default=-1,
                    help="map tasks per library call (s3s_compress_map_outputs_batch_device: one codec launch over the "
                         "chunks of all of them, one stream sync); -1 = all of a task thread's map tasks, 0 = one call "
                         "per map task (s3s_compress_map_output_device)")
    ap.add_argument("--lz4-variant", type=int, default=-1, help="S3S_OPT_LZ4_VARIANT override")
    ap.add_argument("--lz4-decode-variant", type=int, default=-1, help="S3S_OPT_LZ4_DECODE_VARIANT override")
    ap.add_argument("--direction", default="compress", choices=["compress", "decompress"],
                    help="decompress = reduce-side verify + decode of the same map outputs (batch-fetch range)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target wall time of the CPU leg")
    ap.add_argument("--verify", action="store_true", help="check one map task against the oracle first")
    ap.add_argument("--no-image-check", action="store_true",
                    help="skip the untimed image_verified check after a compress leg (profiling runs: it adds checksum launches)")
    ap.add_argument("--secondary", dest="secondary", action="store_true", default=None,
                    help="after the headline, run short (3-step) passes of the other BASELINE.json configurations and attach "
                         "them as `secondary` to the JSON line (default: on for the plain N=1 headline command)")
    ap.add_argument("--no-secondary", dest="secondary", action="store_false")
    ap.add_argument("--host-path-only", action="store_true", help="run only the host-buffer leg (run_host_path) and print it")
    ap.add_argument("--hbm-stages-only", action="store_true", help="run only the HBM-bound stage lines (run_hbm_stages) and print them")
    ap.add_argument("--full-line", action="store_true", help="also print the full (long) record of the run as an earlier stdout line")
    ap.add_argument("--dry-run", action="store_true",
                    help="no timing: every rank reports (rank, local rank, device, its map ids); rank 0 checks that "
                         "mapId %% nGPU covers every map task exactly once and prints the table as one JSON line "
                         "(backend nccl on GPUs, gloo without)")
    return ap.parse_args()


_GEN_CACHE = {}  # (generator, partitions, map id, bytes) -> host arrays: the secondary passes reuse the headline's inputs


_SKEW_CACHE = {}


def make_map_output(workload: str, map_id: int, n_bytes: int):
    from s3shuffle import datagen

    gen, nparts, _, _ = WORKLOADS[workload]
    key = (gen, nparts, map_id, n_bytes)
    if key in _GEN_CACHE:
        return _GEN_CACHE[key]
    if gen == "terasort":
        r = datagen.terasort_map_output(n_bytes, nparts, seed=2, map_id=map_id)
    elif gen == "tpcds":
        r = datagen.tpcds_wide_map_output(n_bytes, nparts, seed=3, map_id=map_id)
    else:
        # single-partition blocks are a prefix of the same record stream (records are generated by index): the block-size
        # sweep slices the largest block generated so far for this map task instead of regenerating it
        big = _SKEW_CACHE.get(map_id)
        if big is None or big.size < n_bytes:
            big = datagen.skew_block(n_bytes, "terasort", seed=5, map_id=map_id)[0]
            _SKEW_CACHE[map_id] = big
        return big[:n_bytes], np.array([0, n_bytes], dtype=np.int64)
    if n_bytes <= (256 << 20):
        _GEN_CACHE[key] = r
    return r


def _kernel_stamp():
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from src_stamp import kernel_sources_sha256

        return kernel_sources_sha256(ROOT)
    except Exception:
        return None


def usable_cores() -> int:
    """Host threads this process may really use: min(affinity mask, cgroup cpu.max quota)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    return n


_LIB_NAMES = {
    "lz4": ("liblz4 1.9.3 LZ4_compress_default (the code lz4-java JNI binds)", "liblz4 1.9.3 LZ4_decompress_safe (the code lz4-java JNI binds)"),
    "snappy": ("libsnappy 1.1.8 snappy_compress (the C++ code snappy-java binds, there in version 1.1.10)",
               "libsnappy 1.1.8 snappy_uncompress (the C++ code snappy-java binds, there in version 1.1.10)"),
}


def _simd_names(simd: int) -> str:
    return ("CRC32 by PCLMULQDQ folding" if simd & 1 else "CRC32 slice-by-8") + " / " + \
           ("Adler32 by SSSE3 PSADBW+PMADDUBSW" if simd & 2 else "Adler32 byte loop") + " (the JVM runs both as intrinsics)"


def cpu_baseline(workload: str, target_s: float, map_mib: int = 128):
    """The CPU path timed beside the GPU one: one map task per host thread (how Spark runs the
    reference: one task per executor core), each doing LZ4Block framing around liblz4 1.9.3's
    LZ4_compress_default + xxh32 + per-partition checksum + index (oracle/s3s_oracle_mt.c)."""
    from oracle import binding as oracle

    _, nparts, codec, algo = WORKLOADS[workload]
    cores = usable_cores()
    sample_mib = min(map_mib, 256)  # the whole map task the GPU leg runs (not a slice); 1 GiB blocks: the first 256 MiB
    parts = nparts
    data, offs = make_map_output(workload, 0, sample_mib << 20)
    algo_id = {"adler32": oracle.CHECKSUM_ADLER32, "crc32": oracle.CHECKSUM_CRC32}[algo]
    codec_o = oracle.CODEC_LZ4 if codec == "lz4" else oracle.CODEC_SNAPPY
    have_lib = bool(oracle.lib().s3o_mt_have_liblz4()) if codec == "lz4" else bool(oracle.lib().s3o_mt_have_libsnappy())
    simd = int(oracle.lib().s3o_simd_available())
    # calibrate with one rep, then size the run to ~target_s
    s1, _ = oracle.mt_compress_bench(codec_o, algo_id, data, offs, cores, reps=1)
    reps = max(1, min(400, int(target_s / max(s1, 1e-3))))
    s, _ = oracle.mt_compress_bench(codec_o, algo_id, data, offs, cores, reps=reps)
    value = data.size * cores * reps / s / 1e9
    t1, _ = oracle.mt_compress_bench(codec_o, algo_id, data, offs, 1, reps=2)
    one = data.size * 2 / t1 / 1e9
    return {
        "value": round(value, 3), "unit": "GB/s", "cores": cores, "kind": "reference-lib" if have_lib else "port",
        "sample": f"{cores} threads x {reps} reps x one whole {sample_mib} MiB map task of the workload "
                  f"({parts} partitions), {codec}+{algo}, map-side compress+checksum; block compressor = "
                  f"{_LIB_NAMES[codec][0] if have_lib else 'oracle restatement'}; checksums = {_simd_names(simd)}; "
                  f"framing / index by oracle/s3s_oracle_mt.c; "
                  f"JVM/JNI overheads not included (upper bound on the reference path); "
                  f"os.cpu_count()={os.cpu_count()}, cgroup/affinity limit={cores}",
        "sample_short": f"{cores} thr x {reps} reps x one {sample_mib} MiB map task ({parts} parts), {codec}+{algo}; "
                        f"{'lib' + codec if have_lib else 'oracle port'} + SIMD checksums; no JVM/JNI overhead",
        "single_thread_GBps": round(one, 3), "wall_s": round(s, 2),
    }


def cpu_baseline_decompress(workload: str, target_s: float, map_mib: int = 128, block_size: int = 0):
    """Reduce side on the host cores: one fetched block range per thread, per-partition checksum validation
    (S3ChecksumValidationStream) + LZ4BlockInputStream over liblz4 1.9.3's LZ4_decompress_safe + xxh32 frame
    checks (oracle/s3s_oracle_mt.c) — what the JVM reader does per task, without JVM/JNI overheads."""
    from oracle import binding as oracle
    from s3shuffle import datagen

    gen, nparts, codec, algo = WORKLOADS[workload]
    cores = usable_cores()
    sample_mib = min(map_mib, 256)
    parts = nparts
    data, offs = make_map_output(workload, 0, sample_mib << 20)
    algo_id = {"adler32": oracle.CHECKSUM_ADLER32, "crc32": oracle.CHECKSUM_CRC32}[algo]
    codec_o = oracle.CODEC_LZ4 if codec == "lz4" else oracle.CODEC_SNAPPY
    if block_size:  # a JVM writer's larger LZ4 blocks: the same liblz4-written image the GPU leg decodes
        img, index, sums = jvm_lz4_map_output_image(data, offs, algo, block_size)
    else:
        img, index, sums = oracle.compress_map_output(codec_o, algo_id, data, offs)
    have_lib = bool(oracle.lib().s3o_mt_have_liblz4()) if codec == "lz4" else bool(oracle.lib().s3o_mt_have_libsnappy())
    simd = int(oracle.lib().s3o_simd_available())
    s1, n = oracle.mt_decompress_bench(codec_o, algo_id, img, index, sums, data.size, cores, reps=1)
    if s1 < 0 or n != data.size:
        return None
    reps = max(1, min(2000, int(target_s / max(s1, 1e-3))))
    s, _ = oracle.mt_decompress_bench(codec_o, algo_id, img, index, sums, data.size, cores, reps=reps)
    t1, _ = oracle.mt_decompress_bench(codec_o, algo_id, img, index, sums, data.size, 1, reps=4)
    return {
        "value": round(data.size * cores * reps / s / 1e9, 3), "unit": "GB/s", "cores": cores, "kind": "reference-lib" if have_lib else "port",
        "sample": f"{cores} threads x {reps} reps x one whole {sample_mib} MiB map task of the workload ({parts} partitions), "
                  f"{codec}+{algo}, reduce-side verify+decompress; block decoder = "
                  f"{_LIB_NAMES[codec][1] if have_lib else 'oracle restatement'}; checksums = {_simd_names(simd)}; "
                  f"JVM/JNI overheads not included; os.cpu_count()={os.cpu_count()}, cgroup/affinity limit={cores}",
        "sample_short": f"{cores} thr x {reps} reps x one {sample_mib} MiB map task ({parts} parts), {codec}+{algo} verify+decode; "
                        f"{'lib' + codec if have_lib else 'oracle port'} + SIMD checksums; no JVM/JNI overhead",
        "single_thread_GBps": round(data.size * 4 / t1 / 1e9, 3), "wall_s": round(s, 2),
    }


def verify_images(workload, codec_id, algo_id, codec, tasks, map_ids, n_bytes, last_result) -> bool:
    """Untimed self-check of a compress leg (rank 0): for every map task, CRC32 of dst[:total] computed ON THE DEVICE by the
    library (s3s_checksum_ranges_device) == zlib.crc32 of the oracle's .data image of the same map output, and the index and
    per-partition checksums the last timed call returned == the oracle's.  The oracle is the checker here, never timed."""
    import zlib

    import s3shuffle
    from oracle import binding as oracle

    ok = [None] * len(tasks)

    def one(i):
        data, offs = make_map_output(workload, map_ids[i], n_bytes)
        r_img, r_index, r_sums = oracle.compress_map_output(codec_id, algo_id, data, offs)
        total, index, sums = last_result[i]
        if total != r_img.size or not np.array_equal(index, r_index) or not np.array_equal(sums, r_sums):
            return
        ok[i] = zlib.crc32(r_img) & 0xFFFFFFFF

    th = [threading.Thread(target=one, args=(i,)) for i in range(len(tasks))]
    [t.start() for t in th]
    [t.join() for t in th]
    for i, t in enumerate(tasks):
        if ok[i] is None:
            return False
        got = codec.checksum_ranges_device(s3shuffle.CHECKSUM_CRC32, t["dst"].data_ptr(), np.array([0, last_result[i][0]], np.int64))
        if int(got[0]) != ok[i]:
            return False
    return True


def dry_run(args, rank: int, local_rank: int, world: int, launched: bool):
    """Validate rank <-> device <-> mapId without touching the codec (S3ShuffleDispatcher.scala:142-143 rule)."""
    import torch
    from s3shuffle import sharding

    has_gpu = torch.cuda.is_available()
    mine = {"rank": rank, "local_rank": local_rank, "device": f"cuda:{local_rank}" if has_gpu else "cpu",
            "visible_devices": torch.cuda.device_count() if has_gpu else 0,
            "map_ids": [int(m) for m in sharding.map_ids_for_rank(rank, world, args.maps_per_gpu)]}
    table = [mine]
    if launched:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if has_gpu:
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo")
        table = [None] * world
        dist.all_gather_object(table, mine)
        t = torch.ones(1, device=f"cuda:{local_rank}" if has_gpu else "cpu")
        dist.all_reduce(t)  # the collective the timed run uses (MAX of elapsed, SUM of bytes)
        assert int(t.item()) == world
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        all_ids = sorted(m for r in table for m in r["map_ids"])
        ok = all_ids == list(range(world * args.maps_per_gpu))
        ok = ok and all(sharding.device_for_map(m, world) == r["rank"] for r in table for m in r["map_ids"])
        ok = ok and (not has_gpu or all(r["local_rank"] < max(r["visible_devices"], 1) for r in table))
        print(json.dumps({"dry_run": True, "world": world, "maps_per_gpu": args.maps_per_gpu,
                          "backend": ("nccl" if has_gpu else "gloo") if launched else None, "ranks": table, "ok": ok}))
        if not ok:
            raise SystemExit(2)


def run_workload(args, rank: int, local_rank: int, world: int, dist):
    """One measured workload: W warmup + K timed steps of the hot path over this rank's map tasks.  Returns the
    result dict on rank 0 (None elsewhere)."""
    import torch
    import s3shuffle
    from s3shuffle import sharding

    gen, nparts, codec_name, algo_name = WORKLOADS[args.workload]
    codec_id = {"lz4": s3shuffle.CODEC_LZ4, "snappy": s3shuffle.CODEC_SNAPPY, "zstd": s3shuffle.CODEC_ZSTD, "lzf": s3shuffle.CODEC_LZF}[codec_name]
    algo_id = {"adler32": s3shuffle.CHECKSUM_ADLER32, "crc32": s3shuffle.CHECKSUM_CRC32}[algo_name]
    if codec_name in ("zstd", "lzf") and args.direction != "decompress":
        raise SystemExit("zstd / lzf are decode-only on the GPU path (compression stays on the JVM codec): use --direction decompress")

    # ---- this rank's shard: map tasks with mapId % nGPU == rank --------------------------------
    map_ids = sharding.map_ids_for_rank(rank, world, args.maps_per_gpu)
    assert all(sharding.device_for_map(m, world) == rank for m in map_ids)
    n_bytes = args.map_mib << 20
    outputs = [None] * len(map_ids)

    def _gen(i):
        outputs[i] = make_map_output(args.workload, map_ids[i], n_bytes)

    th = [threading.Thread(target=_gen, args=(i,)) for i in range(len(map_ids))]
    [t.start() for t in th]
    [t.join() for t in th]

    dev = torch.device("cuda", local_rank)
    tasks = []
    # defaults: compress 4 task threads (spark.executor.cores of the reference's examples); decompress 2 — each a batched call over
    # half of the step's fetched ranges, so one call's frame checks / discovery / checksums overlap the other's decode kernel
    # (round 6, profiles/r06b_*: 8 TeraSort tasks 416 GB/s with one thread, 464 with two, 428 with four)
    task_threads = args.task_threads if args.task_threads > 0 else (2 if args.direction == "decompress" else 4)
    if args.task_threads <= 0 and world > 1:
        # one process per GPU on ONE host: the ranks share its cores; with fewer than four usable cores per rank the task threads of
        # the ranks would queue for them (two threads x four map tasks per call measured 107.8 against 109 GB/s with four x two: r05q)
        task_threads = max(1, min(task_threads, max(2, usable_cores() // world)))
    n_threads = max(1, min(task_threads, len(map_ids)))
    codecs = [s3shuffle.Codec(local_rank) for _ in range(n_threads)]
    for c in codecs:
        c.set_option(s3shuffle.codec.OPT_PROFILE, 1)
        if args.lz4_variant >= 0:
            c.set_option(s3shuffle.codec.OPT_LZ4_VARIANT, args.lz4_variant)
        if args.lz4_decode_variant >= 0:
            c.set_option(s3shuffle.codec.OPT_LZ4_DECODE_VARIANT, args.lz4_decode_variant)
    zstd_images = None  # (host-built images of JVM writers: zstd frames, or LZ4Block frames of a larger block size)
    jvm_block = JVM_LZ4_BLOCK.get(args.workload)
    if jvm_block and args.direction != "decompress":
        raise SystemExit("LZ4 blocks above 32 KiB are decoded, not written, by the GPU path: use --direction decompress")
    if codec_name in ("zstd", "lzf") or jvm_block:  # the JVM-written objects: built on the host with libzstd / liblz4 (input generation, untimed)
        zstd_images = [None] * len(outputs)

        def _zimg(i):
            if jvm_block:
                zstd_images[i] = jvm_lz4_map_output_image(outputs[i][0], outputs[i][1], algo_name, jvm_block)
            elif codec_name == "lzf":
                zstd_images[i] = lzf_map_output_image(outputs[i][0], outputs[i][1], algo_name)
            else:
                zstd_images[i] = zstd_map_output_image(outputs[i][0], outputs[i][1], algo_name)

        zt = [threading.Thread(target=_zimg, args=(i,)) for i in range(len(outputs))]
        [t.start() for t in zt]
        [t.join() for t in zt]
    for ti, (data, offs) in enumerate(outputs):
        d_src = torch.from_numpy(data).to(dev)
        cap = codecs[0].max_compressed_size(codec_id, offs) if zstd_images is None else int(zstd_images[ti][0].size) + 64
        d_dst = torch.empty(cap, dtype=torch.uint8, device=dev)
        tasks.append({"src": d_src, "dst": d_dst, "offs": offs, "cap": cap, "u": int(data.size)})
    torch.cuda.synchronize()

    if args.verify and rank == 0:
        from oracle import binding as oracle

        data, offs = outputs[0]
        t = tasks[0]
        total, index, sums = codecs[0].compress_map_output_device(codec_id, algo_id, t["src"].data_ptr(), offs,
                                                                  t["dst"].data_ptr(), t["cap"])
        r_img, r_index, r_sums = oracle.compress_map_output(codec_id, algo_id, data, offs)
        img = t["dst"][:total].cpu().numpy()
        assert np.array_equal(index, r_index) and np.array_equal(sums, r_sums) and np.array_equal(img, r_img), \
            "GPU output differs from the oracle"
        print("verify: bit-exact vs oracle", file=sys.stderr)
    outputs = None  # host copies are not needed any more

    stage = {"codec": 0.0, "hash": 0.0, "assemble": 0.0, "checksum": 0.0, "total": 0.0, "launches": 0}
    comp_bytes = [0] * len(tasks)
    last_result = [None] * len(tasks)  # (total, index, checksums) of each map task's most recent compress call
    lock = threading.Lock()

    decompress = args.direction == "decompress"
    per_thread = (len(tasks) + n_threads - 1) // n_threads
    batch_n = per_thread if args.batch < 0 else min(args.batch, per_thread)
    if decompress:
        # reduce side: every map output is first compressed once (untimed); the timed step verifies the
        # per-partition checksums and decodes the whole range [index[0], index[N]) — a
        # ShuffleBlockBatchId-style batch fetch of all partitions of the map output
        for i, t in enumerate(tasks):
            if zstd_images is not None:
                img, index, sums = zstd_images[i]
                total = int(img.size)
                t["dst"][:total].copy_(torch.from_numpy(img))
            else:
                total, index, sums = codecs[0].compress_map_output_device(codec_id, algo_id, t["src"].data_ptr(), t["offs"],
                                                                          t["dst"].data_ptr(), t["cap"])
            t["total"], t["index"], t["sums"] = total, index, sums
            t["out"] = torch.empty(t["u"], dtype=torch.uint8, device=dev)
            comp_bytes[i] = total
        torch.cuda.synchronize()

    def run_steps(n_steps: int, record: bool):
        """n_steps passes over all map tasks of this rank.  Every task thread runs its own tasks n_steps times back to
        back (an executor core that keeps taking map tasks); the threads are joined once, at the end, so the
        timed region holds exactly n_steps steps of work without a drain between them."""
        def worker(tid):
            for _ in range(n_steps):
                one_pass(tid)

        def one_pass(tid):
            c = codecs[tid]
            acc = [0.0] * 5
            n = 0
            mine = list(range(tid, len(tasks), n_threads))
            if batch_n > 0 and not decompress:
                for b0 in range(0, len(mine), batch_n):
                    grp = mine[b0:b0 + batch_n]
                    res = c.compress_map_outputs_batch_device(
                        codec_id, algo_id, [(tasks[i]["src"].data_ptr(), tasks[i]["offs"], tasks[i]["dst"].data_ptr(),
                                             tasks[i]["cap"]) for i in grp])
                    for i, r in zip(grp, res):
                        comp_bytes[i] = r[0]
                        last_result[i] = r
                    if record:
                        acc[0] += c.stage_ms(s3shuffle.codec.STAGE_CODEC)
                        acc[1] += c.stage_ms(s3shuffle.codec.STAGE_HASH)
                        acc[2] += c.stage_ms(s3shuffle.codec.STAGE_ASSEMBLE)
                        acc[3] += c.stage_ms(s3shuffle.codec.STAGE_CHECKSUM)
                        acc[4] += c.stage_ms(s3shuffle.codec.STAGE_TOTAL)
                        n += 1
                mine = []
            if batch_n > 1 and decompress:
                for b0 in range(0, len(mine), batch_n):
                    grp = mine[b0:b0 + batch_n]
                    res = c.decompress_ranges_batch_device(
                        codec_id, algo_id, [(tasks[i]["dst"].data_ptr(), tasks[i]["total"], tasks[i]["index"], tasks[i]["sums"],
                                             tasks[i]["out"].data_ptr(), tasks[i]["u"]) for i in grp])
                    for i, r in zip(grp, res):
                        assert r[0] == 0 and r[1] == tasks[i]["u"]
                    if record:
                        acc[0] += c.stage_ms(s3shuffle.codec.STAGE_CODEC)
                        acc[1] += c.stage_ms(s3shuffle.codec.STAGE_HASH)
                        acc[3] += c.stage_ms(s3shuffle.codec.STAGE_CHECKSUM)
                        acc[4] += c.stage_ms(s3shuffle.codec.STAGE_TOTAL)
                        n += 1
                mine = []
            for i in mine:
                t = tasks[i]
                if decompress:
                    got = c.decompress_range_device(codec_id, algo_id, t["dst"].data_ptr(), t["total"], t["index"],
                                                    t["sums"], t["out"].data_ptr(), t["u"])
                    assert got == t["u"]
                else:
                    last_result[i] = c.compress_map_output_device(codec_id, algo_id, t["src"].data_ptr(), t["offs"],
                                                                  t["dst"].data_ptr(), t["cap"])
                    comp_bytes[i] = last_result[i][0]
                if record:
                    acc[0] += c.stage_ms(s3shuffle.codec.STAGE_CODEC)
                    acc[1] += c.stage_ms(s3shuffle.codec.STAGE_HASH)
                    acc[2] += c.stage_ms(s3shuffle.codec.STAGE_ASSEMBLE)
                    acc[3] += c.stage_ms(s3shuffle.codec.STAGE_CHECKSUM)
                    acc[4] += c.stage_ms(s3shuffle.codec.STAGE_TOTAL)
                    n += 1
            if record:
                with lock:
                    for k, name in enumerate(("codec", "hash", "assemble", "checksum", "total")):
                        stage[name] += acc[k]
                    stage["launches"] += n

        if n_threads == 1:
            worker(0)
        else:
            ths = [threading.Thread(target=worker, args=(j,)) for j in range(n_threads)]
            [t.start() for t in ths]
            [t.join() for t in ths]

    run_steps(args.warmup, False)

    # Python's cyclic collector runs when allocation counts say so - in a process that has just dropped gigabytes of numpy /
    # torch objects that is a 40 ms pause, and it fell into the timed region of the short secondary passes (measured: one
    # 44 ms library call among ten of 3.3 ms).  Collect now, keep the collector off while the K steps are timed.
    import gc

    gc.collect()
    gc_was_on = gc.isenabled()
    gc.disable()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(args.steps, True)
    torch.cuda.synchronize()  # library calls already synchronise their own stream before returning
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if gc_was_on:
        gc.enable()

    # reduce side: the timed calls check status and size only, and the per-partition checksum covers the COMPRESSED bytes —
    # so every decoded map output is compared with its source, byte for byte, once, after the timed region (untimed)
    bytes_verified = None
    if decompress:
        bytes_verified = all(bool(torch.equal(t["out"], t["src"])) for t in tasks)
        if not bytes_verified:
            raise SystemExit(f"{args.workload}: decoded bytes differ from the source")

    # map side: the timed calls return sizes only — so, once, after the timed region (untimed), the .data image every task's LAST
    # timed call left in HBM is hashed on the device (the library's own CRC32 over dst[:total]) and compared, together with the
    # index and the checksums, with the oracle's image of the same map output computed on the host
    image_verified = None
    if not decompress and not args.no_image_check:
        image_verified = verify_images(args.workload, codec_id, algo_id, codecs[0], tasks, map_ids, n_bytes, last_result)
        if dist:  # every rank checks its own map outputs; the line reports the AND over ranks
            iv = torch.tensor([int(image_verified)], dtype=torch.int32, device=dev)
            dist.all_reduce(iv, op=dist.ReduceOp.MIN)
            image_verified = bool(iv.item())
        if not image_verified:
            raise SystemExit(f"{args.workload}: the .data image / index / checksums of the timed calls differ from the oracle's")

    u_rank = sum(t["u"] for t in tasks)
    c_rank = sum(comp_bytes)
    elapsed_rank = elapsed
    per_rank = [{"rank": rank, "device": local_rank, "elapsed_s": round(elapsed_rank, 5),
                 "GBps": round(u_rank * args.steps / elapsed_rank / 1e9, 3), "map_tasks": len(tasks)}]
    rccl_ranks = 1
    if dist:
        rccl_ranks = int(dist.get_world_size())
        # per-rank balance for the scaling record: every rank's own elapsed time and bytes, gathered with
        # one all_gather of three doubles per rank — after the timed region, so it costs the measurement nothing
        mine = torch.tensor([float(elapsed_rank), float(u_rank), float(len(tasks))], dtype=torch.float64, device=dev)
        allr = [torch.zeros(3, dtype=torch.float64, device=dev) for _ in range(rccl_ranks)]
        dist.all_gather(allr, mine)
        per_rank = [{"rank": r, "device": r % max(torch.cuda.device_count(), 1), "elapsed_s": round(float(v[0]), 5),
                     "GBps": round(float(v[1]) * args.steps / max(float(v[0]), 1e-9) / 1e9, 3), "map_tasks": int(v[2])}
                    for r, v in enumerate(allr)]
    if dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        agg = torch.tensor([u_rank, c_rank], dtype=torch.int64, device=dev)
        dist.all_reduce(agg, op=dist.ReduceOp.SUM)
        u_all, c_all = int(agg[0].item()), int(agg[1].item())
    else:
        u_all, c_all = u_rank, c_rank

    out = None
    if rank == 0:
        lz4_parse = None
        if codec_name == "lz4" and not decompress:
            setting = codecs[0].get_option(s3shuffle.codec.OPT_LZ4_VARIANT)
            used = sorted({int(c.get_option(s3shuffle.codec.OPT_LZ4_VARIANT_USED)) for c in codecs})
            lz4_parse = {"setting": "auto" if setting == 9 else int(setting), "ran_in_last_call": used}
        value = u_all * args.steps / elapsed / 1e9
        launches = max(stage["launches"], 1)
        codec_ms = stage["codec"] / launches          # the LZ4 block-compress kernel alone
        tasks_per_launch = batch_n if batch_n > 0 else 1
        u_launch = u_rank / len(tasks) * tasks_per_launch
        c_launch = c_rank / len(tasks) * tasks_per_launch
        if codec_name == "lz4" and not decompress:
            payload_launch = c_launch - 21.0 * (u_launch / 32768.0 + nparts * tasks_per_launch)  # frame headers come later
        else:
            payload_launch = c_launch
        alg_bytes = u_launch + max(payload_launch, 0.0)  # chunk bytes read + payload bytes written (or the reverse)
        achieved = alg_bytes / (codec_ms * 1e-3) / 1e9 if codec_ms > 0 else 0.0
        out = {
            "metric": "shuffle_block_verify_decompress_throughput" if decompress else "shuffle_block_compress_checksum_throughput",
            "value": round(value, 3),
            "unit": "GB/s",
            "n_gpus": world,
            "rccl_ranks": rccl_ranks if dist else 0,  # ranks the RCCL process group saw (0 = plain `python bench.py`, no group)
            "per_rank": per_rank,                      # each rank's own clock over its own map tasks (value uses the MAX)
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": args.workload,
                "generator": {"terasort": "TeraGen-like 100-byte records, seed 2", "tpcds": "UnsafeRow-like wide rows, seed 3",
                              "skew": "single-partition TeraGen-like block, seed 5"}[gen],
                "direction": args.direction,
                "codec": ("lz4 (decode only: LZ4Block frames of %d KiB blocks written by liblz4 1.9.3 on the host, as a JVM writer with that spark.io.compression.lz4.blockSize)" % (jvm_block >> 10)) if jvm_block
                         else "lz4 (LZ4Block frames, 32 KiB blocks; payload bit-exact with liblz4 1.9.3 LZ4_compress_default, framing restated from lz4-java 1.8.0)" if codec_name == "lz4"
                         else "snappy (SnappyOutputStream framing, 32 KiB blocks, byte-exact with snappy 1.1.8)" if codec_name == "snappy"
                         else "lzf (decode only: LZFOutputStream chunks of <= 65 535 bytes around liblzf 3.6 blocks, as compress-lzf writes them)" if codec_name == "lzf"
                         else "zstd (decode only: libzstd 1.4.8 streaming frames, level 1, one per partition, as zstd-jni writes them)",
                "checksum": algo_name,
                "partitions_per_map_task": nparts,
                "map_task_bytes": tasks[0]["u"],
                "map_tasks_per_gpu": len(tasks),
                "uncompressed_bytes_per_step": u_all,
                "compressed_bytes_per_step": c_all,
                "compression_ratio": round(u_all / max(c_all, 1), 4),
                "sharding": "mapId % nGPU, no data-path collective",
                "task_threads_per_gpu": n_threads,
                "map_tasks_per_library_call": tasks_per_launch,
                "lz4_parse_variant": lz4_parse,
                "inputs": "resident in HBM before the timed region; index/checksums returned to host per map task",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": ("zstd decode pass (a sequence wavefront per partition frame + a literal wavefront per two, serial entropy stages)" if codec_name == "zstd" else
                           "%s batch decoder (one wavefront per frame, one sequence per lane)" % codec_name if decompress
                           else "%s_compress (one wavefront per 32 KiB block)" % codec_name),
                "achieved": round(achieved, 3),
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 6),
                "traffic": None,
                "avg_launch_ms": round(codec_ms, 4),
                "concurrent_launches": n_threads,  # one per task thread / stream; they share the GPU
                "launch_shape": ("persistent grid: 10 wavefronts per CU, 5 when map-side calls overlap (two launches resident "
                                 "together) - avg_launch_ms is the duration of ONE such launch while the others run"
                                 if codec_name == "lz4" and not decompress else "one workgroup (wavefront) per block / frame"),
                "achieved_all_streams": round(achieved * n_threads, 3),
                "algorithmic_bytes_per_launch": int(alg_bytes),
                "frac_of_copy_ceiling": round(achieved / HBM_COPY_CEILING_GBPS, 6),
                "whole_path_read_frac": round(value / world / HBM_PEAK_GBPS, 6),
            },
            "stages_ms_per_library_call": {k: round(stage[k] / launches, 4) for k in ("hash", "codec", "assemble", "checksum", "total")},
        }
        if image_verified is not None:
            out["image_verified"] = bool(image_verified)  # CRC32 of every task's .data image in HBM + index + checksums == the oracle's, after the timed region
        if decompress:
            out["bytes_verified"] = bool(bytes_verified)  # torch.equal(decoded, source) for every map task of the step, after the timed region
        if world == 1 and not args.no_cpu_baseline:
            if codec_name == "zstd":
                cb = cpu_baseline_zstd_decompress(args.workload, args.cpu_seconds, args.map_mib)
            elif codec_name == "lzf":
                cb = cpu_baseline_lzf_decompress(args.workload, args.cpu_seconds, args.map_mib)
            else:
                if decompress:
                    cb = cpu_baseline_decompress(args.workload, args.cpu_seconds, args.map_mib, JVM_LZ4_BLOCK.get(args.workload, 0))
                else:
                    cb = cpu_baseline(args.workload, args.cpu_seconds, args.map_mib)
            out["cpu_baseline"] = cb
            if cb:
                out["speedup_vs_cpu_all_cores"] = round(value / cb["value"], 3)
                out["speedup_vs_cpu_1_core"] = round(value / cb["single_thread_GBps"], 3)
        else:
            out["cpu_baseline"] = None
        # roofline.traffic = HBM bytes per launch of the dominant kernel from the PMC counters (2 x FETCH_SIZE + WRITE_SIZE KB,
        # gfx950 correction as in tools/profile_report.py).  Counters cannot be collected inside a timed run (rocprofv3 serialises
        # the kernels), so the figure comes from the tracked PMC pass of the SAME kernels: profiles/traffic_latest.json names
        # the sha256 of the kernel sources each entry was taken with, and it is used only when that equals the stamp of the
        # sources this library was built from (tools/src_stamp.py); otherwise traffic stays null and the entry is only cited.
        stamp = _kernel_stamp()
        out["kernel_sources_sha256"] = stamp
        if os.environ.get("S3S_BENCH_TRAFFIC_BYTES"):
            out["roofline"]["traffic"] = int(os.environ["S3S_BENCH_TRAFFIC_BYTES"])
            out["roofline"]["traffic_source"] = os.environ.get("S3S_BENCH_TRAFFIC_SOURCE", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of this command")
        else:
            traffic_file = os.path.join(ROOT, "profiles", "traffic_latest.json")
            if os.path.exists(traffic_file):
                try:
                    ref = json.load(open(traffic_file))
                    key = f"{args.workload}:{args.direction}"
                    if key in ref:
                        e = ref[key]
                        if stamp and e.get("kernel_sources_sha256") == stamp and e.get("map_tasks_per_launch"):
                            out["roofline"]["traffic"] = int(e["hbm_bytes_per_launch"] * tasks_per_launch / e["map_tasks_per_launch"])
                            out["roofline"]["traffic_over_algorithmic"] = round(out["roofline"]["traffic"] / max(alg_bytes, 1.0), 3)
                            out["roofline"]["traffic_source"] = (
                                f"{e.get('source')}; PMC pass of the same kernel sources (sha256 {stamp[:12]}), "
                                f"{e['map_tasks_per_launch']} map task(s) per profiled launch scaled to {tasks_per_launch}")
                        else:
                            out["roofline"]["traffic_reference"] = dict(e, note="PMC pass of OTHER kernel sources than this run's (tracked under profiles/); not used as traffic")
                except Exception:
                    pass
    for c in codecs:
        c.close()
    tasks.clear()
    torch.cuda.synchronize()
    return out if rank == 0 else None


SECONDARY = [
    # (label, workload, direction, map MiB, map tasks): BASELINE.json configs[2..4] and the reduce side, 3 timed steps each
    ("tpcds-wide-snappy:compress", "tpcds-wide-100g-200p-snappy", "compress", 128, 4),
    ("tpcds-wide-lz4:compress", "tpcds-wide-100g-200p-lz4", "compress", 128, 4),
    ("terasort-2000p-lz4-crc32:compress", "terasort-100g-2000p-lz4-crc32", "compress", 128, 4),
    ("terasort-200p-lz4:decompress", "terasort-10g-200p-lz4", "decompress", 128, 8),  # (round 6: the headline's 8 map outputs, two task threads x 4)
    ("tpcds-wide-snappy:decompress", "tpcds-wide-100g-200p-snappy", "decompress", 128, 8),
    ("skew-1gib-lz4:compress", "skew-1part-lz4", "compress", 1024, 1),
    ("skew-1gib-lz4:decompress", "skew-1part-lz4", "decompress", 1024, 2),  # two fetched 1 GiB blocks in flight, one per task thread
    # zstd: 16 map outputs, two task threads x 8 (round 6, profiles/r06j_*: the kernel is saturated from 1 600 frames on — four
    # workgroups per CU by LDS and VGPRs — so more frames per call add nothing (8 / 16 / 32 per call: 60.3 / 60.4 / 63.0 GB/s kernel),
    # but a second call's size pass and compaction overlap the first's decode: 49.3 -> 55.3 and 17.2 -> 22.8 GB/s)
    ("terasort-200p-zstd:decompress", "terasort-10g-200p-zstd", "decompress", 128, 16),
    ("tpcds-wide-zstd:decompress", "tpcds-wide-100g-200p-zstd", "decompress", 128, 16),  # 3.3 x the sequences per byte: the decoder's weak side, reported
    ("terasort-200p-lz4-256k-blocks:decompress", "terasort-10g-200p-lz4-256k", "decompress", 128, 8),  # round 4: frames above 32 KiB, batch decoder
    ("terasort-2000p-zstd:decompress", "terasort-100g-2000p-zstd", "decompress", 128, 4),  # 64 KiB frames, one zstd block each
    ("terasort-200p-lzf:decompress", "terasort-10g-200p-lzf", "decompress", 128, 8),  # LZFOutputStream chunks written by liblzf
]


def run_secondaries(args, rank: int, local_rank: int):
    """Short driver-visible passes of the other configurations: same code path as the headline
    (run_workload), 8 timed steps after 3 warmup steps, each with its own roofline and a bounded cpu_baseline."""
    import copy

    res = {}
    t_all = time.perf_counter()
    for label, workload, direction, mib, maps in SECONDARY:
        a = copy.copy(args)
        a.workload, a.direction, a.map_mib, a.maps_per_gpu = workload, direction, mib, maps
        a.steps, a.warmup, a.task_threads, a.batch, a.verify = 8, 3, 0, -1, False  # (3 / 1 until round 4: the first steps still grow the context's workspace)
        a.cpu_seconds = min(args.cpu_seconds, 2.5)
        t0 = time.perf_counter()
        try:
            o = run_workload(a, rank, local_rank, 1, None)
        except Exception as e:  # a secondary line must never take the headline down
            res[label] = {"error": repr(e)}
            continue
        cb = o.get("cpu_baseline") or {}
        res[label] = {
            "value": o["value"], "unit": o["unit"], "ms_per_step": o["ms_per_step"], "steps": o["steps"], "warmup": o["warmup"],
            "metric": o["metric"],
            "config": {k: o["config"][k] for k in ("workload", "direction", "codec", "checksum", "partitions_per_map_task",
                                                    "map_task_bytes", "map_tasks_per_gpu", "compression_ratio",
                                                    "task_threads_per_gpu", "map_tasks_per_library_call")},
            "roofline": {k: o["roofline"][k] for k in ("kernel", "achieved", "peak", "unit", "frac", "avg_launch_ms",
                                                        "algorithmic_bytes_per_launch", "whole_path_read_frac")},
            "cpu_baseline": {k: cb.get(k) for k in ("value", "unit", "cores", "kind", "single_thread_GBps", "wall_s")} if cb else None,
            "speedup_vs_cpu_all_cores": o.get("speedup_vs_cpu_all_cores"),
            "wall_s": round(time.perf_counter() - t0, 2),
        }
        for k in ("bytes_verified", "image_verified"):
            if k in o:
                res[label][k] = o[k]
    t0 = time.perf_counter()
    try:
        res["block_size_sweep"] = run_block_size_sweep(args, rank, local_rank, res)
        res["block_size_sweep"]["wall_s"] = round(time.perf_counter() - t0, 2)
    except Exception as e:
        res["block_size_sweep"] = {"error": repr(e)}
    t0 = time.perf_counter()
    try:
        res["hbm_bound_stages"] = run_hbm_stages(args, local_rank)
        res["hbm_bound_stages"]["wall_s"] = round(time.perf_counter() - t0, 2)
    except Exception as e:
        res["hbm_bound_stages"] = {"error": repr(e)}
    _SKEW_CACHE.clear()
    t0 = time.perf_counter()
    try:
        res["host_path"] = run_host_path(args, local_rank)
        res["host_path"]["wall_s"] = round(time.perf_counter() - t0, 2)
    except Exception as e:
        res["host_path"] = {"error": repr(e)}
    res["wall_s_total"] = round(time.perf_counter() - t_all, 2)
    return res


def run_hbm_stages(args, local_rank: int):
    """The path's HBM-bound stages on their own: per-partition checksums over a 1 GiB range — four times
    the 256 MiB Infinity Cache, so the bytes come from HBM — and the xxHash32 pre-pass of the LZ4Block frames over a 1 GiB
    single-partition block.  Reference: S3ChecksumValidationStream.scala:54-86 / S3ShuffleHelper.scala:94-103 (read side),
    the checksums delivered at S3ShuffleMapOutputWriter.scala:91 (write side), LZ4BlockOutputStream's frame hash.
    `value` = wall clock around whole library calls (s3s_checksum_ranges_device incl. its launch + stream sync);
    `roofline.achieved` = range bytes / the HIP-event time of the kernels on the library's stream (algorithmic bytes = 1 B
    read per byte)."""
    import torch
    import s3shuffle

    dev = torch.device("cuda", local_rank)
    n_bytes = 1 << 30
    data, _ = make_map_output("skew-1part-lz4", 0, n_bytes)
    d = torch.from_numpy(np.ascontiguousarray(data)).to(dev)
    c = s3shuffle.Codec(local_rank)
    c.set_option(s3shuffle.codec.OPT_PROFILE, 1)
    res = {"what": "HBM-bound stages alone on a 1 GiB TeraSort block resident in HBM (4 x the 256 MiB Infinity Cache); 6 timed calls after 2 warm-up",
           "unit": "GB/s", "range_bytes": n_bytes}
    import zlib

    for name, algo, ref in (("adler32", s3shuffle.CHECKSUM_ADLER32, lambda b: zlib.adler32(b) & 0xFFFFFFFF),
                            ("crc32", s3shuffle.CHECKSUM_CRC32, lambda b: zlib.crc32(b) & 0xFFFFFFFF),
                            ("crc32c", s3shuffle.CHECKSUM_CRC32C, None)):
        for label, offs in (("1-range", np.array([0, n_bytes], np.int64)),
                            ("2000-ranges", np.linspace(0, n_bytes, 2001).astype(np.int64))):
            for _ in range(2):
                sums = c.checksum_ranges_device(algo, d.data_ptr(), offs)
            ev = []
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(6):
                sums = c.checksum_ranges_device(algo, d.data_ptr(), offs)
                ev.append(c.stage_ms(s3shuffle.codec.STAGE_CHECKSUM))
            dt = (time.perf_counter() - t0) / 6
            ms = sum(ev) / len(ev)
            e = {"value": round(n_bytes / dt / 1e9, 1), "ms_per_call": round(dt * 1e3, 4),
                 "roofline": {"bound": "hbm", "kernel": "checksum_segments_kernel + checksum_combine_kernel (%s)" % name,
                              "achieved": round(n_bytes / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                              "frac": round(n_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                              "frac_of_copy_ceiling": round(n_bytes / (ms * 1e-3) / 1e9 / HBM_COPY_CEILING_GBPS, 4),
                              "avg_kernels_ms": round(ms, 4), "algorithmic_bytes_per_launch": n_bytes, "traffic": None}}
            if ref is not None and label == "1-range":  # the value itself, against zlib on the host (untimed)
                e["matches_zlib"] = bool(int(sums[0]) == ref(data))
            res[f"checksum-only-1gib:{name}:{label}"] = e
    # xxHash32 pre-pass: the hash stage of one map-side call over the 1 GiB single-partition block (HIP events in the library)
    offs = np.array([0, n_bytes], np.int64)
    cap = c.max_compressed_size(s3shuffle.CODEC_LZ4, offs)
    dst = torch.empty(cap, dtype=torch.uint8, device=dev)
    hs = []
    for k in range(4):
        c.compress_map_output_device(s3shuffle.CODEC_LZ4, s3shuffle.CHECKSUM_ADLER32, d.data_ptr(), offs, dst.data_ptr(), cap)
        if k:
            hs.append(c.stage_ms(s3shuffle.codec.STAGE_HASH))
    ms = sum(hs) / len(hs)
    if ms > 0:
        res["xxh32-prepass-1gib"] = {"value": round(n_bytes / (ms * 1e-3) / 1e9, 1), "ms": round(ms, 4),
                                     "roofline": {"bound": "hbm", "kernel": "xxh32 of every 32 KiB block (frame check values)",
                                                  "achieved": round(n_bytes / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                                  "frac": round(n_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                                                  "algorithmic_bytes_per_launch": n_bytes, "traffic": None},
                                     "note": "HIP-event time of the hash stage inside s3s_compress_map_output_device (3 calls after 1 warm-up)"}
    c.close()
    del d, dst
    torch.cuda.synchronize()
    # roofline.traffic of these lines: the stamped PMC pass of the same command (tools/profile_sets.sh hbm, tools/profile_report.py)
    try:
        ref = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json")))
        stamp = _kernel_stamp()
        for key, e in res.items():
            if not isinstance(e, dict) or "roofline" not in e:
                continue
            t = ref.get("hbm-stages:adler32" if ":adler32:" in key else "hbm-stages:crc32" if ":crc32:" in key else "hbm-stages:xxh32" if key.startswith("xxh32") else "")
            if t and stamp and t.get("kernel_sources_sha256") == stamp:
                e["roofline"]["traffic"] = int(t["hbm_bytes_per_launch"])
                e["roofline"]["traffic_over_algorithmic"] = round(t["hbm_bytes_per_launch"] / n_bytes, 3)
                e["roofline"]["traffic_source"] = t.get("source")
    except Exception:
        pass
    return res


SWEEP = [(8, 8), (32, 8), (128, 8), (512, 2), (1024, 1)]  # (MiB per single-partition block, blocks per step)
SWEEP_IN_FLIGHT = [(8, 32)]  # the smallest block size again with 256 MiB per step


def run_block_size_sweep(args, rank: int, local_rank: int, have: dict):
    """north_star: "GB/s on synthetic 8 MiB-1 GiB shuffle blocks" (sweep 8 / 32 / 128 / 512 / 1024 MiB).  Single-
    partition TeraSort blocks resident in HBM, both directions, the batched entry points, 6 timed steps after 2 warmup, no
    CPU leg.  8 MiB is the reference's default write buffer (S3ShuffleDispatcher.scala:55): the common case is the small end.
    `have`: results of the secondary passes that already ran one of these points (the 1 GiB block)."""
    import copy

    res = {"what": "single-partition TeraSort blocks in HBM, LZ4 + Adler32; GB/s of uncompressed bytes; 6 timed steps, 2 warmup (the 1 GiB point: the secondary pass's 8 / 3)",
           "unit": "GB/s", "points": []}
    for mib, maps in SWEEP:
        point = {"block_MiB": mib, "blocks_per_step": maps}
        for direction in ("compress", "decompress"):
            prior = have.get(f"skew-1gib-lz4:{direction}") if (mib == 1024 and direction == "compress") else None
            if prior and "value" in prior:
                point[direction], point[direction + "_ms_per_step"] = prior["value"], prior["ms_per_step"]
                continue
            a = copy.copy(args)
            a.workload, a.direction, a.map_mib, a.maps_per_gpu = "skew-1part-lz4", direction, mib, maps
            a.steps, a.warmup, a.task_threads, a.batch, a.verify, a.no_cpu_baseline = 6, 2, 0, -1, False, True
            if direction == "decompress" and mib * maps <= 64:
                a.task_threads = 1  # (a small step is one batched call: 108.7 GB/s against 97.4 with two, profiles/r06h_*)
            if direction == "compress" and mib * maps <= 64:
                # small blocks: two task threads, each with ONE batched call over half of the step's blocks (what the shim's commit
                # queue does with the commits of concurrent tasks, S3GpuCommitQueue).  8 x 8 MiB are 2 048 block chains for a chip
                # that holds 2 560: the step lasts one block chain (0.9 - 1.0 ms) whatever the split; measured
                # (profiles/archive/r05k_small_blocks.txt) 49.9 / 52.4 / 42.6 / 37.9 GB/s with 1 / 2 / 4 / 8 threads
                a.task_threads = 2
            try:
                o = run_workload(a, rank, local_rank, 1, None)
                point[direction], point[direction + "_ms_per_step"] = o["value"], o["ms_per_step"]
                point[direction + "_task_threads"] = o["config"]["task_threads_per_gpu"]
            except Exception as e:
                point[direction] = None
                point[direction + "_error"] = repr(e)
        res["points"].append(point)
    # The small points above hold 64 / 256 MiB per step: they measure the LATENCY of one block chain (a 32 KiB block is 0.9 ms of
    # serial parse whoever runs it), not what the chip does with small blocks.  The same block sizes with as many blocks per
    # step as the executor's tasks commit together (32 x 8 MiB = 256 MiB in flight), two task threads:
    res["points_more_blocks_in_flight"] = []
    for mib, maps in SWEEP_IN_FLIGHT:
        point = {"block_MiB": mib, "blocks_per_step": maps}
        for direction in ("compress", "decompress"):
            a = copy.copy(args)
            a.workload, a.direction, a.map_mib, a.maps_per_gpu = "skew-1part-lz4", direction, mib, maps
            a.steps, a.warmup, a.task_threads, a.batch, a.verify, a.no_cpu_baseline = 6, 2, 2, -1, False, True
            try:
                o = run_workload(a, rank, local_rank, 1, None)
                point[direction], point[direction + "_ms_per_step"] = o["value"], o["ms_per_step"]
                point[direction + "_task_threads"] = o["config"]["task_threads_per_gpu"]
            except Exception as e:
                point[direction] = None
                point[direction + "_error"] = repr(e)
        res["points_more_blocks_in_flight"].append(point)
    return res


def run_host_path(args, local_rank: int):
    """The path a Spark executor gets: host buffers in, host buffers out, through the batched
    host entry points the JNI shim binds (s3s_compress_map_outputs_batch / s3s_decompress_ranges_batch) — PCIe
    inclusive, NEVER the headline `value`.  Page-locked buffers (s3s_host_alloc), the headline's TeraSort map outputs,
    1 / 2 / 4 task threads with one context each; every thread passes all its map tasks in one call."""
    import s3shuffle

    workload, maps, steps = "terasort-10g-200p-lz4", 8, 3
    codec_id, algo_id = s3shuffle.CODEC_LZ4, s3shuffle.CHECKSUM_ADLER32
    outs = [make_map_output(workload, m, args.map_mib << 20) for m in range(maps)]
    probe = s3shuffle.Codec(local_rank)
    src, dst, caps = [], [], []
    for data, offs in outs:
        b = s3shuffle.PinnedBuffer(data.size)
        b.array[:] = data
        src.append(b)
        caps.append(probe.max_compressed_size(codec_id, offs))
        dst.append(s3shuffle.PinnedBuffer(caps[-1]))
    probe.close()
    u_total = sum(int(d.size) for d, _ in outs)
    totals, images = [0] * maps, [None] * maps

    def timed(n_threads, fn):
        codecs = [s3shuffle.Codec(local_rank) for _ in range(n_threads)]

        def worker(tid, n):
            mine = list(range(tid, maps, n_threads))
            for _ in range(n):
                fn(codecs[tid], mine)

        def run(n):
            ths = [threading.Thread(target=worker, args=(j, n)) for j in range(n_threads)]
            [t.start() for t in ths]
            [t.join() for t in ths]

        run(1)
        import gc

        gc.collect()  # (see run_workload: a collector pause inside a 60 ms timed region is a 15 % error)
        gc.disable()
        t0 = time.perf_counter()
        run(steps)
        dt = time.perf_counter() - t0
        gc.enable()
        for c in codecs:
            c.close()
        return round(u_total * steps / dt / 1e9, 3)

    def do_compress(c, mine):
        res = c.compress_map_outputs_batch(codec_id, algo_id, [(src[i].ptr, outs[i][1], dst[i].ptr, caps[i]) for i in mine])
        for i, r in zip(mine, res):
            totals[i], images[i] = r[0], (r[1], r[2])

    comp = {str(t): timed(t, do_compress) for t in (1, 2, 4)}
    c_total = sum(totals)

    def do_decompress(c, mine):  # decoded bytes land in the source buffers (same content)
        res = c.decompress_ranges_batch(codec_id, algo_id, [(dst[i].ptr, totals[i], images[i][0], images[i][1], src[i].ptr,
                                                             int(outs[i][0].size)) for i in mine])
        assert all(r[0] == 0 and r[1] == outs[i][0].size for i, r in zip(mine, res))

    for b in src:
        b.array[:] = 0
    dec = {str(t): timed(t, do_decompress) for t in (1, 2)}
    ok = all(np.array_equal(src[i].array[:outs[i][0].size], outs[i][0]) for i in range(maps))
    for b in src + dst:
        b.free()
    return {
        "what": "host (page-locked) buffers in and out through s3s_compress_map_outputs_batch / s3s_decompress_ranges_batch: "
                "the figure a Spark executor gets through the JNI shim; PCIe-inclusive, not the headline value",
        "workload": workload, "map_tasks": maps, "map_task_bytes": int(outs[0][0].size), "steps": steps, "warmup": 1,
        "unit": "GB/s of uncompressed bytes", "compress_by_task_threads": comp, "verify_decompress_by_task_threads": dec,
        "pcie_bytes_per_step": {"compress": {"host_to_device": u_total, "device_to_host": c_total},
                                "decompress": {"host_to_device": c_total, "device_to_host": u_total}},
        "round_trip_bit_exact": bool(ok),
    }


SECONDARY_FILE = "bench_secondary.json"  # written next to bench.py (and under gpurun_out/ when that directory exists)
HEADLINE_MAX_BYTES = 4096                 # the LAST stdout line; tests/test_bench_line.py holds it under 8 192


def _short(s, n: int):
    s = str(s)
    return s if len(s) <= n else s[:n - 1] + "~"


def compact_headline(out: dict, secondary_file=None) -> dict:
    """The one line the driver parses (the round-5 line had grown to 22 KB and was not parsed): every
    contract key, `roofline` incl. `traffic`, `cpu_baseline`, short strings only.  Everything else — per-stage times, the long
    `sample` / `what` descriptions, the secondary workloads, the sweep, HBM-bound stages, host path — goes to SECONDARY_FILE
    and to EARLIER stdout lines."""
    cfg, rf, cb = out["config"], out["roofline"], out.get("cpu_baseline")
    line = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "rccl_ranks", "steps", "warmup", "ms_per_step",
                                "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    line["config"] = {
        "workload": cfg["workload"], "direction": cfg["direction"], "codec": _short(cfg["codec"], 48).split(" (")[0],
        "checksum": cfg["checksum"], "partitions_per_map_task": cfg["partitions_per_map_task"],
        "map_task_bytes": cfg["map_task_bytes"], "map_tasks_per_gpu": cfg["map_tasks_per_gpu"],
        "uncompressed_bytes_per_step": cfg["uncompressed_bytes_per_step"],
        "compressed_bytes_per_step": cfg["compressed_bytes_per_step"], "compression_ratio": cfg["compression_ratio"],
        "sharding": "mapId % nGPU", "task_threads_per_gpu": cfg["task_threads_per_gpu"],
        "map_tasks_per_library_call": cfg["map_tasks_per_library_call"], "inputs": "resident in HBM",
    }
    line["roofline"] = {k: rf.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic",
                                               "avg_launch_ms", "concurrent_launches", "algorithmic_bytes_per_launch",
                                               "achieved_all_streams", "frac_of_copy_ceiling", "whole_path_read_frac")}
    line["roofline"]["kernel"] = _short(rf["kernel"], 60)
    if rf.get("traffic_source"):
        line["roofline"]["traffic_source"] = _short(rf["traffic_source"], 64)
    if cb:
        line["cpu_baseline"] = {"value": cb["value"], "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                                "sample": _short(cb.get("sample_short") or cb.get("sample", ""), 120),
                                "single_thread_GBps": cb.get("single_thread_GBps"), "wall_s": cb.get("wall_s")}
    else:
        line["cpu_baseline"] = None
    for k in ("speedup_vs_cpu_all_cores", "speedup_vs_cpu_1_core", "image_verified", "bytes_verified", "kernel_sources_sha256"):
        if k in out:
            line[k] = out[k]
    line["stages_ms_per_library_call"] = out.get("stages_ms_per_library_call")
    line["per_rank"] = [{"rank": r["rank"], "GBps": r["GBps"], "elapsed_s": r["elapsed_s"]} for r in out.get("per_rank", [])][:16]
    if secondary_file:
        line["secondary_file"] = secondary_file
    return line


def secondary_summary(sec: dict) -> dict:
    """One short stdout line (the one before the headline) with the driver-visible figure of every secondary leg: value in
    GB/s of uncompressed bytes, the dominant kernel's roofline fraction, the speed-up over the host cores and the
    post-run byte checks.  Full records: SECONDARY_FILE."""
    legs = {}
    for label, e in sec.items():
        if not isinstance(e, dict):
            continue
        if "error" in e:
            legs[label] = {"error": _short(e["error"], 80)}
        elif "value" in e and "roofline" in e:
            legs[label] = {"GBps": e["value"], "ms_per_step": e["ms_per_step"], "frac": e["roofline"]["frac"],
                           "kernel_GBps": e["roofline"]["achieved"],
                           "cpu_GBps": (e.get("cpu_baseline") or {}).get("value"), "x_cpu": e.get("speedup_vs_cpu_all_cores")}
            for k in ("image_verified", "bytes_verified"):
                if k in e:
                    legs[label][k] = e[k]
    res = {"secondary_summary": legs}
    sw = sec.get("block_size_sweep") or {}
    if "points" in sw:
        res["sweep_MiB_blocks_compress_decompress_GBps"] = [
            [pt["block_MiB"], pt["blocks_per_step"], pt.get("compress"), pt.get("decompress")]
            for pt in sw["points"] + sw.get("points_more_blocks_in_flight", [])]
    hb = sec.get("hbm_bound_stages") or {}
    res["hbm_stage_kernel_GBps_frac"] = {k: [e["roofline"]["achieved"], e["roofline"]["frac"]] + ([e["matches_zlib"]] if "matches_zlib" in e else [])
                                         for k, e in hb.items() if isinstance(e, dict) and "roofline" in e}
    hp = sec.get("host_path") or {}
    if "compress_by_task_threads" in hp:
        res["host_path_GBps"] = {"compress": hp["compress_by_task_threads"], "decompress": hp["verify_decompress_by_task_threads"],
                                 "bit_exact": hp.get("round_trip_bit_exact")}
    return res


def write_secondary_file(full: dict):
    """The full record (headline with its long descriptions + every secondary leg) as a file; returns the path reported in the
    headline."""
    blob = json.dumps(full, indent=1)
    paths = [os.path.join(ROOT, SECONDARY_FILE)]
    if os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        paths.append(os.path.join(ROOT, "gpurun_out", SECONDARY_FILE))
    for p in paths:
        try:
            with open(p, "w") as f:
                f.write(blob)
        except OSError:
            pass
    return SECONDARY_FILE


def self_launch(args) -> int:
    """`python bench.py --gpus N` without a launcher: re-exec this command under torch.distributed.run, one
    rank per GPU on 127.0.0.1 (S3ShuffleDispatcher.scala:142-143: map task m -> GPU m % N); the ranks print, this process only
    relays their exit code."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def main():
    args = parse_args()
    if args.secondary is None:  # plain headline command only (the driver's); profiling / A-B commands stay short
        args.secondary = (args.gpus == 1 and args.workload == "terasort-10g-200p-lz4" and args.direction == "compress"
                          and not args.no_cpu_baseline and not args.dry_run and args.map_mib == 128)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")

    import torch
    import s3shuffle
    from s3shuffle import sharding

    # one rank per GPU; under torch.distributed.run the process group is ALWAYS created (also with one rank), so the
    # single-GPU launch line exercises the same RCCL barrier / all_reduce code the N-GPU runs use
    launched = world > 1 or "TORCHELASTIC_RUN_ID" in os.environ or "GROUP_RANK" in os.environ
    if args.dry_run:
        return dry_run(args, rank, local_rank, world, launched)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the codec library has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if launched:
        import torch.distributed as dist_mod

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist_mod.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        dist = dist_mod

    if args.host_path_only:
        print(json.dumps({"host_path": run_host_path(args, local_rank)}), flush=True)
        return
    if args.hbm_stages_only:
        print(json.dumps({"hbm_bound_stages": run_hbm_stages(args, local_rank)}), flush=True)
        return
    out = run_workload(args, rank, local_rank, world, dist)
    if rank == 0:
        # stdout: [full record incl. secondaries, one long line] [secondary summary, short] [THE headline, short, LAST]
        sec_file = None
        if world == 1 and args.secondary and not args.dry_run:
            sec = run_secondaries(args, rank, local_rank)
            full = dict(out, secondary=sec)
            sec_file = write_secondary_file(full)
            print(json.dumps({"full_record": full}), flush=True)
            print(json.dumps(secondary_summary(sec)), flush=True)
        elif args.full_line:
            print(json.dumps({"full_record": out}), flush=True)
        line = json.dumps(compact_headline(out, sec_file))
        assert len(line) < HEADLINE_MAX_BYTES, len(line)
        print(line, flush=True)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
