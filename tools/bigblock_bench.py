"""(GPU) verify + decompress of objects written with spark.io.compression.lz4.blockSize above the default: 32k / 64k / 256k /
1m LZ4Block frames built on the host with liblz4 (what a JVM writer with that key produces: LZ4_compress_default per block,
byU32 parse from 64 KiB on), 200 partitions of a 128 MiB TeraSort map output, decoded by the default reduce-side call.
Round 4: frames above 32 KiB go through the batch decoder (before: ring decoder, one sequence per step).
usage: python tools/bigblock_bench.py [--mib 128] [--steps 5]"""
import argparse
import ctypes
import os
import struct
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "spark-s3-shuffle_amd"))
import numpy as np  # noqa: E402
import xxhash  # noqa: E402


def jvm_stream(L, data, bs):
    level = max(0, int(bs).bit_length() - 1 - 10)
    out = bytearray()
    buf = np.empty(bs + bs // 200 + 64, np.uint8)
    for p in range(0, data.size, bs):
        chunk = np.ascontiguousarray(data[p:p + bs])
        n = L.LZ4_compress_default(chunk.ctypes.data, buf.ctypes.data, chunk.size, buf.size)
        raw = n <= 0 or n >= chunk.size
        body = chunk.tobytes() if raw else buf[:n].tobytes()
        out += b"LZ4Block" + bytes([(0x10 if raw else 0x20) | level]) + struct.pack(
            "<iiI", len(body), chunk.size, xxhash.xxh32(chunk.tobytes(), seed=0x9747B28C).intdigest() & 0x0FFFFFFF) + body
    if data.size:
        out += b"LZ4Block" + bytes([0x10 | level]) + struct.pack("<iii", 0, 0, 0)
    return bytes(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mib", type=int, default=128)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--partitions", type=int, default=200)
    args = ap.parse_args()
    import torch

    import s3shuffle
    from s3shuffle import datagen

    L = ctypes.CDLL("liblz4.so.1")
    L.LZ4_compress_default.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    data, offs = datagen.terasort_map_output(args.mib << 20, args.partitions, seed=2, map_id=0)
    dev = torch.device("cuda", 0)
    codec = s3shuffle.Codec(0)
    out = torch.empty(data.size, dtype=torch.uint8, device=dev)
    for bs in (32768, 65536, 262144, 1 << 20):
        streams = [jvm_stream(L, data[offs[p]:offs[p + 1]], bs) for p in range(args.partitions)]
        img = np.frombuffer(b"".join(streams), np.uint8)
        index = np.concatenate([[0], np.cumsum([len(s) for s in streams])]).astype(np.int64)
        sums = np.array([zlib.adler32(s) for s in streams], np.int64)
        d_img = torch.from_numpy(img.copy()).to(dev)
        args_ = (s3shuffle.CODEC_LZ4, s3shuffle.CHECKSUM_ADLER32, d_img.data_ptr(), img.size, index, sums, out.data_ptr(), data.size)
        assert codec.decompress_range_device(*args_) == data.size
        assert np.array_equal(out.cpu().numpy(), data), bs
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            codec.decompress_range_device(*args_)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        print(f"lz4.blockSize {bs >> 10:5d}k: {len(img) / 1e6:8.1f} MB compressed (ratio {data.size / len(img):.2f}), "
              f"verify + decompress {data.size / dt / 1e9:7.1f} GB/s ({dt * 1e3:.2f} ms per {args.mib} MiB map output)", flush=True)
    codec.close()


if __name__ == "__main__":
    main()
