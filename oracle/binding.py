"""ctypes binding of oracle/liboracle.so.

TEST INFRASTRUCTURE ONLY (see s3s_oracle.h): imported by tests/, __graft_entry__.smoke() and
the cpu_baseline leg of bench.py — never by the product package.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_LIB = None

CODEC_NONE, CODEC_LZ4, CODEC_SNAPPY = 0, 1, 2
CODEC_LZF = 4  # the oracle writes LZF streams with its own greedy encoder (test data); the GPU path only decodes them
CHECKSUM_NONE, CHECKSUM_ADLER32, CHECKSUM_CRC32, CHECKSUM_CRC32C = 0, 1, 2, 3
E_INVALID, E_CAPACITY, E_BAD_FRAME, E_CHECKSUM, E_UNSUPPORTED = -1, -2, -3, -4, -6


def build(force: bool = False) -> str:
    path = os.path.join(_DIR, "liboracle.so")
    srcs = [os.path.join(_DIR, f) for f in os.listdir(_DIR) if f.endswith((".c", ".h"))]
    stale = not os.path.exists(path) or any(os.path.getmtime(s) > os.path.getmtime(path) for s in srcs)
    if force or stale:
        subprocess.run(["make", "-C", _DIR, "-B" if force else "-s"], check=True)
    return path


def lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is not None:
        return _LIB
    L = ctypes.CDLL(build())
    vp, i32, i64, u32 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_uint32
    L.s3o_xxh32.restype = u32
    L.s3o_xxh32.argtypes = [vp, ctypes.c_size_t, u32]
    L.s3o_crc32.restype = u32
    L.s3o_crc32.argtypes = [u32, vp, ctypes.c_size_t]
    L.s3o_adler32.restype = u32
    L.s3o_adler32.argtypes = [u32, vp, ctypes.c_size_t]
    L.s3o_checksum.restype = i64
    L.s3o_checksum.argtypes = [i32, vp, ctypes.c_size_t]
    L.s3o_lz4_compress_bound.argtypes = [i32]
    L.s3o_lz4_compress_block.argtypes = [vp, i32, vp, i32]
    L.s3o_lz4_decompress_block.argtypes = [vp, i32, vp, i32, vp]
    for name in ("lz4block", "snappy"):
        f = getattr(L, f"s3o_{name}_max_stream_size")
        f.restype = i64
        f.argtypes = [i64, i32]
        f = getattr(L, f"s3o_{name}_compress_stream")
        f.restype = i64
        f.argtypes = [vp, i64, i32, vp, i64]
        f = getattr(L, f"s3o_{name}_decompress_stream")
        f.restype = i64
        f.argtypes = [vp, i64, vp, i64]
    L.s3o_lzf_decompress_block.argtypes = [vp, i32, vp, i32]
    L.s3o_lzf_compress_block.argtypes = [vp, i32, vp, i32]
    L.s3o_lzf_max_stream_size.restype = i64
    L.s3o_lzf_max_stream_size.argtypes = [i64]
    L.s3o_lzf_compress_stream.restype = i64
    L.s3o_lzf_compress_stream.argtypes = [vp, i64, vp, i64]
    L.s3o_lzf_decompress_stream.restype = i64
    L.s3o_lzf_decompress_stream.argtypes = [vp, i64, vp, i64]
    L.s3o_snappy_max_compressed_length.argtypes = [i32]
    L.s3o_snappy_compress_block.argtypes = [vp, i32, vp, i32]
    L.s3o_snappy_decompress_block.argtypes = [vp, i32, vp, i32]
    L.s3o_max_compressed_size.restype = i64
    L.s3o_max_compressed_size.argtypes = [i32, i32, vp, ctypes.c_int32]
    L.s3o_compress_map_output.argtypes = [i32, i32, i32, vp, vp, ctypes.c_int32, vp, i64, vp, vp, vp]
    L.s3o_decompress_range.argtypes = [i32, i32, vp, i64, vp, vp, ctypes.c_int32, vp, i64, vp, vp]
    L.s3o_longs_to_be.argtypes = [vp, i64, vp]
    L.s3o_longs_from_be.argtypes = [vp, i64, vp]
    L.s3o_mt_have_liblz4.restype = i32
    L.s3o_mt_compress_bench.restype = ctypes.c_double
    L.s3o_mt_compress_bench.argtypes = [i32, i32, i32, i32, vp, vp, ctypes.c_int32, i32, i32, vp]
    L.s3o_mt_decompress_bench.restype = ctypes.c_double
    L.s3o_mt_decompress_bench.argtypes = [i32, i32, i32, vp, i64, vp, vp, ctypes.c_int32, i64, i32, i32, vp]
    L.s3o_mt_stream_liblz4.restype = i64
    L.s3o_mt_stream_liblz4.argtypes = [vp, i64, i32, vp]
    L.s3o_mt_have_libsnappy.restype = i32
    L.s3o_mt_stream_libsnappy.restype = i64
    L.s3o_mt_stream_libsnappy.argtypes = [vp, i64, i32, vp]
    L.s3o_mt_decode_libsnappy.restype = i64
    L.s3o_mt_decode_libsnappy.argtypes = [vp, i64, vp, i64]
    L.s3o_crc32_fast.restype = u32
    L.s3o_crc32_fast.argtypes = [u32, vp, ctypes.c_size_t]
    for f in (L.s3o_crc32c, L.s3o_crc32c_hw):
        f.restype = u32
        f.argtypes = [u32, vp, ctypes.c_size_t]
    L.s3o_adler32_fast.restype = u32
    L.s3o_adler32_fast.argtypes = [u32, vp, ctypes.c_size_t]
    L.s3o_checksum_fast.restype = i64
    L.s3o_checksum_fast.argtypes = [i32, vp, ctypes.c_size_t]
    L.s3o_simd_available.restype = i32
    L.s3o_simd_crc_constants.argtypes = [vp]
    _LIB = L
    return L


def _u8(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint8)


def _i64(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.int64))


def xxh32(data, seed: int = 0x9747B28C) -> int:
    d = _u8(data)
    return int(lib().s3o_xxh32(d.ctypes.data, d.size, seed))


def checksum(algo: int, data) -> int:
    d = _u8(data)
    return int(lib().s3o_checksum(algo, d.ctypes.data, d.size))


def checksum_fast(algo: int, data, init: Optional[int] = None) -> int:
    """the cpu_baseline's checksums (s3s_oracle_simd.c: PCLMULQDQ CRC32 / SSSE3 Adler32); init = running value"""
    d = _u8(data)
    if init is None:
        return int(lib().s3o_checksum_fast(algo, d.ctypes.data, d.size))
    f = {CHECKSUM_ADLER32: lib().s3o_adler32_fast, CHECKSUM_CRC32: lib().s3o_crc32_fast, CHECKSUM_CRC32C: lib().s3o_crc32c_hw}[algo]
    return int(f(init, d.ctypes.data, d.size))


def crc32c(data, init: int = 0, hw: bool = False) -> int:
    """java.util.zip.CRC32C restated (table form) or, hw=True, through the x86 crc32 instruction"""
    d = _u8(data)
    return int((lib().s3o_crc32c_hw if hw else lib().s3o_crc32c)(init, d.ctypes.data, d.size))


def crc32c_hw_available() -> bool:
    return bool(lib().s3o_crc32c_hw_available())


def simd_crc_constants():
    k = (ctypes.c_uint64 * 7)()
    lib().s3o_simd_crc_constants(k)
    return [int(x) for x in k]


def lz4_compress_block(data) -> np.ndarray:
    d = _u8(data)
    cap = lib().s3o_lz4_compress_bound(d.size)
    out = np.empty(cap + 8, dtype=np.uint8)
    r = lib().s3o_lz4_compress_block(d.ctypes.data, d.size, out.ctypes.data, cap)
    return out[:r].copy()


def lzf_compress_block(data) -> np.ndarray:
    """the oracle's own greedy LZF encoder (test data only)"""
    d = _u8(data)
    out = np.empty(d.size + d.size // 16 + 64, dtype=np.uint8)
    r = lib().s3o_lzf_compress_block(d.ctypes.data, d.size, out.ctypes.data, out.size)
    if r < 0:
        raise RuntimeError(f"oracle lzf_compress_block rc={r}")
    return out[:r].copy()


def lzf_decompress_block(block, ulen: int):
    """-> decoded bytes, or the negative error code"""
    b = _u8(block)
    out = np.empty(max(ulen, 1), dtype=np.uint8)
    r = lib().s3o_lzf_decompress_block(b.ctypes.data, b.size, out.ctypes.data, ulen)
    return out[:r].copy() if r >= 0 else int(r)


def snappy_compress_block(data) -> np.ndarray:
    d = _u8(data)
    cap = lib().s3o_snappy_max_compressed_length(d.size)
    out = np.empty(cap + 8, dtype=np.uint8)
    r = lib().s3o_snappy_compress_block(d.ctypes.data, d.size, out.ctypes.data, cap)
    return out[:r].copy()


def compress_stream(codec: int, data, block_size: int = 32768) -> np.ndarray:
    d = _u8(data)
    name = "lz4block" if codec == CODEC_LZ4 else "snappy"
    cap = int(getattr(lib(), f"s3o_{name}_max_stream_size")(d.size, block_size))
    out = np.empty(cap + 8, dtype=np.uint8)
    r = getattr(lib(), f"s3o_{name}_compress_stream")(d.ctypes.data, d.size, block_size, out.ctypes.data, cap)
    if r < 0:
        raise RuntimeError(f"oracle compress_stream rc={r}")
    return out[:r].copy()


def decompress_stream(codec: int, comp, capacity: int) -> np.ndarray:
    c = _u8(comp)
    name = "lz4block" if codec == CODEC_LZ4 else "snappy"
    out = np.empty(max(capacity, 1), dtype=np.uint8)
    r = getattr(lib(), f"s3o_{name}_decompress_stream")(c.ctypes.data, c.size, out.ctypes.data, capacity)
    if r < 0:
        raise RuntimeError(f"oracle decompress_stream rc={r}")
    return out[:r].copy()


def compress_map_output(codec: int, checksum_algo: int, data, offsets, block_size: int = 32768
                        ) -> Tuple[np.ndarray, np.ndarray, Optional[np.ndarray]]:
    """-> (.data image, index[N+1], checksums[N] or None); the checker for
    s3s_compress_map_output."""
    d = _u8(data)
    offs = _i64(offsets)
    n = len(offs) - 1
    cap = int(lib().s3o_max_compressed_size(codec, block_size, offs.ctypes.data, n))
    if cap < 0:
        raise RuntimeError(f"oracle max_compressed_size rc={cap}")
    dst = np.empty(max(cap, 1), dtype=np.uint8)
    index = np.zeros(n + 1, dtype=np.int64)
    sums = np.zeros(max(n, 1), dtype=np.int64)
    total = ctypes.c_int64(0)
    rc = lib().s3o_compress_map_output(codec, checksum_algo, block_size, d.ctypes.data, offs.ctypes.data, n,
                                       dst.ctypes.data, cap, index.ctypes.data,
                                       sums.ctypes.data if checksum_algo else None, ctypes.byref(total))
    if rc != 0:
        raise RuntimeError(f"oracle compress_map_output rc={rc}")
    return dst[: total.value].copy(), index, (sums[:n] if checksum_algo else None)


def decompress_range(codec: int, checksum_algo: int, comp, part_offsets, ref_checksums, capacity: int
                     ) -> Tuple[int, np.ndarray, int]:
    """-> (rc, decoded bytes, bad_partition)"""
    c = _u8(comp)
    offs = _i64(part_offsets)
    n = len(offs) - 1
    refs = _i64(ref_checksums) if ref_checksums is not None else None
    out = np.empty(max(capacity, 1), dtype=np.uint8)
    out_len = ctypes.c_int64(0)
    bad = ctypes.c_int32(-1)
    rc = lib().s3o_decompress_range(codec, checksum_algo, c.ctypes.data, c.size, offs.ctypes.data,
                                    refs.ctypes.data if refs is not None else None, n, out.ctypes.data,
                                    capacity, ctypes.byref(out_len), ctypes.byref(bad))
    return int(rc), out[: out_len.value].copy(), int(bad.value)


def longs_to_be(values) -> bytes:
    v = _i64(values)
    out = np.empty(8 * v.size, dtype=np.uint8)
    lib().s3o_longs_to_be(v.ctypes.data, v.size, out.ctypes.data)
    return out.tobytes()


def mt_compress_bench(codec: int, checksum_algo: int, data, offsets, nthreads: int, reps: int = 1,
                      block_size: int = 32768, use_liblz4: bool = True) -> Tuple[float, int]:
    """cpu_baseline: `nthreads` map tasks in parallel, one per thread. -> (seconds, bytes out)"""
    d = _u8(data)
    offs = _i64(offsets)
    total = ctypes.c_int64(0)
    s = lib().s3o_mt_compress_bench(codec, checksum_algo, block_size, int(use_liblz4), d.ctypes.data,
                                    offs.ctypes.data, len(offs) - 1, nthreads, reps, ctypes.byref(total))
    return float(s), int(total.value)


def mt_decompress_bench(codec: int, checksum_algo: int, comp, part_offsets, ref_checksums, decoded_size: int,
                        nthreads: int, reps: int = 1, use_liblz4: bool = True) -> Tuple[float, int]:
    """cpu_baseline (reduce side): `nthreads` tasks verify + decode the same fetched range. -> (seconds, bytes)"""
    c = _u8(comp)
    offs = _i64(part_offsets)
    sums = _i64(ref_checksums) if ref_checksums is not None else None
    out_len = ctypes.c_int64(0)
    s = lib().s3o_mt_decompress_bench(codec, checksum_algo, int(use_liblz4), c.ctypes.data, c.size, offs.ctypes.data,
                                      sums.ctypes.data if sums is not None else None, len(offs) - 1, int(decoded_size),
                                      nthreads, reps, ctypes.byref(out_len))
    return float(s), int(out_len.value)
