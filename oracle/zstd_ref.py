"""libzstd 1.4.8 (the library in this image: /usr/lib/x86_64-linux-gnu/libzstd.so.1) through ctypes.

TEST INFRASTRUCTURE ONLY, like the rest of oracle/.  For Zstandard there is nothing to restate: the reference's
writer side is zstd-jni (a JNI build of libzstd) driven by Spark's ZStdCompressionCodec — level 1, 32 KiB buffer,
streaming API, no content size, no checksum — and the product only DECODES such streams (S3S_CODEC_ZSTD on the reduce
side).  `compress_stream` produces what that writer produces (same library calls); other levels / one-shot frames /
checksummed frames widen the decoder's test coverage."""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import numpy as np

_Z = None
CODEC_ZSTD = 3
ZSTD_c_compressionLevel, ZSTD_c_windowLog, ZSTD_c_checksumFlag, ZSTD_c_contentSizeFlag = 100, 101, 201, 200
ZSTD_e_continue, ZSTD_e_flush, ZSTD_e_end = 0, 1, 2


class _Buf(ctypes.Structure):
    _fields_ = [("p", ctypes.c_void_p), ("size", ctypes.c_size_t), ("pos", ctypes.c_size_t)]


def lib():
    global _Z
    if _Z is None:
        z = ctypes.CDLL("libzstd.so.1")
        z.ZSTD_versionNumber.restype = ctypes.c_uint
        z.ZSTD_compressBound.restype = ctypes.c_size_t
        z.ZSTD_compressBound.argtypes = [ctypes.c_size_t]
        z.ZSTD_compress.restype = ctypes.c_size_t
        z.ZSTD_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        z.ZSTD_decompress.restype = ctypes.c_size_t
        z.ZSTD_decompress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
        z.ZSTD_isError.restype = ctypes.c_uint
        z.ZSTD_isError.argtypes = [ctypes.c_size_t]
        z.ZSTD_createCCtx.restype = ctypes.c_void_p
        z.ZSTD_freeCCtx.argtypes = [ctypes.c_void_p]
        z.ZSTD_CCtx_setParameter.restype = ctypes.c_size_t
        z.ZSTD_CCtx_setParameter.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        z.ZSTD_compressStream2.restype = ctypes.c_size_t
        z.ZSTD_compressStream2.argtypes = [ctypes.c_void_p, ctypes.POINTER(_Buf), ctypes.POINTER(_Buf), ctypes.c_int]
        z.ZSTD_createDCtx.restype = ctypes.c_void_p
        z.ZSTD_freeDCtx.argtypes = [ctypes.c_void_p]
        z.ZSTD_decompressStream.restype = ctypes.c_size_t
        z.ZSTD_decompressStream.argtypes = [ctypes.c_void_p, ctypes.POINTER(_Buf), ctypes.POINTER(_Buf)]
        _Z = z
    return _Z


def version() -> int:
    return int(lib().ZSTD_versionNumber())


def compress_stream(data, level: int = 1, chunk: int = 32768, checksum: bool = False, window_log: int = 0) -> np.ndarray:
    """One frame the way zstd-jni's ZstdOutputStream writes it: the bytes arrive `chunk` at a time (Spark's 32 KiB
    BufferedOutputStream), ZSTD_e_continue for each, ZSTD_e_end on close.  No content size in the header."""
    z = lib()
    d = np.ascontiguousarray(data, dtype=np.uint8)
    cctx = z.ZSTD_createCCtx()
    try:
        z.ZSTD_CCtx_setParameter(cctx, ZSTD_c_compressionLevel, level)
        z.ZSTD_CCtx_setParameter(cctx, ZSTD_c_checksumFlag, 1 if checksum else 0)
        if window_log:
            z.ZSTD_CCtx_setParameter(cctx, ZSTD_c_windowLog, window_log)
        cap = int(z.ZSTD_compressBound(d.size)) + 1024
        out = np.empty(cap, dtype=np.uint8)
        ob = _Buf(out.ctypes.data, cap, 0)
        pos = 0
        while pos < d.size:
            n = min(chunk, d.size - pos)
            ib = _Buf(d.ctypes.data + pos, n, 0)
            while ib.pos < ib.size:
                r = z.ZSTD_compressStream2(cctx, ctypes.byref(ob), ctypes.byref(ib), ZSTD_e_continue)
                if z.ZSTD_isError(r):
                    raise RuntimeError("ZSTD_compressStream2 failed")
            pos += n
        ib = _Buf(d.ctypes.data, 0, 0)
        while True:
            r = z.ZSTD_compressStream2(cctx, ctypes.byref(ob), ctypes.byref(ib), ZSTD_e_end)
            if z.ZSTD_isError(r):
                raise RuntimeError("ZSTD_compressStream2(end) failed")
            if r == 0:
                break
        return out[: ob.pos].copy()
    finally:
        z.ZSTD_freeCCtx(cctx)


def compress(data, level: int = 3) -> np.ndarray:
    """One-shot frame (single segment, content size in the header)."""
    z = lib()
    d = np.ascontiguousarray(data, dtype=np.uint8)
    cap = int(z.ZSTD_compressBound(d.size))
    out = np.empty(cap, dtype=np.uint8)
    r = z.ZSTD_compress(out.ctypes.data, cap, d.ctypes.data, d.size, level)
    if z.ZSTD_isError(r):
        raise RuntimeError("ZSTD_compress failed")
    return out[:r].copy()


def decompress(comp, capacity: int) -> Optional[np.ndarray]:
    """All frames of `comp` (streaming decoder: concatenated and skippable frames as zstd-jni's ZstdInputStream with
    setContinuous sees them).  None when libzstd reports an error."""
    z = lib()
    c = np.ascontiguousarray(comp, dtype=np.uint8)
    out = np.empty(max(capacity, 1), dtype=np.uint8)
    dctx = z.ZSTD_createDCtx()
    try:
        ib = _Buf(c.ctypes.data, c.size, 0)
        ob = _Buf(out.ctypes.data, capacity, 0)
        r = 0
        while ib.pos < ib.size:
            r = z.ZSTD_decompressStream(dctx, ctypes.byref(ob), ctypes.byref(ib))
            if z.ZSTD_isError(r):
                return None
            if ob.pos == ob.size and ib.pos < ib.size and r != 0:
                return None  # capacity
        if r != 0:
            return None  # truncated frame
        return out[: ob.pos].copy()
    finally:
        z.ZSTD_freeDCtx(dctx)


def compress_map_output(checksum_algo: int, data, offsets, level: int = 1) -> Tuple[np.ndarray, np.ndarray, Optional[np.ndarray]]:
    """The `.data` image / index / checksums of a map task written with spark.io.compression.codec=zstd: one frame per
    non-empty partition."""
    from oracle import binding

    d = np.ascontiguousarray(data, dtype=np.uint8)
    offs = np.asarray(offsets, dtype=np.int64)
    n = len(offs) - 1
    parts = [compress_stream(d[offs[p]:offs[p + 1]], level) if offs[p + 1] > offs[p] else np.zeros(0, np.uint8) for p in range(n)]
    index = np.zeros(n + 1, dtype=np.int64)
    np.cumsum([p.size for p in parts], out=index[1:])
    img = np.concatenate(parts) if parts else np.zeros(0, np.uint8)
    sums = np.array([binding.checksum(checksum_algo, p) for p in parts], dtype=np.int64) if checksum_algo else None
    return img, index, sums
