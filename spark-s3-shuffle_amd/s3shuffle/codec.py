"""ctypes binding of include/s3shuffle_codec.h (one method per C entry point)."""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Sequence, Tuple

import numpy as np

CODEC_NONE, CODEC_LZ4, CODEC_SNAPPY = 0, 1, 2
CODEC_ZSTD = 3  # reduce side only: decode of Zstandard frames (the compress entry points refuse it)
CODEC_LZF = 4  # reduce side only: LZFCompressionCodec streams (compress-lzf chunks around liblzf blocks)
CHECKSUM_NONE, CHECKSUM_ADLER32, CHECKSUM_CRC32, CHECKSUM_CRC32C = 0, 1, 2, 3

OPT_LZ4_BLOCK_SIZE, OPT_SNAPPY_BLOCK_SIZE, OPT_PROFILE = 1, 2, 3
STAGE_TOTAL, STAGE_CODEC, STAGE_ASSEMBLE, STAGE_CHECKSUM, STAGE_DISCOVER, STAGE_HASH = 0, 1, 2, 3, 4, 5
OPT_LZ4_VARIANT = 4
OPT_LZ4_DECODE_VARIANT = 5
OPT_SNAPPY_VARIANT = 6
OPT_LZ4_VARIANT_USED = 7

E_INVALID, E_CAPACITY, E_BAD_FRAME, E_CHECKSUM, E_HIP, E_UNSUPPORTED, E_NOMEM = -1, -2, -3, -4, -5, -6, -7
STATUS_NOT_RUN = -100  # per-entry status of a batch call that failed as a call before this entry had a verdict (ABI 6)
_ERR_NAMES = {
    E_INVALID: "S3S_E_INVALID",
    E_CAPACITY: "S3S_E_CAPACITY",
    E_BAD_FRAME: "S3S_E_BAD_FRAME",
    E_CHECKSUM: "S3S_E_CHECKSUM",
    E_HIP: "S3S_E_HIP",
    E_UNSUPPORTED: "S3S_E_UNSUPPORTED",
    E_NOMEM: "S3S_E_NOMEM",
}

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class CodecError(RuntimeError):
    """A negative S3S_E_* return code.  `.code` is the code, `.partition` the failing
    partition for E_CHECKSUM (mirrors SparkException("Invalid checksum detected for ..."))."""

    def __init__(self, code: int, message: str, partition: int = -1):
        super().__init__(f"{_ERR_NAMES.get(code, code)}: {message}")
        self.code = code
        self.partition = partition


def library_path() -> str:
    """The product library; S3S_CODEC_LIB selects another build of the SAME library (e.g. the
    instrumented one used by tools/lz4_timing.py) — never a fallback implementation."""
    override = os.environ.get("S3S_CODEC_LIB")
    if override:
        return override
    return os.path.join(_PKG_ROOT, "lib", "libs3shuffle_codec.so")


class MapTask(ctypes.Structure):
    """struct s3s_map_task (include/s3shuffle_codec.h): one map task of a batched compress call."""
    _fields_ = [("d_src", ctypes.c_void_p), ("src_offsets", ctypes.POINTER(ctypes.c_int64)),
                ("num_partitions", ctypes.c_int32), ("d_dst", ctypes.c_void_p), ("dst_capacity", ctypes.c_int64),
                ("out_index", ctypes.POINTER(ctypes.c_int64)), ("out_checksums", ctypes.POINTER(ctypes.c_int64)),
                ("out_total", ctypes.c_int64), ("status", ctypes.c_int32)]


_LIB = None


class FetchRange(ctypes.Structure):
    """s3s_fetch_range (include/s3shuffle_codec.h)."""
    _fields_ = [("d_comp", ctypes.c_void_p), ("comp_len", ctypes.c_int64),
                ("part_offsets", ctypes.POINTER(ctypes.c_int64)), ("ref_checksums", ctypes.POINTER(ctypes.c_int64)),
                ("num_partitions", ctypes.c_int32), ("d_dst", ctypes.c_void_p), ("dst_capacity", ctypes.c_int64),
                ("out_len", ctypes.c_int64), ("bad_partition", ctypes.c_int32), ("status", ctypes.c_int32)]


def load_library() -> ctypes.CDLL:
    """Loads the HIP codec library.  Fails loudly if it has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # The HIP runtime maps all streams of a process onto GPU_MAX_HW_QUEUES hardware queues (default 4) and reads
    # the variable when libamdhip64 is loaded; kernels that share a queue run one after the other.  One context
    # per task thread with 8 threads on 8 MiB map outputs: 12 GB/s with 4 queues, 23 GB/s with 16.  A JVM gets
    # the same through spark.executorEnv.GPU_MAX_HW_QUEUES; an explicit setting wins.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    path = library_path()
    if not os.path.exists(path):
        raise FileNotFoundError(
            f"{path} is missing: build it with `python __graft_entry__.py` (or "
            f"`make -C spark-s3-shuffle_amd/csrc`). There is no CPU fallback."
        )
    lib = ctypes.CDLL(path)
    c_i64p = ctypes.POINTER(ctypes.c_int64)
    vp = ctypes.c_void_p
    lib.s3s_version.restype = ctypes.c_char_p
    lib.s3s_abi_version.restype = ctypes.c_int
    lib.s3s_device_count.restype = ctypes.c_int
    lib.s3s_create.restype = vp
    lib.s3s_create.argtypes = [ctypes.c_int, ctypes.c_int64]
    lib.s3s_destroy.argtypes = [vp]
    lib.s3s_last_error.restype = ctypes.c_char_p
    lib.s3s_last_error.argtypes = [vp]
    lib.s3s_set_option.argtypes = [vp, ctypes.c_int, ctypes.c_int64]
    lib.s3s_get_option.restype = ctypes.c_int64
    lib.s3s_get_option.argtypes = [vp, ctypes.c_int]
    lib.s3s_stream.restype = vp
    lib.s3s_stream.argtypes = [vp]
    lib.s3s_stage_ms.restype = ctypes.c_double
    lib.s3s_stage_ms.argtypes = [vp, ctypes.c_int]
    lib.s3s_max_compressed_size.restype = ctypes.c_int64
    lib.s3s_max_compressed_size.argtypes = [vp, ctypes.c_int, c_i64p, ctypes.c_int32]
    comp_args = [vp, ctypes.c_int, ctypes.c_int, vp, c_i64p, ctypes.c_int32, vp, ctypes.c_int64,
                 c_i64p, c_i64p, c_i64p]
    lib.s3s_compress_map_output.argtypes = comp_args
    lib.s3s_compress_map_output_device.argtypes = comp_args
    ck_args = [vp, ctypes.c_int, vp, c_i64p, ctypes.c_int32, c_i64p]
    lib.s3s_checksum_ranges.argtypes = ck_args
    lib.s3s_checksum_ranges_device.argtypes = ck_args
    dec_args = [vp, ctypes.c_int, ctypes.c_int, vp, ctypes.c_int64, c_i64p, c_i64p, ctypes.c_int32,
                vp, ctypes.c_int64, c_i64p, ctypes.POINTER(ctypes.c_int32)]
    lib.s3s_decompress_range.argtypes = dec_args
    lib.s3s_decompress_range_device.argtypes = dec_args
    lib.s3s_decompressed_size.argtypes = [vp, ctypes.c_int, vp, ctypes.c_int64, c_i64p]
    c_i32p = ctypes.POINTER(ctypes.c_int32)
    lib.s3s_max_compressed_size_segments.restype = ctypes.c_int64
    lib.s3s_max_compressed_size_segments.argtypes = [vp, ctypes.c_int, c_i64p, ctypes.c_int32]
    seg_args = [vp, ctypes.c_int, ctypes.c_int, vp, c_i64p, ctypes.c_int32, c_i32p, ctypes.c_int32, vp,
                ctypes.c_int64, c_i64p, c_i64p, c_i64p]
    lib.s3s_compress_map_output_segments.argtypes = seg_args
    lib.s3s_compress_map_output_segments_device.argtypes = seg_args
    lib.s3s_compress_map_outputs_batch_device.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(MapTask),
                                                          ctypes.c_int32]
    lib.s3s_decompress_ranges_batch_device.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(FetchRange),
                                                       ctypes.c_int32]
    lib.s3s_compress_map_outputs_batch.argtypes = lib.s3s_compress_map_outputs_batch_device.argtypes
    lib.s3s_decompress_ranges_batch.argtypes = lib.s3s_decompress_ranges_batch_device.argtypes
    lib.s3s_host_alloc.restype = vp
    lib.s3s_host_alloc.argtypes = [ctypes.c_int64]
    lib.s3s_host_free.argtypes = [vp]
    _LIB = lib
    return lib


def device_count() -> int:
    return int(load_library().s3s_device_count())


def _i64(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.int64))


def _p64(a: np.ndarray):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))


def max_compressed_size(codec: int, src_offsets: Sequence[int], ctx: Optional["Codec"] = None) -> int:
    offs = _i64(src_offsets)
    r = load_library().s3s_max_compressed_size(ctx._h if ctx else None, codec, _p64(offs), len(offs) - 1)
    if r < 0:
        raise CodecError(int(r), "s3s_max_compressed_size")
    return int(r)


class Codec:
    """One s3s_ctx: not thread-safe, one per task thread; device = mapId % nGPU by convention
    (mirrors mapId % folderPrefixes, S3ShuffleDispatcher.scala:142)."""

    def __init__(self, device: int = 0, scratch_bytes: int = 0):
        self._lib = load_library()
        self._h = self._lib.s3s_create(int(device), int(scratch_bytes))
        if not self._h:
            msg = self._lib.s3s_last_error(None).decode()
            raise CodecError(E_HIP, f"s3s_create(device={device}) failed: {msg}")
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            self._lib.s3s_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- helpers -------------------------------------------------------------------------
    def _check(self, rc: int, partition: int = -1):
        if rc != 0:
            raise CodecError(int(rc), self._lib.s3s_last_error(self._h).decode(), partition)

    def set_option(self, key: int, value: int):
        self._check(self._lib.s3s_set_option(self._h, key, int(value)))

    def get_option(self, key: int) -> int:
        return int(self._lib.s3s_get_option(self._h, key))

    def stage_ms(self, stage: int) -> float:
        return float(self._lib.s3s_stage_ms(self._h, stage))

    @property
    def stream(self) -> int:
        return int(self._lib.s3s_stream(self._h) or 0)

    def max_compressed_size(self, codec: int, src_offsets) -> int:
        return max_compressed_size(codec, src_offsets, self)

    # ---- map side ----------------------------------------------------------------------------
    def compress_map_output(self, codec: int, checksum: int, src: np.ndarray, src_offsets,
                            dst_capacity: Optional[int] = None, out: Optional[np.ndarray] = None
                            ) -> Tuple[np.ndarray, np.ndarray, Optional[np.ndarray]]:
        """Host buffers in/out.  Returns (data image, index[N+1], checksums[N] or None).
        `out`: caller-owned destination (e.g. a pinned_buffer) instead of a fresh numpy array."""
        src = np.ascontiguousarray(src, dtype=np.uint8)
        offs = _i64(src_offsets)
        n = len(offs) - 1
        cap = self.max_compressed_size(codec, offs) if dst_capacity is None else int(dst_capacity)
        if out is not None:
            cap = min(cap, out.size) if dst_capacity is None else cap
            dst = out
        else:
            dst = np.empty(max(cap, 1), dtype=np.uint8)
        index = np.zeros(n + 1, dtype=np.int64)
        sums = np.zeros(max(n, 1), dtype=np.int64)
        total = ctypes.c_int64(0)
        rc = self._lib.s3s_compress_map_output(
            self._h, codec, checksum, src.ctypes.data, _p64(offs), n, dst.ctypes.data, cap,
            _p64(index), _p64(sums) if checksum != CHECKSUM_NONE else None, ctypes.byref(total))
        self._check(rc)
        return dst[: total.value], index, (sums[:n] if checksum != CHECKSUM_NONE else None)

    def compress_map_output_segments(self, codec: int, checksum: int, src: np.ndarray, seg_offsets,
                                     part_first_seg) -> Tuple[np.ndarray, np.ndarray, Optional[np.ndarray]]:
        """Multi-spill map task: partition p = pieces [part_first_seg[p], part_first_seg[p+1]) of
        seg_offsets, one complete codec stream per non-empty piece (host buffers)."""
        src = np.ascontiguousarray(src, dtype=np.uint8)
        segs = _i64(seg_offsets)
        pfs = np.ascontiguousarray(part_first_seg, dtype=np.int32)
        ns, n = len(segs) - 1, len(pfs) - 1
        cap = int(self._lib.s3s_max_compressed_size_segments(self._h, codec, _p64(segs), ns))
        if cap < 0:
            self._check(cap)
        dst = np.empty(max(cap, 1), dtype=np.uint8)
        index = np.zeros(n + 1, dtype=np.int64)
        sums = np.zeros(max(n, 1), dtype=np.int64)
        total = ctypes.c_int64(0)
        rc = self._lib.s3s_compress_map_output_segments(
            self._h, codec, checksum, src.ctypes.data, _p64(segs), ns,
            pfs.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), n, dst.ctypes.data, cap, _p64(index),
            _p64(sums) if checksum != CHECKSUM_NONE else None, ctypes.byref(total))
        self._check(rc)
        return dst[: total.value], index, (sums[:n] if checksum != CHECKSUM_NONE else None)

    def compress_map_output_device(self, codec: int, checksum: int, d_src: int, src_offsets,
                                   d_dst: int, dst_capacity: int
                                   ) -> Tuple[int, np.ndarray, Optional[np.ndarray]]:
        """Device pointers in/out.  Returns (total bytes, index[N+1], checksums[N] or None)."""
        offs = _i64(src_offsets)
        n = len(offs) - 1
        index = np.zeros(n + 1, dtype=np.int64)
        sums = np.zeros(max(n, 1), dtype=np.int64)
        total = ctypes.c_int64(0)
        rc = self._lib.s3s_compress_map_output_device(
            self._h, codec, checksum, ctypes.c_void_p(d_src), _p64(offs), n, ctypes.c_void_p(d_dst),
            int(dst_capacity), _p64(index), _p64(sums) if checksum != CHECKSUM_NONE else None,
            ctypes.byref(total))
        self._check(rc)
        return int(total.value), index, (sums[:n] if checksum != CHECKSUM_NONE else None)

    def compress_map_outputs_batch(self, codec: int, checksum: int, tasks):
        """Batched HOST-buffer form (what the JNI shim binds): `tasks` = [(src_address, src_offsets, dst_address,
        dst_capacity), ...] with host addresses (e.g. `PinnedBuffer.address`, or `ndarray.ctypes.data`) -> per task
        (total bytes, index[N+1], checksums[N] or None).  Upload, codec and download of consecutive groups overlap."""
        return self.compress_map_outputs_batch_device(codec, checksum, tasks, _host=True)

    def compress_map_outputs_batch_device(self, codec: int, checksum: int, tasks, _host: bool = False):
        """Batched device form: `tasks` = [(d_src, src_offsets, d_dst, dst_capacity), ...] -> per task
        (total bytes, index[N+1], checksums[N] or None).  One codec launch and one stream sync for all."""
        arr = (MapTask * len(tasks))()
        keep = []
        for i, (d_src, src_offsets, d_dst, cap) in enumerate(tasks):
            offs = _i64(src_offsets)
            n = len(offs) - 1
            index = np.zeros(n + 1, dtype=np.int64)
            sums = np.zeros(max(n, 1), dtype=np.int64)
            keep.append((offs, index, sums, n))
            arr[i].d_src = d_src
            arr[i].src_offsets = _p64(offs)
            arr[i].num_partitions = n
            arr[i].d_dst = d_dst
            arr[i].dst_capacity = int(cap)
            arr[i].out_index = _p64(index)
            arr[i].out_checksums = _p64(sums) if checksum != CHECKSUM_NONE else None
        fn = self._lib.s3s_compress_map_outputs_batch if _host else self._lib.s3s_compress_map_outputs_batch_device
        rc = fn(self._h, codec, checksum, arr, len(tasks))
        self._check(rc)
        return [(int(arr[i].out_total), k[1], (k[2][:k[3]] if checksum != CHECKSUM_NONE else None))
                for i, k in enumerate(keep)]

    # ---- checksum only -----------------------------------------------------------------------
    def checksum_ranges(self, algo: int, data: np.ndarray, offsets) -> np.ndarray:
        data = np.ascontiguousarray(data, dtype=np.uint8)
        offs = _i64(offsets)
        n = len(offs) - 1
        out = np.zeros(max(n, 1), dtype=np.int64)
        self._check(self._lib.s3s_checksum_ranges(self._h, algo, data.ctypes.data, _p64(offs), n, _p64(out)))
        return out[:n]

    def checksum_ranges_device(self, algo: int, d_data: int, offsets) -> np.ndarray:
        offs = _i64(offsets)
        n = len(offs) - 1
        out = np.zeros(max(n, 1), dtype=np.int64)
        self._check(self._lib.s3s_checksum_ranges_device(self._h, algo, ctypes.c_void_p(d_data), _p64(offs), n, _p64(out)))
        return out[:n]

    # ---- reduce side ---------------------------------------------------------------------------
    def decompressed_size(self, codec: int, comp: np.ndarray) -> int:
        comp = np.ascontiguousarray(comp, dtype=np.uint8)
        out = ctypes.c_int64(0)
        self._check(self._lib.s3s_decompressed_size(self._h, codec, comp.ctypes.data, comp.size, ctypes.byref(out)))
        return int(out.value)

    def decompress_range(self, codec: int, checksum: int, comp: np.ndarray, part_offsets,
                         ref_checksums=None, dst_capacity: Optional[int] = None,
                         out: Optional[np.ndarray] = None) -> np.ndarray:
        comp = np.ascontiguousarray(comp, dtype=np.uint8)
        offs = _i64(part_offsets)
        n = len(offs) - 1
        refs = _i64(ref_checksums) if ref_checksums is not None else None
        if out is not None:
            cap = out.size if dst_capacity is None else int(dst_capacity)
            dst = out
        else:
            cap = self.decompressed_size(codec, comp) if dst_capacity is None else int(dst_capacity)
            dst = np.empty(max(cap, 1), dtype=np.uint8)
        out_len = ctypes.c_int64(0)
        bad = ctypes.c_int32(-1)
        rc = self._lib.s3s_decompress_range(
            self._h, codec, checksum, comp.ctypes.data, comp.size, _p64(offs),
            _p64(refs) if refs is not None else None, n, dst.ctypes.data, cap,
            ctypes.byref(out_len), ctypes.byref(bad))
        self._check(rc, bad.value)
        return dst[: out_len.value]

    def decompress_range_device(self, codec: int, checksum: int, d_comp: int, comp_len: int,
                                part_offsets, ref_checksums, d_dst: int, dst_capacity: int) -> int:
        offs = _i64(part_offsets)
        n = len(offs) - 1
        refs = _i64(ref_checksums) if ref_checksums is not None else None
        out_len = ctypes.c_int64(0)
        bad = ctypes.c_int32(-1)
        rc = self._lib.s3s_decompress_range_device(
            self._h, codec, checksum, ctypes.c_void_p(d_comp), int(comp_len), _p64(offs),
            _p64(refs) if refs is not None else None, n, ctypes.c_void_p(d_dst), int(dst_capacity),
            ctypes.byref(out_len), ctypes.byref(bad))
        self._check(rc, bad.value)
        return int(out_len.value)



    def decompress_ranges_batch(self, codec: int, checksum: int, ranges, raise_on_error: bool = True):
        """Batched HOST-buffer form: `ranges` = [(comp_address, comp_len, part_offsets, ref_checksums, dst_address,
        dst_capacity), ...] with host addresses -> per range (status, decoded bytes, bad partition)."""
        return self.decompress_ranges_batch_device(codec, checksum, ranges, raise_on_error, _host=True)

    def decompress_ranges_batch_device(self, codec: int, checksum: int, ranges, raise_on_error: bool = True,
                                       _host: bool = False):
        """Batched device form: `ranges` = [(d_comp, comp_len, part_offsets, ref_checksums, d_dst, dst_capacity), ...]
        -> per range (status, decoded bytes, bad partition).  One decode launch for all, three stream syncs."""
        arr = (FetchRange * len(ranges))()
        keep = []
        for i, (d_comp, comp_len, part_offsets, refs, d_dst, cap) in enumerate(ranges):
            offs = _i64(part_offsets)
            r = _i64(refs) if refs is not None else None
            keep.append((offs, r))
            arr[i].d_comp = d_comp
            arr[i].comp_len = int(comp_len)
            arr[i].part_offsets = _p64(offs)
            arr[i].ref_checksums = _p64(r) if r is not None else None
            arr[i].num_partitions = len(offs) - 1
            arr[i].d_dst = d_dst
            arr[i].dst_capacity = int(cap)
        fn = self._lib.s3s_decompress_ranges_batch if _host else self._lib.s3s_decompress_ranges_batch_device
        rc = fn(self._h, codec, checksum, arr, len(ranges))
        out = [(int(arr[i].status), int(arr[i].out_len), int(arr[i].bad_partition)) for i in range(len(ranges))]
        if raise_on_error:
            bad = next((i for i, o in enumerate(out) if o[0] != 0), -1)
            self._check(rc, out[bad][2] if bad >= 0 else -1)
        return out


class PinnedBuffer:
    """Page-locked host memory from s3s_host_alloc (what the JVM shim would wrap with
    NewDirectByteBuffer).  `.array` is a uint8 view; the host-buffer entry points move it with
    plain DMA.  Keep the object alive while views of `.array` are in use; `free()` is explicit."""

    def __init__(self, nbytes: int):
        self._lib = load_library()
        self.nbytes = max(int(nbytes), 1)
        self.ptr = self._lib.s3s_host_alloc(self.nbytes)
        if not self.ptr:
            raise MemoryError(f"s3s_host_alloc({self.nbytes}) failed")
        self.array = np.frombuffer((ctypes.c_uint8 * self.nbytes).from_address(self.ptr), dtype=np.uint8)

    def free(self) -> None:
        if self.ptr:
            self.array = None
            self._lib.s3s_host_free(ctypes.c_void_p(self.ptr))
            self.ptr = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.free()

    def __del__(self):
        try:
            self.free()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass
