"""Thin ctypes binding of the MI355X shuffle-block codec C-ABI (include/s3shuffle_codec.h).

This package is plumbing for tests, bench.py and __graft_entry__: the product is the shared
library `spark-s3-shuffle_amd/lib/libs3shuffle_codec.so` (HIP kernels for gfx950 + C-ABI) that
a JNI shim binds from the unchanged S3ShuffleManager / S3ShuffleDataIO plugin.
There is no CPU fallback: importing works anywhere, but creating a Codec without the built
library or without a HIP device raises.
"""
from .codec import (  # noqa: F401
    CHECKSUM_ADLER32,
    CHECKSUM_CRC32,
    CHECKSUM_CRC32C,
    CHECKSUM_NONE,
    CODEC_LZ4,
    CODEC_LZF,
    CODEC_NONE,
    CODEC_SNAPPY,
    CODEC_ZSTD,
    Codec,
    CodecError,
    PinnedBuffer,
    device_count,
    library_path,
    load_library,
    max_compressed_size,
)
