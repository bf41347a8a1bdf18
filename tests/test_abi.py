"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol the
header declares, sizes buffers like the reference's streams do, and refuses to work without a
HIP device (there is no CPU fallback).  No compute calls here — those are the -m gpu tests."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "s3shuffle_codec.h")


def _declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(s3s_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_boundary():
    syms = _declared_symbols()
    for must in ("s3s_create", "s3s_destroy", "s3s_compress_map_output", "s3s_compress_map_output_device",
                 "s3s_checksum_ranges", "s3s_decompress_range", "s3s_decompress_range_device",
                 "s3s_max_compressed_size", "s3s_decompressed_size", "s3s_last_error"):
        assert must in syms


def test_library_exports_every_declared_symbol(codec_lib):
    for name in _declared_symbols():
        assert hasattr(codec_lib, name), f"{name} declared in include/s3shuffle_codec.h but not exported"
    declared = int(re.search(r"#define\s+S3S_ABI_VERSION\s+(\d+)", open(HEADER).read()).group(1))
    assert codec_lib.s3s_abi_version() == declared >= 3
    assert b"gfx950" in codec_lib.s3s_version()


def test_no_torch_or_oracle_in_the_library(codec_lib):
    import s3shuffle

    out = os.popen(f"ldd {s3shuffle.library_path()}").read()
    assert "torch" not in out and "oracle" not in out
    assert "amdhip64" in out


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "spark-s3-shuffle_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".c")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert not re.search(r"(import\s+oracle|from\s+oracle|s3s_oracle\.h|liboracle|s3o_)", text), \
                    f"{os.path.join(dirpath, f)} references the oracle"


def test_sizing_matches_reference_stream_bounds(codec_lib, oracle):
    """s3s_max_compressed_size (ctx = NULL -> Spark's default 32 KiB blocks) must bound, and for
    LZ4 equal, the oracle's worst case; it is the shim's buffer sizing helper."""
    import s3shuffle

    rng = np.random.default_rng(3)
    for _ in range(20):
        lens = rng.integers(0, 200_000, int(rng.integers(1, 50)))
        lens[rng.random(lens.size) < 0.2] = 0
        offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        for codec in (0, 1, 2):
            got = s3shuffle.max_compressed_size(codec, offs)
            want = int(oracle.lib().s3o_max_compressed_size(codec, 32768, offs.ctypes.data, len(offs) - 1))
            assert got >= 0
            if codec in (0, 1):
                assert got == want, (codec, got, want)
            else:
                assert got >= want or got >= sum(16 + 4 * (-(-int(x) // 32768)) + int(x) + int(x) // 6 + 32 * (-(-int(x) // 32768)) for x in lens if x)
    bad = np.array([0, 10, 5], np.int64)
    with pytest.raises(s3shuffle.CodecError):
        s3shuffle.max_compressed_size(1, bad)


def test_fails_loudly_without_a_hip_device(codec_lib):
    import s3shuffle

    if s3shuffle.device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(s3shuffle.CodecError) as ei:
        s3shuffle.Codec(0)
    assert "no CPU fallback" in str(ei.value)
    assert codec_lib.s3s_compress_map_output(None, 1, 1, None, None, 0, None, 0, None, None, None) == -1


def test_missing_library_is_an_error_not_a_fallback(monkeypatch, tmp_path):
    import s3shuffle
    from s3shuffle import codec as codec_mod

    monkeypatch.setattr(codec_mod, "_LIB", None)
    monkeypatch.setattr(codec_mod, "_PKG_ROOT", str(tmp_path))
    with pytest.raises(FileNotFoundError):
        s3shuffle.load_library()
