import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "spark-s3-shuffle_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (test infrastructure) — builds oracle/liboracle.so on demand."""
    from oracle import binding

    binding.lib()
    return binding


@pytest.fixture(scope="session")
def codec_lib():
    """The product library; built on demand on the CPU box (hipcc cross-compiles)."""
    import s3shuffle

    if not os.path.exists(s3shuffle.library_path()):
        import subprocess

        subprocess.run(["make", "-C", os.path.join(PKG, "csrc"), "-j", "8"], check=True)
    return s3shuffle.load_library()


@pytest.fixture(scope="session")
def gpu_codec(codec_lib):
    import s3shuffle

    if s3shuffle.device_count() < 1:
        pytest.fail("gpu-marked test running without a HIP device: there is no CPU fallback")
    c = s3shuffle.Codec(0)
    yield c
    c.close()
