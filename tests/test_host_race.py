"""Race detection and memory checking of the C++ host mirror (the reference's CI has no race detector; its
prefetcher is the one multi-threaded piece of the path).  spark-s3-shuffle_amd/host/*.cpp is compiled together with
tests/host_race/host_race_driver.cpp and a toy stand-in for the codec library (tests/mock_jni/fake_codec.c — no GPU here)
under ThreadSanitizer, and again under ASan / UBSan (with leak detection): four task threads write map outputs at the same
time, four read through the prefetch pipeline at the same time (fetch threads blocked on a two-buffer budget, a block
larger than the whole budget, batches, the ThreadPredictor), iterators are abandoned with blocks in flight and in the
consumer's hands, a damaged object raises in the consumer.  Any sanitizer report fails; so does a wrong byte."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "spark-s3-shuffle_amd", "host")


@pytest.mark.parametrize("san", ["thread", "address,undefined"], ids=["tsan", "asan-ubsan"])
def test_host_mirror_under_sanitizers(tmp_path, san):
    flags = ["-O1", "-g", "-fsanitize=" + san, "-fno-sanitize-recover=undefined"]
    fake = str(tmp_path / "fake_codec.o")
    subprocess.run(["gcc", *flags, "-c", os.path.join(ROOT, "tests", "mock_jni", "fake_codec.c"),
                    "-I", os.path.join(ROOT, "include"), "-o", fake], check=True)
    exe = str(tmp_path / "host_race")
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", *flags, "-pthread", os.path.join(HOST, "s3shuffle_host.cpp"),
                    os.path.join(HOST, "s3shuffle_prefetch.cpp"), os.path.join(ROOT, "tests", "host_race", "host_race_driver.cpp"),
                    fake, "-o", exe], check=True)
    store = tmp_path / "store"
    store.mkdir()
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 second_deadlock_stack=1", ASAN_OPTIONS="detect_leaks=1")
    r = subprocess.run([exe, str(store)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "host_race ok" in r.stdout, (r.stdout[-2000:], r.stderr[-6000:])
    assert "WARNING: ThreadSanitizer" not in r.stderr and "ERROR: AddressSanitizer" not in r.stderr, r.stderr[-6000:]
