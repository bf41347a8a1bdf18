"""SURVEY §8 f1, executed: jni/s3s_jni.c runs against a mock JNIEnv (tests/mock_jni/mock_jvm.h — copy-in / copy-out array
semantics, counted pins and local references) and every native of S3SCodec.scala is called the way the Scala shim calls
it (tests/mock_jni/jni_exec.c).  There is no JDK in this image, so this is as close to a JVM as the binding gets here.

* CPU: the harness is linked with a toy stand-in for the library (tests/mock_jni/fake_codec.c) under ASan / UBSan — what is
  under test is the translation unit: which array goes to which argument, which arrays are copied back, what is released.
  Two mutants of the shim (an output array released with JNI_ABORT, a local reference that is never deleted) must FAIL.
* GPU (`-m gpu`): the same harness linked with the real libs3shuffle_codec: the natives' results equal the C-ABI called
  directly (LZ4 + Adler32, three map tasks, single and batched forms) and the reduce-side natives return the sources."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MOCK = os.path.join(ROOT, "tests", "mock_jni")
JNI_C = os.path.join(ROOT, "jni", "s3s_jni.c")
BASE = ["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-g", "-I", MOCK, "-I", os.path.join(ROOT, "include")]


def _build_fake(tmp_path, shim, name):
    exe = str(tmp_path / name)
    subprocess.run(BASE + ["-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", shim,
                           os.path.join(MOCK, "jni_exec.c"), os.path.join(MOCK, "fake_codec.c"), "-o", exe], check=True)
    return exe


def _run(exe, leaks=1):
    return subprocess.run([exe], capture_output=True, text=True, timeout=300,
                          env=dict(os.environ, ASAN_OPTIONS="detect_leaks=%d" % leaks))


def test_jni_shim_executes_against_the_mock_jvm(tmp_path):
    r = _run(_build_fake(tmp_path, JNI_C, "jni_exec_fake"))
    assert r.returncode == 0 and "jni_exec ok" in r.stdout, (r.stdout, r.stderr[-2000:])


@pytest.mark.parametrize("old,new", [
    ("unpin(e, arrs[3 * i + 1], (jlong*)t[i].out_index, 0);", "unpin(e, arrs[3 * i + 1], (jlong*)t[i].out_index, JNI_ABORT);"),
    ("unpin(e, outChecksums, oc, 0);", "unpin(e, outChecksums, oc, JNI_ABORT);"),
    ("    (*e)->DeleteLocalRef(e, db);\n    r[i].comp_len", "    r[i].comp_len"),
    ("(*e)->ReleaseIntArrayElements(e, outBadPartition, ob, 0);", "(*e)->ReleaseIntArrayElements(e, outBadPartition, ob, JNI_ABORT);"),
], ids=["batch-index-not-copied-back", "checksums-not-copied-back", "leaked-local-reference", "bad-partition-not-copied-back"])
def test_the_harness_sees_a_broken_shim(tmp_path, old, new):
    src = open(JNI_C).read()
    assert src.count(old) >= 1
    mutant = tmp_path / "s3s_jni_mutant.c"
    mutant.write_text(src.replace(old, new, 1))
    r = _run(_build_fake(tmp_path, str(mutant), "jni_exec_mutant"), leaks=0)  # (the harness stops at the first failed check)
    assert r.returncode != 0 and "FAILED" in r.stdout, (r.stdout, r.stderr[-2000:])


@pytest.mark.gpu
def test_jni_shim_executes_against_the_real_library(tmp_path, gpu_codec):
    import s3shuffle

    lib_dir = os.path.dirname(s3shuffle.library_path())
    exe = str(tmp_path / "jni_exec_real")
    subprocess.run(BASE + ["-O2", JNI_C, os.path.join(MOCK, "jni_exec.c"), "-L", lib_dir, "-ls3shuffle_codec",
                           "-Wl,-rpath," + lib_dir, "-Wl,--allow-shlib-undefined", "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "jni_exec ok" in r.stdout, (r.stdout, r.stderr[-2000:])
