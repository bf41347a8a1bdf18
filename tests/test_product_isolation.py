"""The oracle is test infrastructure: nothing the product ships may import, link or include it, and the product
has no CPU codec fallback (brief ③).  Static checks over the sources and the built libraries."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "spark-s3-shuffle_amd")


def _files(top, exts):
    for d, _, names in os.walk(top):
        if "build" in d.split(os.sep) or "__pycache__" in d:
            continue
        for n in names:
            if n.endswith(exts):
                yield os.path.join(d, n)


def test_product_sources_never_touch_the_oracle():
    pat = re.compile(r"(^\s*(from|import)\s+oracle\b)|(#include\s+[\"<][^\">]*oracle)|(liboracle)|(s3o_)", re.M)
    offenders = []
    for f in _files(PKG, (".py", ".h", ".hip", ".cpp", ".c", "Makefile")):
        if pat.search(open(f, errors="ignore").read()):
            offenders.append(os.path.relpath(f, ROOT))
    for f in [os.path.join(ROOT, "include", "s3shuffle_codec.h")]:
        if pat.search(open(f).read()):
            offenders.append(os.path.relpath(f, ROOT))
    assert not offenders, offenders


def test_built_libraries_do_not_link_the_oracle_or_a_cpu_codec(codec_lib):
    for lib in ("libs3shuffle_codec.so", "libs3shuffle_host.so"):
        path = os.path.join(PKG, "lib", lib)
        assert os.path.exists(path), path
        needed = subprocess.run(["readelf", "-d", path], capture_output=True, text=True, check=True).stdout
        libs = re.findall(r"\(NEEDED\)\s+Shared library: \[([^\]]+)\]", needed)
        assert not [x for x in libs if re.search(r"oracle|lz4|snappy|libz\.", x)], (lib, libs)
        if lib == "libs3shuffle_codec.so":
            assert any("amdhip64" in x for x in libs), libs


def test_oracle_is_only_used_by_tests_bench_baseline_and_smoke():
    users = []
    for f in _files(ROOT, (".py",)):
        rel = os.path.relpath(f, ROOT)
        if rel.startswith(("tests" + os.sep, "oracle" + os.sep, "gpurun_out" + os.sep)):
            continue
        if re.search(r"^\s*(from|import)\s+oracle\b", open(f, errors="ignore").read(), re.M):
            users.append(rel)
    allowed = {"bench.py", "__graft_entry__.py"}
    assert set(users) <= allowed, users
