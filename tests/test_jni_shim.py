"""SURVEY §8 f1 as source: the JNI translation unit (jni/s3s_jni.c) must compile, warning-free, against a mock
<jni.h> and the real include/s3shuffle_codec.h — so every wrapper's argument list is type-checked against the C-ABI —
and the Scala side (scala/.../S3SCodec.scala) must declare exactly the natives the C file defines, with the same
number of parameters.  No JDK is needed (there is none in this image)."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JNI_C = os.path.join(ROOT, "jni", "s3s_jni.c")
SCALA = os.path.join(ROOT, "scala", "org", "apache", "spark", "shuffle", "gpu", "S3SCodec.scala")


def test_jni_translation_unit_type_checks_against_the_header(tmp_path):
    obj = tmp_path / "s3s_jni.o"
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-fPIC", "-c",
                    "-I", os.path.join(ROOT, "tests", "mock_jni"), "-I", os.path.join(ROOT, "include"),
                    JNI_C, "-o", str(obj)], check=True)
    syms = subprocess.run(["nm", "--defined-only", str(obj)], check=True, capture_output=True, text=True).stdout
    # S3SCodec is a Scala object: its natives live on the module class S3SCodec$ ("$" = _00024 in JNI's name mangling)
    assert not re.search(r"Java_org_apache_spark_shuffle_gpu_S3SCodec_(?!00024_)", syms)
    exported = set(re.findall(r"Java_org_apache_spark_shuffle_gpu_S3SCodec_00024_(\w+)", syms))
    assert {"create", "destroy", "compressMapOutput", "compressMapOutputSegments", "decompressRange",
            "checksumRanges", "maxCompressedSize", "decompressedSize", "hostAlloc", "hostFree", "lastError",
            "compressMapOutputsBatch", "decompressRangesBatch"} <= exported
    # every C-ABI function the wrappers call exists in the header (the compile above checked the types)
    used = set(re.findall(r"\b(s3s_[a-z0-9_]+)\s*\(", open(JNI_C).read()))
    header = open(os.path.join(ROOT, "include", "s3shuffle_codec.h")).read()
    for fn in used:
        assert re.search(r"\b%s\s*\(" % fn, header), fn


def _c_natives():
    src = re.sub(r"/\*.*?\*/", "", open(JNI_C).read(), flags=re.S)
    out = {}
    for m in re.finditer(r"FN\((\w+)\)\s*\(([^)]*)\)", src):
        params = [p for p in m.group(2).split(",") if p.strip()]
        out[m.group(1)] = len(params) - 2  # JNIEnv*, jclass
    return out


def _scala_natives():
    src = open(SCALA).read()
    out = {}
    for m in re.finditer(r"@native def (\w+)\(([^)]*)\)", src, flags=re.S):
        params = [p for p in m.group(2).split(",") if p.strip()]
        out[m.group(1)] = len(params)
    return out


def test_scala_natives_match_the_c_side():
    c, s = _c_natives(), _scala_natives()
    assert c and s
    assert set(c) == set(s), (sorted(set(c) - set(s)), sorted(set(s) - set(c)))
    for name in c:
        assert c[name] == s[name], (name, c[name], s[name])


def test_scala_constants_match_the_header():
    header = open(os.path.join(ROOT, "include", "s3shuffle_codec.h")).read()
    scala = open(SCALA).read()
    abi = int(re.search(r"#define\s+S3S_ABI_VERSION\s+(\d+)", header).group(1))
    assert int(re.search(r"val ABI_VERSION = (\d+)", scala).group(1)) == abi
    for name, sc in (("S3S_E_INVALID", "E_INVALID"), ("S3S_E_CAPACITY", "E_CAPACITY"), ("S3S_E_BAD_FRAME", "E_BAD_FRAME"),
                     ("S3S_E_CHECKSUM", "E_CHECKSUM")):
        hv = int(re.search(r"%s\s*=\s*(-?\d+)" % name, header).group(1))
        sv = int(re.search(r"val %s = (-?\d+)" % sc, scala).group(1))
        assert hv == sv, name


_SCALA_TO_JNI = {"Int": "jint", "Long": "jlong", "ByteBuffer": "jobject", "String": "jstring", "Unit": "void",
                 "Array[Long]": "jlongArray", "Array[Int]": "jintArray", "Array[ByteBuffer]": "jobjectArray",
                 "Array[Array[Long]]": "jobjectArray"}


def test_scala_native_types_match_the_c_side():
    """not only the number of parameters: every parameter's and the result's JNI type (Int -> jint, Array[Long] -> jlongArray,
    Array[Array[Long]] -> jobjectArray, ...) is what the C wrapper declares — an Int passed where the C side reads a jlong
    would only show up as garbage at run time"""
    c_src = re.sub(r"/\*.*?\*/", "", open(JNI_C).read(), flags=re.S)
    c = {}
    for m in re.finditer(r"JNIEXPORT\s+(\w+)\s+JNICALL\s+FN\((\w+)\)\s*\(([^)]*)\)", c_src):
        params = [p.strip() for p in m.group(3).split(",") if p.strip()][2:]  # (JNIEnv*, jclass)
        c[m.group(2)] = (m.group(1), [p.split()[0] for p in params])
    s_src = open(SCALA).read()
    seen = 0
    for m in re.finditer(r"@native def (\w+)\(([^)]*)\)\s*:\s*([\w\[\]]+)", s_src, flags=re.S):
        name, params, ret = m.group(1), m.group(2), m.group(3)
        types = [p.split(":", 1)[1].strip() for p in re.split(r",(?![^\[]*\])", params) if p.strip()]
        want = (_SCALA_TO_JNI[ret], [_SCALA_TO_JNI[t] for t in types])
        assert c[name] == want, (name, c[name], want)
        seen += 1
    assert seen == len(c) >= 17
