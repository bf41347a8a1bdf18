"""Generates the committed golden fixtures under tests/golden/ — WITHOUT the oracle.

The reference repo holds no golden vectors for the codec path and cannot run here
(no JVM), so the fixtures are produced by an independent pure-Python assembly of the on-store
format on top of the third-party native code the JVM path itself reaches:

    LZ4   payload  = liblz4 1.9.3 LZ4_compress_default (ctypes; what lz4-java's JNI instance calls)
          check    = python-xxhash xxh32(seed 0x9747b28c) & 0x0FFFFFFF
          framing  = lz4-java 1.8.0 LZ4BlockOutputStream (written out below from its format)
    SNAPPY payload = libsnappy 1.1.8 snappy_compress; framing = snappy-java SnappyOutputStream
    checksum       = zlib adler32 / crc32 over the compressed partition bytes
    index          = cumulative big-endian longs (S3ShuffleHelper.scala:44-59)

Each fixture `<name>.bin` = .data image ‖ .index image ‖ .checksum image; inputs are regenerated
from seeds (tests/corpus.py, s3shuffle/datagen.py), so only outputs are stored.

    python tests/golden/make_golden.py        # rewrites the .bin files and manifest.json
"""
import ctypes
import hashlib
import json
import os
import struct
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (os.path.dirname(HERE), os.path.join(ROOT, "spark-s3-shuffle_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

LZ4, SNAPPY, NONE = 1, 2, 0
ADLER, CRC = 1, 2
BLOCK = 32768

CASES = [
    {"name": "lz4_adler_kvint", "codec": LZ4, "checksum": ADLER, "gen": "kv_int", "args": [20000, 4, 1]},
    {"name": "lz4_crc_terasort", "codec": LZ4, "checksum": CRC, "gen": "terasort", "args": [150000, 7, 2]},
    {"name": "lz4_adler_ragged", "codec": LZ4, "checksum": ADLER, "gen": "ragged", "args": [11, 9, 70000]},
    {"name": "lz4_crc_skew_random", "codec": LZ4, "checksum": CRC, "gen": "skew", "args": [70000, "random", 5]},
    {"name": "lz4_adler_skew_zeros", "codec": LZ4, "checksum": ADLER, "gen": "skew", "args": [100000, "zeros", 5]},
    {"name": "snappy_adler_ragged", "codec": SNAPPY, "checksum": ADLER, "gen": "ragged", "args": [12, 6, 50000]},
    {"name": "none_crc_ragged", "codec": NONE, "checksum": CRC, "gen": "ragged", "args": [13, 5, 3000]},
]


def case_input(case):
    import corpus
    from s3shuffle import datagen

    g, a = case["gen"], case["args"]
    if g == "kv_int":
        return datagen.kv_int_map_output(a[0], a[1], seed=a[2])
    if g == "terasort":
        return datagen.terasort_map_output(a[0], a[1], seed=a[2])
    if g == "skew":
        return datagen.skew_block(a[0], a[1], seed=a[2])
    if g == "ragged":
        return corpus.ragged_map_output(np.random.default_rng(a[0]), a[1], a[2])
    raise ValueError(g)


def sha256(b: bytes) -> str:
    return hashlib.sha256(b).hexdigest()


def split_blob(blob: bytes, n: int):
    tail = 8 * (n + 1) + 8 * n
    return blob[:-tail], blob[-tail:-8 * n] if n else blob[-tail:], blob[len(blob) - 8 * n:]


def _lz4_stream(L, part: bytes) -> bytes:
    import xxhash

    if not part:
        return b""
    out = bytearray()
    for pos in range(0, len(part), BLOCK):
        chunk = part[pos:pos + BLOCK]
        cap = L.LZ4_compressBound(len(chunk))
        buf = ctypes.create_string_buffer(cap)
        n = L.LZ4_compress_default(chunk, buf, len(chunk), cap)
        check = xxhash.xxh32(chunk, seed=0x9747B28C).intdigest() & 0x0FFFFFFF
        if n >= len(chunk):
            out += b"LZ4Block" + bytes([0x15]) + struct.pack("<iiI", len(chunk), len(chunk), check) + chunk
        else:
            out += b"LZ4Block" + bytes([0x25]) + struct.pack("<iiI", n, len(chunk), check) + buf.raw[:n]
    out += b"LZ4Block" + bytes([0x15]) + struct.pack("<iii", 0, 0, 0)
    return bytes(out)


def _snappy_stream(S, part: bytes) -> bytes:
    if not part:
        return b""
    out = bytearray(b"\x82SNAPPY\x00" + struct.pack(">ii", 1, 1))
    for pos in range(0, len(part), BLOCK):
        chunk = part[pos:pos + BLOCK]
        buf = ctypes.create_string_buffer(32 + len(chunk) + len(chunk) // 6)
        olen = ctypes.c_size_t(len(buf))
        assert S.snappy_compress(chunk, ctypes.c_size_t(len(chunk)), buf, ctypes.byref(olen)) == 0
        out += struct.pack(">i", olen.value) + buf.raw[:olen.value]
    return bytes(out)


def build_blob(case) -> bytes:
    L = ctypes.CDLL("liblz4.so.1")
    L.LZ4_versionString.restype = ctypes.c_char_p
    assert L.LZ4_versionString() == b"1.9.3"
    S = None
    if case["codec"] == SNAPPY:
        for name in ("libsnappy.so.1", "/opt/conda/lib/libsnappy.so.1"):
            try:
                S = ctypes.CDLL(name)
                break
            except OSError:
                pass
    data, offsets = case_input(case)
    raw = data.tobytes()
    image, index, sums = bytearray(), [0], []
    for p in range(len(offsets) - 1):
        part = raw[int(offsets[p]):int(offsets[p + 1])]
        s = {LZ4: lambda: _lz4_stream(L, part), SNAPPY: lambda: _snappy_stream(S, part), NONE: lambda: part}[case["codec"]]()
        image += s
        index.append(len(image))
        sums.append(zlib.adler32(s) if case["checksum"] == ADLER else zlib.crc32(s))
    n = len(offsets) - 1
    return bytes(image) + struct.pack(f">{n + 1}q", *index) + struct.pack(f">{n}q", *sums)


def main():
    manifest = {"generator": "tests/golden/make_golden.py", "liblz4": "1.9.3", "libsnappy": "1.1.8", "cases": []}
    for case in CASES:
        blob = build_blob(case)
        with open(os.path.join(HERE, case["name"] + ".bin"), "wb") as f:
            f.write(blob)
        manifest["cases"].append(dict(case, sha256=sha256(blob), bytes=len(blob)))
        print(case["name"], len(blob))
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1)


if __name__ == "__main__":
    main()
