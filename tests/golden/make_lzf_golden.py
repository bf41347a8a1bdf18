"""Generates tests/golden/lzf_liblzf.npz: raw inputs and their liblzf 3.6 encodings (Marc Lehmann's C library — the block
format compress-lzf implements — reached through `imagecodecs.lzf_encode` of the image's conda python3.9).  Run with THAT
interpreter:   /opt/conda/bin/python3.9 tests/golden/make_lzf_golden.py
Also usable as a filter for live pins (tests/test_oracle_pins.py):
   ... make_lzf_golden.py --encode  < raw   > lzf      (stdin -> stdout)
   ... make_lzf_golden.py --decode N < lzf   > raw
   ... make_lzf_golden.py --streams  (length-prefixed partitions in, length-prefixed LZFOutputStream images out: bench.py)"""
import os
import sys

import imagecodecs
import numpy as np

if len(sys.argv) > 1 and sys.argv[1] == "--encode":
    sys.stdout.buffer.write(bytes(imagecodecs.lzf_encode(sys.stdin.buffer.read())))
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "--stream":
    # stdin: one partition's bytes -> stdout: what LZFOutputStream writes for it (chunks of at most 65 535 bytes, liblzf
    # blocks, a chunk that does not shrink by at least two bytes stored) - tests/tools/lzf_bench.py builds its inputs with this
    import struct

    raw = sys.stdin.buffer.read()
    out = bytearray()
    for p in range(0, len(raw), 0xFFFF):
        chunk = raw[p:p + 0xFFFF]
        try:
            enc = bytes(imagecodecs.lzf_encode(chunk))
        except Exception:
            enc = b""
        if not enc or len(enc) >= len(chunk) - 2:
            out += b"ZV\x00" + struct.pack(">H", len(chunk)) + chunk
        else:
            out += b"ZV\x01" + struct.pack(">HH", len(enc), len(chunk)) + enc
    sys.stdout.buffer.write(bytes(out))
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "--streams":
    # stdin: <u64 LE length><bytes> per partition -> stdout: <u64 LE length><LZFOutputStream bytes> per partition
    # (bench.py builds a whole map task's LZF image with ONE interpreter start)
    import struct

    def stream(raw):
        out = bytearray()
        for p in range(0, len(raw), 0xFFFF):
            chunk = raw[p:p + 0xFFFF]
            try:
                enc = bytes(imagecodecs.lzf_encode(chunk))
            except Exception:
                enc = b""
            if not enc or len(enc) >= len(chunk) - 2:
                out += b"ZV\x00" + struct.pack(">H", len(chunk)) + chunk
            else:
                out += b"ZV\x01" + struct.pack(">HH", len(enc), len(chunk)) + enc
        return bytes(out)

    buf = sys.stdin.buffer.read()
    pos, w = 0, sys.stdout.buffer
    while pos < len(buf):
        (n,) = struct.unpack_from("<Q", buf, pos)
        s_ = stream(buf[pos + 8:pos + 8 + n])
        w.write(struct.pack("<Q", len(s_)))
        w.write(s_)
        pos += 8 + n
    sys.exit(0)
if len(sys.argv) > 2 and sys.argv[1] == "--decode":
    sys.stdout.buffer.write(bytes(imagecodecs.lzf_decode(sys.stdin.buffer.read(), out=int(sys.argv[2]))))
    sys.exit(0)

rng = np.random.default_rng(20260923)
cases = {}


def add(name, raw):
    raw = np.ascontiguousarray(raw, dtype=np.uint8)
    enc = np.frombuffer(bytes(imagecodecs.lzf_encode(raw.tobytes())), np.uint8)
    assert bytes(imagecodecs.lzf_decode(enc.tobytes(), out=raw.size)) == raw.tobytes()
    cases["raw_" + name] = raw
    cases["lzf_" + name] = enc


rec = np.frombuffer(b"0123456789ABCDEF" * 4 + b"....................................\r\n", np.uint8)
rows = np.tile(rec, 400).copy()
rows[::100] = rng.integers(0, 256, rows[::100].size)
add("records", rows[:30000])
add("zeros", np.zeros(5000, np.uint8))
add("runs", np.repeat(rng.integers(0, 256, 300, dtype=np.uint8), rng.integers(1, 40, 300)))
add("text", np.frombuffer((b"the quick brown fox jumps over the lazy dog; " * 300)[:12000], np.uint8))
add("mixed", np.concatenate([rng.integers(0, 256, 2000, dtype=np.uint8), np.tile(rng.integers(0, 256, 37, dtype=np.uint8), 200),
                             rng.integers(0, 4, 3000, dtype=np.uint8)]))
add("far", np.concatenate([rng.integers(0, 256, 500, dtype=np.uint8), rng.integers(0, 256, 7600, dtype=np.uint8)] * 2))
add("tiny", np.frombuffer(b"abcabcabcabc", np.uint8))
add("max_chunk", np.tile(np.arange(251, dtype=np.uint8), 262)[:65535])
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lzf_liblzf.npz")
np.savez_compressed(out, **cases)
print(out, {k: v.size for k, v in cases.items()})
