"""Runs the compiled `lz4_compress_l2_kernel` (hipcc -S output) on the CPU through tests/isa/gfx950_emu.py.

TEST INFRASTRUCTURE ONLY (uses the oracle as checker).  `compress_chunks(chunks)` returns, for every chunk, the
LZ4 block payload the compiled kernel writes (or None when it stores the chunk RAW) plus the wave statistics.
"""
import hashlib
import os
import struct
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "spark-s3-shuffle_amd", "csrc")
sys.path.insert(0, HERE)
import gfx950_emu as emu  # noqa: E402

K_SLOT_HEADER, K_SLOT_BYTES, K_FRAME_HEADER = 32, 32 + 32768, 21
RAW_FLAG = 0x80000000


def compile_asm(src="lz4_compress.hip", flags=(), cache_dir=None):
    """hipcc -S of one kernel source for gfx950 -> assembly text (cached on the source's hash + flags)."""
    cache_dir = cache_dir or os.path.join(HERE, "_asm")
    os.makedirs(cache_dir, exist_ok=True)
    path = os.path.join(CSRC, src)
    h = hashlib.sha1()
    for p in (path, os.path.join(CSRC, "s3s_internal.h"), os.path.join(CSRC, "s3s_ctx.h")):
        h.update(open(p, "rb").read())
    for extra in sorted(os.listdir(CSRC)):  # every header and include a kernel source can pull in (zstd_decode_core.h was missing
        if extra.endswith((".inc", ".h")):  # until round 4: a header-only change ran the interpreter tests on a stale assembly)
            h.update(open(os.path.join(CSRC, extra), "rb").read())
    h.update(" ".join(flags).encode())
    out = os.path.join(cache_dir, "%s.%s.s" % (src, h.hexdigest()[:12]))
    if not os.path.exists(out):
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-w",
               "-I" + os.path.join(ROOT, "include"), *flags, path, "-o", out + ".tmp"]
        subprocess.run(cmd, check=True, cwd=CSRC)
        os.replace(out + ".tmp", out)
    return open(out).read()


def find_kernel(text, needle):
    for line in text.splitlines():
        if line.endswith(":") and needle in line and not line.startswith((".", "\t", " ")):
            return line[:-1]
        s = line.split(";")[0].strip()
        if s.endswith(":") and needle in s and not s.startswith("."):
            return s[:-1]
    raise KeyError(needle)


_PROGS = {}


def program(windows=True, flags=()):
    key = (windows, tuple(flags))
    if key not in _PROGS:
        text = compile_asm(flags=flags)
        entry = find_kernel(text, "lz4_compress_l2_kernelILb%dE" % (1 if windows else 0))
        _PROGS[key] = (emu.Program(text, entry), entry)
    return _PROGS[key]


def compress_chunks(chunks, windows=True, flags=(), profile=None, lds_order=None, tail_guard=0, hooks=None, block=32768):
    """chunks: list of uint8 arrays (<= block bytes each; block <= 65536: the slot stride the host passes is 32 + block).  The source buffer ends exactly at the last chunk's
    last byte (+ tail_guard), so any read past a chunk that ends the allocation faults."""
    prog, entry = program(windows, flags)
    mem = emu.Memory()
    src = np.concatenate([np.asarray(c, dtype=np.uint8) for c in chunks] + [np.zeros(tail_guard, np.uint8)])
    n = len(chunks)
    items = bytearray()
    off = 0
    for k, c in enumerate(chunks):
        items += struct.pack("<qiiii", off, len(c), 0 | (5 << 8), k, 0)
        off += len(c)
    K_SLOT_BYTES = K_SLOT_HEADER + ((block + 15) & ~15)
    slots = np.zeros(n * K_SLOT_BYTES, dtype=np.uint8)
    sizes = np.zeros(n, dtype=np.uint32)
    checks = np.arange(n, dtype=np.uint32) * np.uint32(0x01010101)
    a_src = mem.map(src if src.size else np.zeros(1, np.uint8), "src", writable=False)
    a_items = mem.map(np.frombuffer(items, dtype=np.uint8), "items", writable=False)
    a_check = mem.map(checks, "item_check", writable=False)
    a_slots = mem.map(slots, "slots")
    a_sizes = mem.map(sizes, "item_size")
    # The kernel is a persistent grid (every wavefront takes blocks from a counter until none is left); the interpreter
    # runs wavefronts one after the other, so each chunk gets its own one-item launch and its own wavefront's statistics
    work = np.zeros(16, dtype=np.uint32)
    a_work = mem.map(work, "work")
    waves = []
    for k in range(n):
        work[:] = 0
        kernarg = struct.pack("<QQiiQQQQ", a_src, a_items + 24 * k, 1, K_SLOT_BYTES, a_check + 4 * k, a_slots, a_sizes + 4 * k, a_work)
        waves += emu.launch(prog, entry, mem, kernarg, 1, 16384, profile=profile, lds_order=lds_order, hooks=hooks)
        assert work[0] == 2, "the wavefront takes its block and then finds the counter exhausted"
    out = []
    for k in range(n):
        sz = int(sizes[k])
        slot = slots[k * K_SLOT_BYTES:(k + 1) * K_SLOT_BYTES]
        hdr = slot[K_SLOT_HEADER - K_FRAME_HEADER:K_SLOT_HEADER]
        plen = (sz & 0x7FFFFFFF) - K_FRAME_HEADER
        payload = None if sz & RAW_FLAG else slot[K_SLOT_HEADER:K_SLOT_HEADER + plen].copy()
        out.append((payload, bytes(hdr), waves[k]))
    return out
