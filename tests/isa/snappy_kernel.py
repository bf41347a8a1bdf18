"""Runs the compiled `snappy_compress_kernel` on the CPU through tests/isa/gfx950_emu.py (TEST INFRASTRUCTURE)."""
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gfx950_emu as emu  # noqa: E402
import lz4_kernel as lk  # noqa: E402

_PROGS = {}


def program(windows=True, flags=()):
    key = (windows, tuple(flags))
    if key not in _PROGS:
        text = lk.compile_asm("snappy_compress.hip", flags=flags)
        entry = lk.find_kernel(text, "snappy_compress_kernelILb%dE" % (1 if windows else 0))
        _PROGS[key] = (emu.Program(text, entry), entry, text)
    return _PROGS[key]


def compress_chunks(chunks, windows=True, profile=None, hooks=None, lds_order=None, flags=()):
    prog, entry, text = program(windows, flags)
    mem = emu.Memory()
    src = np.concatenate([np.asarray(c, dtype=np.uint8) for c in chunks])
    n = len(chunks)
    stride = 32 + 32768 + 32768 // 6 + 64
    stride = (stride + 15) & ~15
    items = bytearray()
    off = 0
    for k, c in enumerate(chunks):
        items += struct.pack("<qiiii", off, len(c), 3, k, 0)
        off += len(c)
    slots = np.zeros(n * stride, dtype=np.uint8)
    sizes = np.zeros(n, dtype=np.uint32)
    a_src = mem.map(src, "src", writable=False)
    a_items = mem.map(np.frombuffer(items, dtype=np.uint8), "items", writable=False)
    a_slots = mem.map(slots, "slots")
    a_sizes = mem.map(sizes, "item_size")
    kernarg = struct.pack("<QQiiQqQ", a_src, a_items, n, 0, a_slots, stride, a_sizes)
    objs = {k: v for k, v in emu.parse_objects(text).items() if "g_sn_sched" in k}
    waves = emu.launch(prog, entry, mem, kernarg, n, 32768, profile=profile, hooks=hooks, lds_order=lds_order, objects=objs)
    out = []
    for k in range(n):
        sz = int(sizes[k])
        slot = slots[k * stride:(k + 1) * stride]
        out.append((slot, sz, waves[k]))
    return out
