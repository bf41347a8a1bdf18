"""Runs the compiled `zstd_partitions_kernel` (Zstandard decode of the reduce side, zstd_decompress.hip +
zstd_decode_core.h) on the CPU through tests/isa/gfx950_emu.py (TEST INFRASTRUCTURE).  Both passes of the product: pass 1
(sizes, literal scratch need) and pass 2 (decode); source, destination and scratch buffers have exactly their declared
sizes, so any access outside them faults in the interpreter."""
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gfx950_emu as emu  # noqa: E402
import lz4_kernel as lk  # noqa: E402

_PROG = None
FLAGS = tuple(__import__("os").environ.get("ZK_FLAGS", "").split())  # extra hipcc flags (env ZK_FLAGS for tools) (build switches kept for measurements, e.g. ("-DZS_X_SOMETHING",)); set before program()


def program():
    global _PROG
    if _PROG is None:
        text = lk.compile_asm("zstd_decompress.hip", FLAGS)
        entry = lk.find_kernel(text, "zstd_partitions_kernel")
        lds = 0
        for line in text.splitlines():
            if ".amdhsa_group_segment_fixed_size" in line:
                lds = max(lds, int(line.split()[-1]))
        # (device functions hipcc keeps out of line: the frame checksum's hash - a rare path, ZS_RARE - and the lean sequence loop seq_fast)
        import re as _re
        callees = [m.group(1) for m in _re.finditer(r'^\s*\.type\s+(_ZN8s3s_zstd\w+),@function', text, _re.M) if m.group(1) != entry]
        _PROG = (emu.Program(text, entry, callees=callees), entry, text, lds)
    return _PROG


def decode_partitions(parts, profile=None):
    """parts: list of (compressed partition bytes = concatenated frames, decoded size).  Returns (list of decoded bytes or
    None, list of rc, waves of pass 2)."""
    prog, entry, text, lds = program()
    mem = emu.Memory()
    comp = np.frombuffer(b"".join(p for p, _ in parts) or b"\0", dtype=np.uint8).copy()
    a_comp = mem.map(comp, "comp", writable=False)
    total = sum(n for _, n in parts)
    dst = np.zeros(max(total, 1), dtype=np.uint8)
    a_dst = mem.map(dst, "dst")
    n = len(parts)
    zres = np.zeros(n * 24, dtype=np.uint8)
    a_res = mem.map(zres, "res")
    objs = dict(emu.parse_objects(text))  # constant tables of the decoder (predefined FSE distributions, code tables)

    def zparts(lit_offs, live=None, caps=None, lit_strides=None):
        b = bytearray()
        co = do = 0
        for k, (p, sz) in enumerate(parts):
            on = live is None or live[k]
            b += struct.pack("<QqQqqq", a_comp + co, len(p) if on else 0, a_dst + do, sz if caps is None else caps[k], lit_offs[k],
                             lit_strides[k] if lit_strides else 0)
            co += len(p)
            do += sz
        return np.frombuffer(bytes(b), dtype=np.uint8).copy()

    # pass 1: sizes
    zp = zparts([0] * n)
    a_parts = mem.map(zp, "parts1", writable=False)
    kernarg = struct.pack("<QiiQQ", a_parts, n, 0, 0, a_res)
    emu.launch(prog, entry, mem, kernarg, (n + 1) // 2, lds, objects=objs, block_x=192, cooperative=True)
    res1 = [struct.unpack("<qqii", bytes(zres[24 * k:24 * k + 24])) for k in range(n)]
    # the host's verdicts between the passes (zstd_decompress_ranges): a partition whose size pass failed, or whose decoded
    # size exceeds its destination, takes no part in pass 2 (size 0); the others decode with cap = what the size pass found
    # and a literal scratch of what it asked for
    lit_offs, lit_strides, lit_total = [], [], 0
    live = [rc == 0 and tot <= parts[k][1] for k, (tot, need, rc, _) in enumerate(res1)]
    for k, (tot, need, rc, _) in enumerate(res1):
        lit_offs.append(lit_total)
        lit_strides.append((need + 64 + 15) & ~15)  # two literal buffers per partition: block k of the literal wavefront -> k & 1
        if live[k]:
            lit_total += 2 * lit_strides[k]
    sizes_ok = [rc == 0 and tot == parts[k][1] for k, (tot, need, rc, _) in enumerate(res1)]
    lit = np.zeros(lit_total + 64, dtype=np.uint8)
    a_lit = mem.map(lit, "lit")
    zp2 = zparts(lit_offs, live, [res1[k][0] if live[k] else 0 for k in range(n)], lit_strides)
    a_parts2 = mem.map(zp2, "parts2", writable=False)
    kernarg = struct.pack("<QiiQQ", a_parts2, n, 1, a_lit, a_res)
    waves = emu.launch(prog, entry, mem, kernarg, (n + 1) // 2, lds, profile=profile, objects=objs, block_x=192, cooperative=True)
    out, rcs, do = [], [], 0
    for k, (p, sz) in enumerate(parts):
        tot, need, rc, _ = struct.unpack("<qqii", bytes(zres[24 * k:24 * k + 24]))
        if not live[k]:  # (pass 2 did not run for it: the size pass's verdict stands)
            tot, need, rc = res1[k][0], res1[k][1], (res1[k][2] or -2)
        rcs.append(rc if sizes_ok[k] or rc else -999)
        out.append(bytes(dst[do:do + sz]) if rc == 0 and tot == sz else None)
        do += sz
    return out, rcs, waves
