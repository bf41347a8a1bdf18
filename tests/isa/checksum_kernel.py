"""Runs the compiled checksum kernels (csrc/checksum.hip: checksum_segments_kernel<ALGO> with 256-thread workgroups, then
checksum_combine_kernel<ALGO>) on the CPU through tests/isa/gfx950_emu.py (TEST INFRASTRUCTURE): what s3s_checksum_ranges and
the map-side / reduce-side per-partition Adler32 / CRC32 launch.  The data buffer ends with its last byte, the partial and
output arrays have exactly their sizes; tables and segment bookkeeping are rebuilt here the way the host code builds them
(checksum_tables_build, the seg_start prefix of codec_api.hip)."""
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gfx950_emu as emu  # noqa: E402
import lz4_kernel as lk  # noqa: E402

SEG = 16384  # kChecksumSegBytes
_PROG = {}


def _mul(a, b, poly=0xEDB88320):
    p = 0
    for i in range(32):
        if a & (0x80000000 >> i):
            p ^= b
        b = (b >> 1) ^ (poly if b & 1 else 0)
    return p


def tables(poly=0xEDB88320):
    """struct Tables of checksum.hip as checksum_tables_build fills it (one set per polynomial: CRC-32 / CRC-32C)"""
    sl = np.zeros((4, 256), np.uint32)
    for i in range(256):
        c = i
        for _ in range(8):
            c = (poly ^ (c >> 1)) if c & 1 else c >> 1
        sl[0, i] = c
    for s in range(1, 4):
        for i in range(256):
            sl[s, i] = (int(sl[s - 1, i]) >> 8) ^ int(sl[0, int(sl[s - 1, i]) & 0xFF])
    x2n = [0x40000000]
    for _ in range(31):
        x2n.append(_mul(x2n[-1], x2n[-1], poly))
    pw = [0x80000000]
    for _ in range(255):
        pw.append(_mul(pw[-1], x2n[9], poly))
    return np.concatenate([sl.reshape(-1), np.array(pw, np.uint32), np.array(x2n, np.uint32),
                           np.array([poly, 0, 0, 0], np.uint32)]).astype(np.uint32)


def _program(needle):
    if "text" not in _PROG:
        _PROG["text"] = lk.compile_asm("checksum.hip")
        _PROG["objs"] = {k: v for k, v in emu.parse_objects(_PROG["text"]).items() if k.startswith("_ZN3s3s")}
    if needle not in _PROG:
        entry = lk.find_kernel(_PROG["text"], needle)
        _PROG[needle] = (emu.Program(_PROG["text"], entry), entry)
    return _PROG[needle]


def checksum_ranges(algo, data: bytes, offsets, data_len=None, fold_group=0):
    """algo 1 = Adler32, 2 = CRC32, 3 = CRC32C (the CRC32 kernels on the other polynomial's tables, as
    launch_checksum_with_tables picks them); offsets: n + 1 ascending positions into data.  -> list of n checksums.
    fold_group = G > 0: the two-level form for ranges of very many segments (checksum_fold_kernel over groups of G segments,
    then the combine kernel over groups) — the product folds 256 segments from 2 048 per range on; a small G exercises the
    same code on ranges the interpreter can afford."""
    offsets = np.asarray(offsets, np.int64)
    n = len(offsets) - 1
    seg_start = np.zeros(n + 1, np.int32)
    for p in range(n):
        seg_start[p + 1] = seg_start[p] + (int(offsets[p + 1] - offsets[p]) + SEG - 1) // SEG
    total = int(seg_start[n])
    mem = emu.Memory()
    a_data = mem.map(np.frombuffer(bytearray(data) or bytearray(1), dtype=np.uint8), "data", writable=False)
    a_off = mem.map(offsets, "offsets", writable=False)
    a_seg = mem.map(seg_start, "seg_start", writable=False)
    a_tab = mem.map(tables(0x82F63B78 if algo == 3 else 0xEDB88320), "tables", writable=False)
    algo = 2 if algo == 3 else algo
    partial = np.zeros(max(4 * total, 4), np.uint32)[: 4 * total] if total else np.zeros(0, np.uint32)
    a_par = mem.map(partial if total else np.zeros(1, np.uint32), "partial")
    out = np.full(n, -1, np.int64)
    a_out = mem.map(out, "out")
    dl = len(data) if data_len is None else data_len
    tag = "checksum_segments_kernelILi%dE" % algo
    if total:
        prog, entry = _program(tag)
        emu.launch(prog, entry, mem, struct.pack("<QQiiQQQq", a_data, a_off, n, 0, a_seg, a_tab, a_par, dl), total, 0,
                   block_x=256, objects=_PROG["objs"])
    unit, groups, a_in = SEG, 0, a_par
    if fold_group:
        groups = max(1, max((int(seg_start[p + 1] - seg_start[p]) + fold_group - 1) // fold_group for p in range(n)))
        partial2 = np.full(4 * n * groups, 0xDEADBEEF, np.uint32)  # (never pre-zeroed by the product either)
        a_par2 = mem.map(partial2, "partial2")
        prog, entry = _program("checksum_fold_kernelILi%dE" % algo)
        emu.launch(prog, entry, mem, struct.pack("<QiiQQQQii", a_off, n, 0, a_seg, a_tab, a_par, a_par2, fold_group, groups),
                   groups * n, 0, objects=_PROG["objs"])
        unit, a_in = fold_group * SEG, a_par2
    prog, entry = _program("checksum_combine_kernelILi%dE" % algo)
    emu.launch(prog, entry, mem, struct.pack("<QiiQQQQqii", a_off, n, 0, a_seg, a_tab, a_in, a_out, unit, groups, 0), n, 0,
               objects=_PROG["objs"])
    return [int(x) for x in out]
