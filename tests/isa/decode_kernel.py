"""Runs the compiled `batch_decode_kernel` (LZ4 / Snappy block decoder of the reduce side) on the CPU through
tests/isa/gfx950_emu.py (TEST INFRASTRUCTURE).  The compressed payloads sit in a buffer that ends with the last
payload's last byte and the destination has exactly the declared decoded size, so any access outside either is a
fault of the interpreter's memory — a byte-exact version of the guard-byte tests of tests/test_gpu_hardening.py."""
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gfx950_emu as emu  # noqa: E402
import lz4_kernel as lk  # noqa: E402

_PROGS = {}


def program(fmt, flags=(), kernel="batch"):
    """kernel: "batch" = batch_decode_kernel<fmt> (the default decoder), "ring" = the ring decoders (decode variant 3, and
    what LZ4 frames above 32 KiB fall back to): lz4_decompress_valu_kernel / snappy_decompress_valu_kernel."""
    key = (fmt, tuple(flags), kernel)
    if key not in _PROGS:
        if kernel == "batch":
            text = lk.compile_asm("lz4_decode_batch.hip", flags=flags)
            entry = lk.find_kernel(text, "batch_decode_kernelILi%dE" % fmt)
        else:
            text = lk.compile_asm("lz4_decompress.hip" if fmt == 0 else "snappy_decompress.hip", flags=flags)
            entry = lk.find_kernel(text, "lz4_decompress_valu_kernel" if fmt == 0 else "snappy_decompress_valu_kernel")
        lds = 0
        for line in text.splitlines():  # the kernel's LDS size from its descriptor block
            if ".amdhsa_group_segment_fixed_size" in line:
                lds = max(lds, int(line.split()[-1]))
        _PROGS[key] = (emu.Program(text, entry), entry, text, lds or 6144)
    return _PROGS[key]


def decode_blocks(blocks, fmt=0, methods=None, profile=None, flags=(), grid=None, kernel="batch", checks=None):
    """blocks: list of (payload bytes, decoded length).  fmt 0 = LZ4 block payloads, 1 = raw Snappy blocks.
    Returns (list of decoded bytes, status word, waves).  flags: extra -D macros of an experiment build.  checks: the frames' LZ4Block check fields (the LZ4 ring kernel verifies
    xxHash32 itself; the batch decoder leaves that to lz4_verify_frames_kernel)."""
    prog, entry, text, lds = program(fmt, flags, kernel)
    mem = emu.Memory()
    comp = b"".join(p for p, _ in blocks)
    frames = bytearray()
    outs = []
    co = oo = 0
    for k, (p, olen) in enumerate(blocks):
        method = (methods[k] if methods else (0x20 if fmt == 0 else 1))
        frames += struct.pack("<qiiIi", co, len(p), olen, checks[k] if checks else 0, method)
        outs.append(oo)
        co += len(p)
        oo += olen
    dst = np.zeros(max(oo, 1), dtype=np.uint8)
    status = np.zeros(4, dtype=np.int32)
    if kernel == "ring":
        # the ring decoders read the stream in ALIGNED dwords, clamped to the last dword that holds a stream byte: the word
        # around the buffer's last byte is touched as a whole (never a byte outside it — an allocation is a multiple of 4).
        # The filler is not zero, so a decoder that USED those bytes would decode wrongly.
        comp = comp + b"\xEE" * (-len(comp) % 4)
    a_comp = mem.map(np.frombuffer(bytearray(comp) or bytearray(1), dtype=np.uint8), "comp", writable=False)
    a_frames = mem.map(np.frombuffer(frames, dtype=np.uint8), "frames", writable=False)
    a_fout = mem.map(np.array(outs + [oo], dtype=np.int64), "frame_out", writable=False)
    a_dst = mem.map(dst, "dst")
    a_status = mem.map(status, "status")
    kernarg = struct.pack("<QQiiQQQ", a_comp, a_frames, len(blocks), 0, a_fout, a_dst, a_status)
    objs = emu.parse_objects(text)
    waves = emu.launch(prog, entry, mem, kernarg, grid or len(blocks), lds, profile=profile,
                       objects={k: v for k, v in objs.items() if k.startswith("_ZN3s3s")})
    res = [bytes(dst[outs[k]:outs[k] + blocks[k][1]]) for k in range(len(blocks))]
    return res, int(status[0]), waves


def decode_range(comp: bytes, recs, outs, verify=True, fmt=0):
    """The decode launches of one fetched range: `recs` / `outs` = the frame records and output offsets the compiled frame
    discovery produced (tests/isa/discover_kernel.py), batch_decode_kernel<LZ4> over them, then lz4_verify_frames_kernel
    (xxHash32 of every decoded block against the frame's check field).  Buffers of exactly their sizes.
    -> (status, decoded bytes)"""
    prog, entry, text, lds = program(fmt)
    verify = verify and fmt == 0  # (Snappy chunks carry no check field: the partition checksum is their guard)
    n = len(recs)
    total = outs[-1]
    mem = emu.Memory()
    frames = bytearray()
    for r in recs:
        frames += struct.pack("<qiiIi", *r)
    dst = np.zeros(max(total, 1), dtype=np.uint8)[:total]
    status = np.zeros(4, dtype=np.int32)
    a_comp = mem.map(np.frombuffer(bytearray(comp), dtype=np.uint8), "comp", writable=False)
    a_frames = mem.map(np.frombuffer(frames or bytearray(24), dtype=np.uint8), "frames", writable=False)
    a_fout = mem.map(np.array(outs, dtype=np.int64), "frame_out", writable=False)
    a_dst = mem.map(dst if total else np.zeros(1, np.uint8), "dst")
    a_status = mem.map(status, "status")
    objs = {k: v for k, v in emu.parse_objects(text).items() if k.startswith("_ZN3s3s")}
    if n:
        emu.launch(prog, entry, mem, struct.pack("<QQiiQQQ", a_comp, a_frames, n, 0, a_fout, a_dst, a_status), n, lds, objects=objs)
        if verify:
            ventry = lk.find_kernel(text, "lz4_verify_frames_kernel")
            emu.launch(emu.Program(text, ventry), ventry, mem, struct.pack("<QiiQQQ", a_frames, n, 0, a_fout, a_dst, a_status),
                       (n + 15) // 16, lds, objects=objs)
    return int(status[0]), dst.tobytes()
