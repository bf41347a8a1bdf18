"""Runs the compiled LZ4Block frame discovery of the reduce side (csrc/lz4_decompress.hip: tile_speculate_kernel ->
tile_resolve_kernel -> scan_u32_kernel -> tile_emit_kernel -> scan_u32_kernel, the launches of launch_lz4_discover /
launch_lz4_emit_frames in their order) on the CPU through tests/isa/gfx950_emu.py (TEST INFRASTRUCTURE).

Every buffer has exactly the size the host code gives it a right to — the compressed range ends with its last byte, the
speculation / entry / base arrays have n_tiles (+1) entries, the frame records n_frames — so a header parse that looks
past the end of a truncated range, or an emit that writes one record too many, is a fault of the interpreter's memory."""
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gfx950_emu as emu  # noqa: E402
import lz4_kernel as lk  # noqa: E402

TILE = 65536
_PROG = {}


def _program(needle):
    if "text" not in _PROG:
        _PROG["text"] = lk.compile_asm("lz4_decompress.hip")
        _PROG["objs"] = {k: v for k, v in emu.parse_objects(_PROG["text"]).items() if k.startswith("_ZN3s3s")}
    if needle not in _PROG:
        entry = lk.find_kernel(_PROG["text"], needle)
        _PROG[needle] = (emu.Program(_PROG["text"], entry), entry)
    return _PROG[needle]


def _launch(needle, mem, kernarg, grid):
    prog, entry = _program(needle)
    emu.launch(prog, entry, mem, kernarg, grid, 0, objects=_PROG["objs"])


def discover(stream: bytes):
    """-> (status, frames) with frames = [(payload offset, payload bytes, decoded bytes, check, method)], and the
    exclusive scan of the decoded sizes (n_frames + 1 entries) the decode launch uses as output offsets."""
    n = len(stream)
    n_tiles = (n + TILE - 1) // TILE
    if n_tiles == 0:
        return 0, [], [0]
    mem = emu.Memory()
    a_comp = mem.map(np.frombuffer(bytearray(stream), dtype=np.uint8), "comp", writable=False)
    spec_entry = np.full(n_tiles, -7, np.int64)
    spec_exit = np.full(n_tiles, -7, np.int64)
    spec_count = np.full(n_tiles, -7, np.int32)
    true_entry = np.full(n_tiles, -7, np.int64)
    frame_base = np.full(n_tiles + 1, -7, np.int64)
    status = np.zeros(1, np.int32)
    a_se, a_sx, a_sc = mem.map(spec_entry, "spec_entry"), mem.map(spec_exit, "spec_exit"), mem.map(spec_count, "spec_count")
    a_te, a_fb, a_st = mem.map(true_entry, "true_entry"), mem.map(frame_base, "frame_base"), mem.map(status, "status")
    _launch("tile_speculate_kernel", mem, struct.pack("<QqiiQQQ", a_comp, n, n_tiles, 0, a_se, a_sx, a_sc), n_tiles)
    _launch("tile_resolve_kernel", mem, struct.pack("<QqiiQQQQQ", a_comp, n, n_tiles, 0, a_se, a_sx, a_sc, a_te, a_st), 1)
    if int(status[0]) != 0:
        return int(status[0]), [], [0]
    _launch("scan_u32_kernel", mem, struct.pack("<QqQ", a_sc, n_tiles, a_fb), 1)
    n_frames = int(frame_base[n_tiles])
    assert 0 <= n_frames <= n // 21 + 1, n_frames
    frames = np.zeros(max(n_frames, 1) * 24, np.uint8)[: n_frames * 24]
    frame_orig = np.zeros(n_frames, np.uint32)
    frame_out = np.full(n_frames + 1, -7, np.int64)
    a_fr = mem.map(frames if n_frames else np.zeros(1, np.uint8), "frames")
    a_fo = mem.map(frame_orig if n_frames else np.zeros(1, np.uint32), "frame_orig")
    a_fout = mem.map(frame_out, "frame_out")
    _launch("tile_emit_kernel", mem, struct.pack("<QqiiQQQQQ", a_comp, n, n_tiles, 0, a_te, a_fb, a_fr, a_fo, a_st),
            (n_tiles + 63) // 64)
    _launch("scan_u32_kernel", mem, struct.pack("<QqQ", a_fo, n_frames, a_fout), 1)
    recs = [struct.unpack_from("<qiiIi", frames, 24 * k) for k in range(n_frames)]
    return int(status[0]), recs, [int(x) for x in frame_out]


def reference_frames(stream: bytes):
    """what LZ4BlockInputStream.refill() accepts, frame by frame: None for a stream it refuses"""
    out, pos, n = [], 0, len(stream)
    while pos < n:
        if n - pos < 21 or stream[pos:pos + 8] != b"LZ4Block":
            return None
        token = stream[pos + 8]
        method, level = token & 0xF0, 10 + (token & 0x0F)
        clen, olen, check = struct.unpack_from("<iiI", stream, pos + 9)
        ok = (method in (0x10, 0x20) and olen >= 0 and clen >= 0 and olen <= (1 << level) and not (olen == 0 and clen != 0)
              and not (olen != 0 and clen == 0) and not (method == 0x10 and olen != clen) and not (olen == 0 and check != 0))
        if not ok or pos + 21 + clen > n:
            return None
        out.append((pos + 21, clen, olen, check, method))
        pos += 21 + clen
    return out


def discover_snappy(stream: bytes, part_offsets):
    """SnappyOutputStream images of several partitions in one range (csrc/snappy_decompress.hip: snappy_count_kernel ->
    scan_u32_kernel -> snappy_emit_kernel -> scan_u32_kernel, as decode_api.hip launches them).
    -> (status, frames [(payload offset, payload bytes, decoded bytes, 0, 1)], output offsets)"""
    offs = np.asarray(part_offsets, np.int64)
    n_parts = len(offs) - 1
    if "sn" not in _PROG:
        text = lk.compile_asm("snappy_decompress.hip")
        _PROG["sn"] = (text, {k: v for k, v in emu.parse_objects(text).items() if k.startswith("_ZN3s3s")})
    text, objs = _PROG["sn"]

    def run(needle, kernarg, grid):
        # these two kernels read blockDim.x: the implicit arguments follow the explicit ones (code object v5: block counts
        # x / y / z as u32, then group sizes x / y / z as u16, then the remainders)
        kernarg += b"\0" * (-len(kernarg) % 8) + struct.pack("<IIIHHHHHH", grid, 1, 1, 64, 1, 1, 0, 0, 0) + bytes(200)
        entry = lk.find_kernel(text, needle)
        emu.launch(emu.Program(text, entry), entry, mem, kernarg, grid, 0, objects=objs)

    mem = emu.Memory()
    a_comp = mem.map(np.frombuffer(bytearray(stream) or bytearray(1), dtype=np.uint8), "comp", writable=False)
    a_off = mem.map(offs, "part_off", writable=False)
    cnt = np.full(n_parts, 0xFFFFFFFF, np.uint32)
    base = np.full(n_parts + 1, -7, np.int64)
    status = np.zeros(1, np.int32)
    a_cnt, a_base, a_st = mem.map(cnt, "part_nframes"), mem.map(base, "frame_base"), mem.map(status, "status")
    run("snappy_count_kernel", struct.pack("<QQiiQQ", a_comp, a_off, n_parts, 0, a_cnt, a_st), (n_parts + 63) // 64)
    if int(status[0]) != 0:
        return int(status[0]), [], [0]  # (the host stops here: "Stream is corrupted (snappy chunk chain)")
    _launch("scan_u32_kernel", mem, struct.pack("<QqQ", a_cnt, n_parts, a_base), 1)
    n_frames = int(base[n_parts])
    frames = np.zeros(max(n_frames, 1) * 24, np.uint8)[: n_frames * 24]
    orig = np.zeros(n_frames, np.uint32)
    fout = np.full(n_frames + 1, -7, np.int64)
    a_fr = mem.map(frames if n_frames else np.zeros(1, np.uint8), "frames")
    a_or = mem.map(orig if n_frames else np.zeros(1, np.uint32), "frame_orig")
    a_fo = mem.map(fout, "frame_out")
    run("snappy_emit_kernel", struct.pack("<QQiiQQQQ", a_comp, a_off, n_parts, 0, a_base, a_fr, a_or, a_st), (n_parts + 63) // 64)
    _launch("scan_u32_kernel", mem, struct.pack("<QqQ", a_or, n_frames, a_fo), 1)
    recs = [struct.unpack_from("<qiiIi", frames, 24 * k) for k in range(n_frames)]
    return int(status[0]), recs, [int(x) for x in fout]
