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


FLAGS = ()  # extra hipcc flags of an experiment build (set before the first launch; e.g. ("-DS3S_X_SOMETHING",))


def use_flags(flags):
    """switch to another build of lz4_decompress.hip (drops the cached programs)"""
    global FLAGS
    FLAGS = tuple(flags)
    for k in [k for k in _PROG if k != "sn"]:
        del _PROG[k]


def _program(needle):
    if "text" not in _PROG:
        _PROG["text"] = lk.compile_asm("lz4_decompress.hip", flags=FLAGS)
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


def discover_snappy(stream: bytes, part_offsets, chunk_format=0):
    """chunk_format 1: compress-lzf chunk chains (LZFInputStream) through the same two kernels (round 4).
    SnappyOutputStream images of several partitions in one range (csrc/snappy_decompress.hip: snappy_count_kernel ->
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
    run("snappy_count_kernel", struct.pack("<QQiiQQi", a_comp, a_off, n_parts, 0, a_cnt, a_st, chunk_format), (n_parts + 63) // 64)
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
    run("snappy_emit_kernel", struct.pack("<QQiiQQQQi", a_comp, a_off, n_parts, 0, a_base, a_fr, a_or, a_st, chunk_format), (n_parts + 63) // 64)
    _launch("scan_u32_kernel", mem, struct.pack("<QqQ", a_or, n_frames, a_fo), 1)
    recs = [struct.unpack_from("<qiiIi", frames, 24 * k) for k in range(n_frames)]
    return int(status[0]), recs, [int(x) for x in fout]


def decode_ranges_batch(ranges, capacities, skip=()):
    """The batched reduce-side call for LZ4 (s3s_decompress_ranges_batch_device, csrc/decode_api.hip) through the compiled
    kernels: tile_speculate_batch -> tile_resolve_batch | host: frame counts -> tile_emit_batch -> frames_finish_batch ->
    ONE batch_decode_kernel launch over the frames of every range (comp = dst = nullptr: the records carry absolute
    addresses) -> lz4_verify_frames_kernel.  ranges: list of LZ4Block streams (bytes); capacities: destination bytes per range
    (the destination buffer of a range has exactly that size); skip: ranges the host marks as failed (checksum) beforehand.
    -> list of (status, decoded bytes or None) per range, and the decode launch's own status word"""
    import decode_kernel as dk

    n_ranges = len(ranges)
    mem = emu.Memory()
    desc, bufs = [], []
    tile_range = []
    status = np.zeros(n_ranges + 1, np.int32)
    a_status = mem.map(status, "status")
    result = np.full(2 * n_ranges, -7, np.int64)
    a_result = mem.map(result, "result")
    for r, stream in enumerate(ranges):
        n = len(stream)
        nt = (n + TILE - 1) // TILE
        b = {"comp": np.frombuffer(bytearray(stream) or bytearray(1), dtype=np.uint8),
             "se": np.full(max(nt, 1), -7, np.int64)[:nt], "sx": np.full(max(nt, 1), -7, np.int64)[:nt],
             "te": np.full(max(nt, 1), -7, np.int64)[:nt], "fb": np.full(nt + 1, -7, np.int64),
             "sc": np.full(max(nt, 1), -7, np.int32)[:nt], "dst": np.full(max(capacities[r], 1), 0xA5, np.uint8)[:capacities[r]]}
        a = {k: mem.map(v if v.size else np.zeros(1, v.dtype), "%s%d" % (k, r), writable=(k != "comp")) for k, v in b.items()}
        d = dict(comp=a["comp"], comp_len=n, n_tiles=nt, tile0=len(tile_range), se=a["se"], sx=a["sx"], te=a["te"], fb=a["fb"],
                 sc=a["sc"], status=a_status + 4 * r, result=a_result + 16 * r, frames=0, forig=0, fout=0, oabs=0, n_frames=0,
                 dst_base=a["dst"], cap=capacities[r], skip=1 if r in skip else 0)
        tile_range += [r] * nt
        desc.append(d)
        bufs.append((b, a))

    def pack():
        out = bytearray()
        for d in desc:
            out += struct.pack("<QqiiQQQQQQQQQQQqqqii", d["comp"], d["comp_len"], d["n_tiles"], d["tile0"], d["se"], d["sx"], d["te"],
                               d["fb"], d["sc"], d["status"], d["result"], d["frames"], d["forig"], d["fout"], d["oabs"],
                               d["n_frames"], d["dst_base"], d["cap"], d["skip"], 0)
        return np.frombuffer(out, dtype=np.uint8).copy()

    total_tiles = len(tile_range)
    a_tr = mem.map(np.array(tile_range or [0], np.int32), "tile_range", writable=False)
    d_ranges = pack()
    a_rg = mem.map(d_ranges, "ranges")
    if total_tiles:
        _launch("tile_speculate_batch_kernel", mem, struct.pack("<QQi", a_rg, a_tr, total_tiles), total_tiles)
    _launch("tile_resolve_batch_kernel", mem, struct.pack("<Qi", a_rg, n_ranges), n_ranges)
    # host: frame counts; a range whose chain is broken (or that was marked) takes no part in what follows
    total_frames = 0
    first = []
    for r, d in enumerate(desc):
        nf = int(result[2 * r]) if d["n_tiles"] > 0 else 0
        if status[r] != 0 or d["skip"]:
            nf, d["skip"] = 0, 1
        d["n_frames"] = nf
        first.append(total_frames)
        total_frames += nf
    frames = np.zeros(max(total_frames, 1) * 24, np.uint8)[: total_frames * 24]
    out_abs = np.full(max(total_frames, 1), -7, np.int64)[:total_frames]
    a_frames = mem.map(frames if total_frames else np.zeros(1, np.uint8), "frames")
    a_oabs = mem.map(out_abs if total_frames else np.zeros(1, np.int64), "out_abs")
    for r, d in enumerate(desc):
        nf = d["n_frames"]
        fo = np.zeros(max(nf, 1), np.uint32)[:nf]
        fout = np.full(nf + 1, -7, np.int64)
        bufs[r][0]["forig"], bufs[r][0]["fout"] = fo, fout
        d["forig"] = mem.map(fo if nf else np.zeros(1, np.uint32), "frame_orig%d" % r)
        d["fout"] = mem.map(fout, "frame_out%d" % r)
        d["frames"] = a_frames + 24 * first[r]
        d["oabs"] = a_oabs + 8 * first[r]
    d_ranges[:] = pack()
    if total_tiles:
        _launch("tile_emit_batch_kernel", mem, struct.pack("<QQi", a_rg, a_tr, total_tiles), (total_tiles + 63) // 64)
    _launch("frames_finish_batch_kernel", mem, struct.pack("<Qi", a_rg, n_ranges), n_ranges)
    if total_frames:
        prog, entry, text, lds = dk.program(0)
        objs = {k: v for k, v in emu.parse_objects(text).items() if k.startswith("_ZN3s3s")}
        a_dec = a_status + 4 * n_ranges
        emu.launch(prog, entry, mem, struct.pack("<QQiiQQQ", 0, a_frames, total_frames, 0, a_oabs, 0, a_dec), total_frames, lds,
                   objects=objs)
        ventry = lk.find_kernel(text, "lz4_verify_frames_kernel")
        emu.launch(emu.Program(text, ventry), ventry, mem, struct.pack("<QiiQQQ", a_frames, total_frames, 0, a_oabs, 0, a_dec),
                   (total_frames + 15) // 16, lds, objects=objs)
    out = []
    for r, d in enumerate(desc):
        st = int(status[r])
        if st == 0 and d["skip"]:
            st = -4  # (the host's own mark: a checksum mismatch found before the decode)
        total = int(result[2 * r + 1]) if d["n_frames"] > 0 else 0
        if st == 0 and total > d["cap"]:
            st = -2
        out.append((st, bufs[r][0]["dst"][:total].tobytes() if st == 0 else None))
    return out, int(status[n_ranges])
