"""One whole map-side call (LZ4 + checksum) through the compiled kernels on the CPU (TEST INFRASTRUCTURE): the launches of
compress_core in csrc/codec_api.hip in their order —

    xxh32_items_quad_kernel   (frame checks)          csrc/lz4_compress.hip
    lz4_compress_l2_kernel    (blocks + end frames)   csrc/lz4_compress.hip   (persistent grid: one wavefront draws every item)
    scan_items_kernel         (item offsets + index)  csrc/assemble.hip
    gather_items_kernel       (.data image)           csrc/assemble.hip       (256-thread workgroups)
    checksum_segments / checksum_combine              csrc/checksum.hip       (tests/isa/checksum_kernel.py)

with the item plan built the way the host code builds it.  Every buffer has exactly its size: the source ends with the last
partition's last byte, the destination has exactly the image's size (dst_capacity = that), the slot / item arrays exactly
n_chunks / n_items entries."""
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import checksum_kernel as ck  # noqa: E402
import gfx950_emu as emu  # noqa: E402
import lz4_kernel as lk  # noqa: E402

BLOCK = 32768
LEVEL = 5  # log2(32 KiB) - 10
SEED = 0x9747B28C
_P = {}


def _prog(src, needle):
    if src not in _P:
        text = lk.compile_asm(src)
        _P[src] = (text, {k: v for k, v in emu.parse_objects(text).items() if k.startswith("_ZN3s3s")})
    if (src, needle) not in _P:
        entry = lk.find_kernel(_P[src][0], needle)
        _P[(src, needle)] = (emu.Program(_P[src][0], entry), entry)
    return _P[(src, needle)] + (_P[src][1],)


def compress_map_output(parts, algo, dst_bytes, codec=1, block=None):
    """parts: list of bytes (one per partition, may be empty); dst_bytes: size of the destination (= dst_capacity);
    codec 1 = LZ4 (LZ4Block frames), 2 = Snappy (SnappyOutputStream chunks; snappy_compress_kernel, one workgroup per item).
    -> (status, image bytes, index list [n + 1], checksums list [n] or None)"""
    n = len(parts)
    snappy = codec == 2
    BLOCK = block or globals()['BLOCK']  # (LZ4: up to 65536, spark.io.compression.lz4.blockSize; token level = ceil(log2) - 10)
    LEVEL = max(0, (BLOCK - 1).bit_length() - 10)
    # a slot holds one chunk's codec output (codec_api.hip): kSlotBytes for LZ4, kSlotHeader + MaxCompressedLength rounded for Snappy
    stride = 32 + ((32 + BLOCK + BLOCK // 6 + 15) & ~15) if snappy else 32 + ((BLOCK + 15) & ~15)
    src = np.frombuffer(b"".join(parts), dtype=np.uint8)
    items = bytearray()
    part_first = []
    off = ch = 0
    for p, b in enumerate(parts):
        part_first.append(len(items) // 24)
        if b and snappy:
            items += struct.pack("<qiiii", 0, 0, 2, -1, p)  # kItemSnappyHeader
        for pos in range(0, len(b), BLOCK):
            items += struct.pack("<qiiii", off + pos, min(BLOCK, len(b) - pos), 3 if snappy else 0 | (LEVEL << 8), ch, p)
            ch += 1
        if b and not snappy:
            items += struct.pack("<qiiii", 0, 0, 1 | (LEVEL << 8), -1, p)  # kItemLz4End
        off += len(b)
    n_items = len(items) // 24
    part_first.append(n_items)
    mem = emu.Memory()
    a_src = mem.map(src.copy() if src.size else np.zeros(1, np.uint8), "src", writable=False)
    a_items = mem.map(np.frombuffer(items or bytearray(24), dtype=np.uint8), "items", writable=False)
    check = np.zeros(max(n_items, 1), np.uint32)
    size = np.zeros(max(n_items, 1), np.uint32)
    item_off = np.full(n_items + 1, -7, np.int64)
    index = np.full(n + 1, -7, np.int64)
    slots = np.zeros(max(ch, 1) * stride, np.uint8)
    work = np.zeros(1, np.uint32)
    status = np.zeros(1, np.int32)
    dst = np.full(max(dst_bytes, 1), 0xA5, np.uint8)[:dst_bytes]
    a_check, a_size, a_off, a_index = mem.map(check, "item_check"), mem.map(size, "item_size"), mem.map(item_off, "item_off"), mem.map(index, "index")
    a_slots, a_work, a_status = mem.map(slots, "slots"), mem.map(work, "work"), mem.map(status, "status")
    a_pf = mem.map(np.array(part_first, np.int32), "part_first", writable=False)
    a_dst = mem.map(dst if dst_bytes else np.zeros(1, np.uint8), "dst")
    if n_items and snappy:
        prog, entry, objs = _prog("snappy_compress.hip", "snappy_compress_kernelILb1E")
        objs = {k: v for k, v in emu.parse_objects(_P["snappy_compress.hip"][0]).items() if "g_sn_sched" in k or k.startswith("_ZN3s3s")}
        emu.launch(prog, entry, mem, struct.pack("<QQiiQqQ", a_src, a_items, n_items, 0, a_slots, stride, a_size), n_items, 32768,
                   objects=objs)
    elif n_items:
        # the kernel launch_lz4_compress launches by default: four lanes per chunk, sixteen chunks per wavefront, ordinary loads
        prog, entry, objs = _prog("lz4_compress.hip", "xxh32_items_quad_kernelILb0E")
        emu.launch(prog, entry, mem, struct.pack("<QQiIQ", a_src, a_items, n_items, SEED, a_check), (n_items + 15) // 16, 0, objects=objs)
        prog, entry, objs = _prog("lz4_compress.hip", "lz4_compress_l2_kernelILb1E")
        emu.launch(prog, entry, mem, struct.pack("<QQiiQQQQ", a_src, a_items, n_items, stride, a_check, a_slots, a_size, a_work), 1, 16384,
                   objects=objs)
        assert int(work[0]) == n_items + 1
    prog, entry, objs = _prog("assemble.hip", "scan_items_kernel")
    emu.launch(prog, entry, mem, struct.pack("<QiiQQiiQ", a_size, n_items, 0, a_off, a_pf, n, 0, a_index), 1, 0, objects=objs)
    if n_items:
        prog, entry, objs = _prog("assemble.hip", "gather_items_kernel")
        emu.launch(prog, entry, mem, struct.pack("<QQiiQqQQQqQ", a_src, a_items, n_items, 0, a_slots, stride, a_size, a_off,
                                                 a_dst, dst_bytes, a_status), n_items, 0, block_x=256, objects=objs)
    idx = [int(x) for x in index]
    sums = None
    if algo and int(status[0]) == 0:
        sums = ck.checksum_ranges(algo, dst.tobytes(), idx, data_len=dst_bytes)
    return int(status[0]), dst.tobytes(), idx, sums


def compress_map_outputs_batch(tasks, algo, dst_bytes_per_task):
    """A BATCHED map-side call (s3s_compress_map_outputs_batch_device, LZ4) through the compiled kernels: one frame-check pass
    and one persistent codec launch over the items of every task, then the tail kernels ONCE per call through TaskTail
    descriptors (round 6: scan_items_batch_kernel, gather_items_batch_kernel, checksum_segments_batch_kernel,
    checksum_combine_batch_kernel).  tasks: list of partition lists; every task has its own destination of exactly
    dst_bytes_per_task[t] bytes.  -> list of (status, image, index, checksums or None)."""
    T = len(tasks)
    stride = 32 + ((BLOCK + 15) & ~15)
    src = np.frombuffer(b"".join(b"".join(p) for p in tasks), dtype=np.uint8)
    items = bytearray()
    pf_all, seg_all = [], []          # packed: n_t + 1 entries per task
    first_item, first_part, first_seg, n_parts = [], [], [], []
    off = ch = seg = 0
    for parts in tasks:
        first_item.append(len(items) // 24)
        first_part.append(sum(n_parts))
        first_seg.append(seg)
        n_parts.append(len(parts))
        seg_t = 0
        for p, b in enumerate(parts):
            pf_all.append(len(items) // 24 - first_item[-1])
            seg_all.append(seg_t)
            nblk = (len(b) + BLOCK - 1) // BLOCK
            worst = nblk * (21 + BLOCK) + 21 if b else 0   # max_partition_size of the LZ4Block stream
            seg_t += (worst + ck.SEG - 1) // ck.SEG
            for pos in range(0, len(b), BLOCK):
                items += struct.pack("<qiiii", off + pos, min(BLOCK, len(b) - pos), 0 | (LEVEL << 8), ch, p)
                ch += 1
            if b:
                items += struct.pack("<qiiii", 0, 0, 1 | (LEVEL << 8), -1, p)
            off += len(b)
        pf_all.append(len(items) // 24 - first_item[-1])
        seg_all.append(seg_t)
        seg += seg_t
    n_items = len(items) // 24
    first_item.append(n_items)
    total_parts, total_segs = sum(n_parts), seg
    mem = emu.Memory()
    a_src = mem.map(src.copy() if src.size else np.zeros(1, np.uint8), "src", writable=False)
    a_items = mem.map(np.frombuffer(items or bytearray(24), dtype=np.uint8), "items", writable=False)
    check = np.zeros(max(n_items, 1), np.uint32)
    size = np.zeros(max(n_items, 1), np.uint32)
    item_off = np.full(n_items + T + 1, -7, np.int64)   # one extra entry per task
    index = np.full(total_parts + T, -7, np.int64)
    slots = np.zeros(max(ch, 1) * stride, np.uint8)
    work = np.zeros(1, np.uint32)
    status = np.zeros(T + 1, np.int32)
    dsts = [np.full(max(n, 1), 0xA5, np.uint8)[:n] for n in dst_bytes_per_task]
    a_dsts = [mem.map(d if d.size else np.zeros(1, np.uint8), "dst%d" % t) for t, d in enumerate(dsts)]
    a_check, a_size, a_off, a_index = mem.map(check, "item_check"), mem.map(size, "item_size"), mem.map(item_off, "item_off"), mem.map(index, "index")
    a_slots, a_work, a_status = mem.map(slots, "slots"), mem.map(work, "work"), mem.map(status, "status")
    a_pf = mem.map(np.array(pf_all, np.int32), "part_first", writable=False)
    a_seg = mem.map(np.array(seg_all, np.int32), "seg_start", writable=False)
    tails = bytearray()
    for t in range(T):
        tails += struct.pack("<iiiiiiiiQqQq", first_item[t], first_item[t + 1] - first_item[t], first_part[t] + t, n_parts[t],
                             first_part[t], first_seg[t], (first_seg[t + 1] if t + 1 < T else total_segs) - first_seg[t], 0,
                             a_dsts[t], dst_bytes_per_task[t], a_dsts[t], dst_bytes_per_task[t])
    a_tails = mem.map(np.frombuffer(tails, dtype=np.uint8), "tails", writable=False)
    if n_items:
        prog, entry, objs = _prog("lz4_compress.hip", "xxh32_items_quad_kernelILb0E")
        emu.launch(prog, entry, mem, struct.pack("<QQiIQ", a_src, a_items, n_items, SEED, a_check), (n_items + 15) // 16, 0, objects=objs)
        prog, entry, objs = _prog("lz4_compress.hip", "lz4_compress_l2_kernelILb1E")
        emu.launch(prog, entry, mem, struct.pack("<QQiiQQQQ", a_src, a_items, n_items, stride, a_check, a_slots, a_size, a_work), 1, 16384,
                   objects=objs)
    prog, entry, objs = _prog("assemble.hip", "scan_items_batch_kernel")
    emu.launch(prog, entry, mem, struct.pack("<QiiQQQQ", a_tails, T, 0, a_size, a_off, a_pf, a_index), T, 0, objects=objs)
    if n_items:
        prog, entry, objs = _prog("assemble.hip", "gather_items_batch_kernel")
        emu.launch(prog, entry, mem, struct.pack("<QiiQQQqQQQ", a_tails, T, n_items, a_src, a_items, a_slots, stride, a_size, a_off, a_status),
                   n_items, 0, block_x=256, objects=objs)
    sums = np.full(max(total_parts, 1), -1, np.int64)
    if algo and total_parts:
        a_tab = mem.map(ck.tables(0x82F63B78 if algo == 3 else 0xEDB88320), "tables", writable=False)
        k = 2 if algo == 3 else algo
        partial = np.zeros(max(4 * total_segs, 4), np.uint32)
        a_par, a_out = mem.map(partial, "partial"), mem.map(sums, "sums")
        text = lk.compile_asm("checksum.hip")
        objs = {kk: v for kk, v in emu.parse_objects(text).items() if kk.startswith("_ZN3s3s")}
        if total_segs:
            entry = lk.find_kernel(text, "checksum_segments_batch_kernelILi%dE" % k)
            emu.launch(emu.Program(text, entry), entry, mem, struct.pack("<QiiQQQQ", a_tails, T, 0, a_index, a_seg, a_tab, a_par), total_segs, 0,
                       block_x=256, objects=objs)
        entry = lk.find_kernel(text, "checksum_combine_batch_kernelILi%dE" % k)
        emu.launch(emu.Program(text, entry), entry, mem, struct.pack("<QiiQQQQQ", a_tails, T, total_parts, a_index, a_seg, a_tab, a_par, a_out),
                   total_parts, 0, objects=objs)
    res = []
    for t in range(T):
        pp = first_part[t] + t
        idx = [int(x) for x in index[pp:pp + n_parts[t] + 1]]
        sm = [int(x) for x in sums[first_part[t]:first_part[t] + n_parts[t]]] if algo else None
        res.append((int(status[t]), dsts[t].tobytes(), idx, sm))
    return res
