"""Test helpers that build codec streams WITHOUT the oracle and without the product: LZ4Block / SnappyInputStream
framing around arbitrary block payloads, a sequence-level LZ4 / Snappy block writer for hand-made (also malformed)
blocks, and tiny reference decoders.  Used by the decoder hardening tests.

Framing follows [EXT] lz4-java LZ4BlockOutputStream / snappy-java SnappyOutputStream as restated in SURVEY.md §8
rows a3 / a4."""
from __future__ import annotations

import struct
from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np

LZ4_SEED = 0x9747B28C


def xxh32(data: bytes, seed: int = LZ4_SEED) -> int:
    import xxhash

    return xxhash.xxh32(data, seed=seed).intdigest()


def lz4_frame(payload: bytes, orig: bytes, raw: bool = False, check: Optional[int] = None) -> bytes:
    """One LZ4Block frame: magic | token | compressedLen LE | originalLen LE | check LE | payload."""
    if check is None:
        check = xxh32(orig) & 0x0FFFFFFF
    token = 0x15 if raw else 0x25
    return b"LZ4Block" + bytes([token]) + struct.pack("<iiI", len(payload), len(orig), check) + payload


def lz4_end_frame() -> bytes:
    return b"LZ4Block" + bytes([0x15]) + struct.pack("<iii", 0, 0, 0)


def lz4_stream(blocks: Iterable[Tuple[bytes, bytes]]) -> bytes:
    """blocks: (payload, original bytes) pairs, every payload an LZ4 block that decodes to `original`."""
    out = bytearray()
    for payload, orig in blocks:
        out += lz4_frame(payload, orig)
    out += lz4_end_frame()
    return bytes(out)


def snappy_stream(blocks: Iterable[bytes]) -> bytes:
    """SnappyOutputStream image: 16-byte header, then i32 BE compressedLen | raw snappy block."""
    out = bytearray(b"\x82SNAPPY\x00" + struct.pack(">ii", 1, 1))
    for b in blocks:
        out += struct.pack(">i", len(b)) + b
    return bytes(out)


# ---- LZ4 blocks from sequences ------------------------------------------------------------------------------------
def _lz4_len_bytes(n: int) -> bytes:
    out = bytearray()
    while n >= 255:
        out.append(255)
        n -= 255
    out.append(n)
    return bytes(out)


def lz4_block(seqs: Sequence[Tuple[bytes, int, int]], last_literals: bytes) -> bytes:
    """seqs: (literals, offset, match_length >= 4); the block ends with a literal-only sequence."""
    out = bytearray()
    for lit, off, ml in seqs:
        assert ml >= 4
        ll, mc = len(lit), ml - 4
        out.append((min(ll, 15) << 4) | min(mc, 15))
        if ll >= 15:
            out += _lz4_len_bytes(ll - 15)
        out += lit
        out += struct.pack("<H", off & 0xFFFF)
        if mc >= 15:
            out += _lz4_len_bytes(mc - 15)
    ll = len(last_literals)
    out.append(min(ll, 15) << 4)
    if ll >= 15:
        out += _lz4_len_bytes(ll - 15)
    out += last_literals
    return bytes(out)


def lz4_decode_py(block: bytes, max_out: int = 1 << 20) -> Optional[bytes]:
    """Plain LZ4 block decoder (None if malformed or longer than max_out)."""
    ip, n, out = 0, len(block), bytearray()
    while True:
        if ip >= n:
            return None
        tok = block[ip]
        ip += 1
        ll = tok >> 4
        if ll == 15:
            while True:
                if ip >= n:
                    return None
                b = block[ip]
                ip += 1
                ll += b
                if b != 255:
                    break
        if ip + ll > n:
            return None
        out += block[ip:ip + ll]
        ip += ll
        if len(out) > max_out:
            return None
        if ip == n:
            return bytes(out)
        if ip + 2 > n:
            return None
        off = block[ip] | (block[ip + 1] << 8)
        ip += 2
        ml = tok & 15
        if ml == 15:
            while True:
                if ip >= n:
                    return None
                b = block[ip]
                ip += 1
                ml += b
                if b != 255:
                    break
        ml += 4
        if off == 0 or off > len(out):
            return None
        for _ in range(ml):
            out.append(out[-off])
        if len(out) > max_out:
            return None


# ---- Snappy blocks from elements ----------------------------------------------------------------------------------
def _varint(n: int) -> bytes:
    out = bytearray()
    while n >= 0x80:
        out.append((n & 0x7F) | 0x80)
        n >>= 7
    out.append(n)
    return bytes(out)


def snappy_block(elements: Sequence[tuple], ulen: Optional[int] = None, force_len_bytes: int = 0) -> bytes:
    """elements: ("lit", bytes) | ("copy", offset, length[, tag_kind 1|2|4]); ulen overrides the preamble."""
    body = bytearray()
    total = 0
    for e in elements:
        if e[0] == "lit":
            data = e[1]
            n = len(data) - 1
            nb = force_len_bytes
            if nb == 0 and n >= 60:
                nb = 1 if n < 256 else 2 if n < 65536 else 3 if n < (1 << 24) else 4
            if nb == 0:
                body.append(n << 2)
            else:
                body.append((59 + nb) << 2)
                body += n.to_bytes(nb, "little")
            body += data
            total += len(data)
        else:
            off, ln = e[1], e[2]
            kind = e[3] if len(e) > 3 else (1 if 4 <= ln <= 11 and off < 2048 else 2)
            if kind == 1:
                body.append(1 | ((ln - 4) << 2) | ((off >> 8) << 5))
                body.append(off & 0xFF)
            elif kind == 2:
                body.append(2 | ((ln - 1) << 2))
                body += struct.pack("<H", off & 0xFFFF)
            else:
                body.append(3 | ((ln - 1) << 2))
                body += struct.pack("<I", off & 0xFFFFFFFF)
            total += ln
    return _varint(total if ulen is None else ulen) + bytes(body)


def snappy_decode_py(block: bytes, max_out: int = 1 << 20) -> Optional[bytes]:
    ip, n = 0, len(block)
    ulen, sh = 0, 0
    while True:
        if ip >= n or sh > 28:
            return None
        b = block[ip]
        ip += 1
        ulen |= (b & 0x7F) << sh
        if not b & 0x80:
            break
        sh += 7
    out = bytearray()
    while ip < n:
        tag = block[ip]
        ip += 1
        ty = tag & 3
        if ty == 0:
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                if ip + nb > n:
                    return None
                ln = int.from_bytes(block[ip:ip + nb], "little")
                ip += nb
            ln += 1
            if ip + ln > n or len(out) + ln > ulen:
                return None
            out += block[ip:ip + ln]
            ip += ln
            continue
        if ty == 1:
            if ip + 1 > n:
                return None
            ln = 4 + ((tag >> 2) & 7)
            off = ((tag >> 5) << 8) | block[ip]
            ip += 1
        elif ty == 2:
            if ip + 2 > n:
                return None
            ln = (tag >> 2) + 1
            off = block[ip] | (block[ip + 1] << 8)
            ip += 2
        else:
            if ip + 4 > n:
                return None
            ln = (tag >> 2) + 1
            off = int.from_bytes(block[ip:ip + 4], "little")
            ip += 4
        if off == 0 or off > len(out) or len(out) + ln > ulen:
            return None
        for _ in range(ln):
            out.append(out[-off])
    return bytes(out) if len(out) == ulen else None


def liblz4():
    import ctypes

    L = ctypes.CDLL("liblz4.so.1")
    L.LZ4_compress_HC.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    L.LZ4_compress_default.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    L.LZ4_decompress_safe.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    return L


def lz4_fast(d: np.ndarray) -> bytes:
    """liblz4's LZ4_compress_default on the whole input (>= 64 KiB: its byU32 parse - what a JVM writer with
    spark.io.compression.lz4.blockSize above 32k puts into one LZ4Block frame)"""
    L = liblz4()
    out = np.empty(d.size + d.size // 200 + 64, np.uint8)
    n = L.LZ4_compress_default(d.ctypes.data, out.ctypes.data, d.size, out.size)
    assert n > 0
    return out[:n].tobytes()


def lz4_hc(d: np.ndarray, level: int = 9) -> bytes:
    L = liblz4()
    out = np.empty(d.size + d.size // 200 + 64, np.uint8)
    n = L.LZ4_compress_HC(d.ctypes.data, out.ctypes.data, d.size, out.size, level)
    assert n > 0
    return out[:n].tobytes()
