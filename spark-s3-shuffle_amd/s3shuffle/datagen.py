"""Deterministic synthetic shuffle map outputs (SURVEY.md §8d workloads S1-S5).

Every generator returns `(data: np.uint8[U], offsets: np.int64[N+1])`: the serialized,
UNcompressed bytes of one map task, already grouped by reduce partition in ascending partition
order — exactly what the Spark writers push through `S3ShuffleMapOutputWriter` partition by
partition (S3ShuffleMapOutputWriter.scala:67-83).  Pure numpy, counter-based randomness, so the
same (seed, map_id) gives the same bytes on every host.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix64(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser, vectorised (uint64 wrap-around arithmetic)."""
    x = x.astype(np.uint64, copy=True)
    with np.errstate(over="ignore"):
        x += np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    return x


def _stream(seed: int, map_id: int, lane: int, idx: np.ndarray) -> np.ndarray:
    base = (np.uint64(seed) * np.uint64(0x100000001B3)) ^ (np.uint64(map_id) << np.uint64(40)) ^ (
        np.uint64(lane) << np.uint64(56))
    return _mix64(idx.astype(np.uint64) ^ base)


_HEX = np.frombuffer(b"0123456789ABCDEF", dtype=np.uint8)


def terasort_records(n_records: int, seed: int, map_id: int = 0, first_record: int = 0) -> np.ndarray:
    """TeraGen-like 100-byte records (workload S2/S4/S5): 10 B uniform-random key | 32 B
    ASCII-hex row id | 56 B filler (7 blocks of 8 identical letters cycling A-Z) | CR LF."""
    ids = np.arange(n_records, dtype=np.uint64) + np.uint64(first_record)
    rec = np.empty((n_records, 100), dtype=np.uint8)
    k0 = _stream(seed, map_id, 1, ids)
    k1 = _stream(seed, map_id, 2, ids)
    for b in range(8):
        rec[:, b] = (k0 >> np.uint64(8 * b)).astype(np.uint8)
    rec[:, 8] = k1.astype(np.uint8)
    rec[:, 9] = (k1 >> np.uint64(8)).astype(np.uint8)
    rowid = ids + (np.uint64(map_id) << np.uint64(32))
    rec[:, 10:26] = ord("0")
    for k in range(16):
        rec[:, 26 + k] = _HEX[((rowid >> np.uint64(4 * (15 - k))) & np.uint64(15)).astype(np.int64)]
    for j in range(7):
        rec[:, 42 + 8 * j: 50 + 8 * j] = (np.uint64(65) + ((ids + np.uint64(j)) % np.uint64(26))).astype(np.uint8)[:, None]
    rec[:, 98] = 13
    rec[:, 99] = 10
    return rec


def _group_fixed(rec: np.ndarray, part: np.ndarray, num_partitions: int) -> Tuple[np.ndarray, np.ndarray]:
    order = np.argsort(part, kind="stable")
    counts = np.bincount(part, minlength=num_partitions).astype(np.int64)
    offsets = np.zeros(num_partitions + 1, dtype=np.int64)
    np.cumsum(counts * rec.shape[1], out=offsets[1:])
    return rec[order].reshape(-1), offsets


def terasort_map_output(n_bytes: int, num_partitions: int, seed: int, map_id: int = 0
                        ) -> Tuple[np.ndarray, np.ndarray]:
    """One TeraSort map task: ~n_bytes of records range-partitioned on the key's top 16 bits
    (S2: N=200, S4: N=2000).  n_bytes is rounded down to whole records."""
    n_records = max(n_bytes // 100, 0)
    rec = terasort_records(n_records, seed, map_id)
    key16 = (rec[:, 0].astype(np.int64) << 8) | rec[:, 1].astype(np.int64)
    part = (key16 * num_partitions) >> 16
    return _group_fixed(rec, part, num_partitions)


def skew_block(n_bytes: int, kind: str, seed: int, map_id: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    """S5: one single-partition block of exactly n_bytes. kind: terasort | zeros | random."""
    if kind == "zeros":
        data = np.zeros(n_bytes, dtype=np.uint8)
    elif kind == "random":
        words = _stream(seed, map_id, 9, np.arange((n_bytes + 7) // 8, dtype=np.uint64))
        data = words.view(np.uint8)[:n_bytes].copy()
    elif kind == "terasort":
        data = terasort_records((n_bytes + 99) // 100, seed, map_id).reshape(-1)[:n_bytes].copy()
    else:
        raise ValueError(kind)
    return data, np.array([0, n_bytes], dtype=np.int64)


def _varint_encode(values: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """LEB128 of uint64 values < 2^35 -> (bytes, per-value lengths)."""
    v = values.astype(np.uint64)
    nb = np.ones(v.shape, dtype=np.int64)
    for k in range(1, 5):
        nb += (v >= np.uint64(1 << (7 * k))).astype(np.int64)
    starts = np.zeros(v.size + 1, dtype=np.int64)
    np.cumsum(nb, out=starts[1:])
    out = np.zeros(int(starts[-1]), dtype=np.uint8)
    for k in range(5):
        m = nb > k
        byte = ((v[m] >> np.uint64(7 * k)) & np.uint64(0x7F)).astype(np.uint8)
        byte |= ((nb[m] > k + 1).astype(np.uint8) << 7)
        out[starts[:-1][m] + k] = byte
    return out, nb


def _group_ragged(data: np.ndarray, rec_start: np.ndarray, rec_len: np.ndarray, part: np.ndarray,
                  num_partitions: int) -> Tuple[np.ndarray, np.ndarray]:
    order = np.argsort(part, kind="stable")
    lens = rec_len[order]
    dst_start = np.zeros(lens.size + 1, dtype=np.int64)
    np.cumsum(lens, out=dst_start[1:])
    src_idx = np.repeat(rec_start[order] - dst_start[:-1], lens) + np.arange(int(dst_start[-1]), dtype=np.int64)
    psize = np.bincount(part, weights=rec_len, minlength=num_partitions).astype(np.int64)
    offsets = np.zeros(num_partitions + 1, dtype=np.int64)
    np.cumsum(psize, out=offsets[1:])
    return data[src_idx], offsets


def kv_int_map_output(n_pairs: int, num_partitions: int, seed: int, map_id: int = 0
                      ) -> Tuple[np.ndarray, np.ndarray]:
    """S1: (Int, Int) pairs, keys uniform in [0, 1e5), Kryo-like zig-zag varints, hash
    partitioned (the reference's local[2] groupByKey / foldByKey shapes, S3ShuffleManagerTest)."""
    idx = np.arange(n_pairs, dtype=np.uint64)
    keys = (_stream(seed, map_id, 3, idx) % np.uint64(100000)).astype(np.int64)
    vals = (_stream(seed, map_id, 4, idx) % np.uint64(1000)).astype(np.int64) - 500
    zz = lambda a: ((a << 1) ^ (a >> 63)).astype(np.uint64)  # noqa: E731
    inter = np.empty(2 * n_pairs, dtype=np.uint64)
    inter[0::2] = zz(keys)
    inter[1::2] = zz(vals)
    data, nb = _varint_encode(inter)
    rec_len = nb[0::2] + nb[1::2]
    rec_start = np.zeros(n_pairs + 1, dtype=np.int64)
    np.cumsum(rec_len, out=rec_start[1:])
    part = keys % num_partitions
    return _group_ragged(data, rec_start[:-1], rec_len, part, num_partitions)


_WORDS = [w.encode() for w in (
    "ALPHA BRAVO CHARLIE DELTA ECHO FOXTROT GOLF HOTEL INDIA JULIET KILO LIMA MIKE NOVEMBER OSCAR "
    "PAPA QUEBEC ROMEO SIERRA TANGO UNIFORM VICTOR WHISKEY XRAY YANKEE ZULU").split()]


def tpcds_wide_map_output(n_bytes: int, num_partitions: int, seed: int, map_id: int = 0
                          ) -> Tuple[np.ndarray, np.ndarray]:
    """S3: UnsafeRow-like wide rows: i32 BE row size | 8 B null bitset | 12 x 8 B LE longs
    (4 low-cardinality dims, 4 sequential surrogate keys, 4 decimals) | 16-48 B ASCII tail."""
    approx = 4 + 8 + 96 + 32
    n = max(n_bytes // approx, 1)
    idx = np.arange(n, dtype=np.uint64)
    tail_len = (16 + (_stream(seed, map_id, 5, idx) % np.uint64(33))).astype(np.int64)
    row_len = 8 + 96 + tail_len
    rec_len = 4 + row_len
    rec_start = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(rec_len, out=rec_start[1:])
    data = np.zeros(int(rec_start[-1]), dtype=np.uint8)
    st = rec_start[:-1]
    for b in range(4):
        data[st + b] = (row_len >> (8 * (3 - b))).astype(np.uint8)
    cols = []
    for c in range(4):
        cols.append(_stream(seed, map_id, 10 + c, idx) % np.uint64(1000))
    for c in range(4):
        cols.append(idx + np.uint64((map_id << 24) + 1_000_000 * (c + 1)))
    for c in range(4):
        cols.append(_stream(seed, map_id, 20 + c, idx) % np.uint64(10_000_000))
    for c, col in enumerate(cols):
        for b in range(8):
            data[st + 12 + 8 * c + b] = (col >> np.uint64(8 * b)).astype(np.uint8)
    # ASCII tail: words from a small vocabulary, truncated/padded with '.' to tail_len
    vocab = np.zeros((len(_WORDS), 48), dtype=np.uint8) + ord(".")
    for i, w in enumerate(_WORDS):
        rep = (w + b" ") * 24
        vocab[i, :] = np.frombuffer(rep[:48], dtype=np.uint8)
    wsel = (_stream(seed, map_id, 6, idx) % np.uint64(len(_WORDS))).astype(np.int64)
    tails = vocab[wsel]
    col_idx = np.arange(48)[None, :]
    mask = col_idx < tail_len[:, None]
    pos = (st + 108)[:, None] + col_idx
    data[pos[mask]] = tails[mask]
    part = (cols[0].astype(np.int64) * 31 + cols[1].astype(np.int64)) % num_partitions
    return _group_ragged(data, st, rec_len, part, num_partitions)
