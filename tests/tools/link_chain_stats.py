"""Bounded structural experiment of round 2: would a pre-pass that emits, per position, a link to
the previous position with the same 13-bit hash — plus an "inserted" bitmap — let the serial LZ4 phase run without
its LDS hash table?  The sequential code's candidate for a probe is the most recent INSERTED same-hash position, so
the serial phase would walk the link chain until it meets an inserted position; every hop is a dependent global
(L2) load.  This script replays LZ4_compress_default's parse (byU16, acceleration 1) on 32 KiB chunks, checks the
replay against the oracle byte for byte, and reports how many hops the probes need.
usage: python tests/tools/link_chain_stats.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "spark-s3-shuffle_amd")); sys.path.insert(0, ROOT)
import numpy as np
from oracle import binding as oracle
from s3shuffle import datagen


def replay(d: bytes):
    n = len(d)
    rd32 = lambda p: int.from_bytes(d[p:p + 4], "little")  # noqa: E731
    H = lambda p: ((rd32(p) * 2654435761) & 0xFFFFFFFF) >> 19  # noqa: E731
    table = {}
    inserted, probes = set(), []
    out = bytearray()
    mfl1, matchlimit = n - 12 + 1, n - 5
    anchor = 0

    def put(p):
        table[H(p)] = p
        inserted.add(p)

    def emit(ip, match, mlen):
        lit = ip - anchor
        tok = (min(lit, 15) << 4) | min(mlen - 4, 15)
        out.append(tok)
        if lit >= 15:
            r = lit - 15
            while r >= 255: out.append(255); r -= 255
            out.append(r)
        out.extend(d[anchor:ip])
        out.extend((ip - match).to_bytes(2, "little"))
        if mlen - 4 >= 15:
            r = mlen - 4 - 15
            while r >= 255: out.append(255); r -= 255
            out.append(r)

    if n >= 13:
        put(0)
        ip = 1
        done = False
        while not done:
            fwd, step, nb = ip, 1, 64
            while True:
                ip = fwd
                fwd += step
                step = nb >> 6
                nb += 1
                if fwd > mfl1:
                    done = True
                    break
                h = H(ip)
                match = table.get(h, 0)
                probes.append(ip)
                put(ip)
                if rd32(match) == rd32(ip):
                    break
            if done:
                break
            while ip > anchor and match > 0 and d[ip - 1] == d[match - 1]:
                ip -= 1; match -= 1
            while True:
                m = 4
                while ip + m < matchlimit and d[ip + m] == d[match + m]:
                    m += 1
                emit(ip, match, m)
                ip += m
                anchor = ip
                if ip >= mfl1:
                    done = True
                    break
                put(ip - 2)
                match = table.get(H(ip), 0)
                probes.append(ip)
                put(ip)
                if rd32(match) == rd32(ip):
                    continue
                ip += 1
                break
    lit = n - anchor
    out.append(min(lit, 15) << 4)
    if lit >= 15:
        r = lit - 15
        while r >= 255: out.append(255); r -= 255
        out.append(r)
    out.extend(d[anchor:])
    return bytes(out), inserted, probes, H


for name, gen in (("terasort", lambda: datagen.skew_block(32768 * 4, "terasort", seed=5)[0]),
                  ("wide rows", lambda: datagen.tpcds_wide_map_output(160000, 1, seed=3)[0][:32768 * 4])):
    data = np.ascontiguousarray(gen())
    hops_all, ins_frac = [], []
    for c in range(4):
        d = data[c * 32768:(c + 1) * 32768]
        blk, inserted, probes, H = replay(d.tobytes())
        assert blk == oracle.lz4_compress_block(d).tobytes(), "replay differs from the oracle"
        last = {}
        link = [-1] * (len(d) - 3)
        for p in range(len(d) - 3):
            h = H(p)
            link[p] = last.get(h, -1)
            last[h] = p
        for p in probes:
            # hops from p back to the most recent same-hash position that was inserted BEFORE p was probed:
            # positions are inserted in increasing order, so "inserted and < p" is exact
            q, hops = link[p], 1
            while q >= 0 and q not in inserted:
                q = link[q]
                hops += 1
            if q >= 0:
                hops_all.append(hops)
        ins_frac.append(len(inserted) / len(d))
    h = np.array(hops_all)
    print(f"{name}: {len(h)} probes with a candidate; positions inserted {100 * np.mean(ins_frac):.0f} %; link hops to the first "
          f"inserted same-hash position: mean {h.mean():.2f}, p50 {np.percentile(h, 50):.0f}, p90 {np.percentile(h, 90):.0f}, "
          f"max {h.max()}; probes needing > 1 hop {100 * (h > 1).mean():.0f} %")
