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
    seqinfo = []
    cands, matches = [], []   # (probe position, candidate the table returned), (ip, match, length) of every sequence
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
                cands.append((ip, match))
                put(ip)
                if rd32(match) == rd32(ip):
                    break
            if done:
                break
            probe_pos = ip
            while ip > anchor and match > 0 and d[ip - 1] == d[match - 1]:
                ip -= 1; match -= 1
            replay.probe_of = getattr(replay, "probe_of", {})
            while True:
                m = 4
                while ip + m < matchlimit and d[ip + m] == d[match + m]:
                    m += 1
                emit(ip, match, m)
                matches.append((ip, match, m))
                pp = probe_pos if probe_pos >= ip else ip  # (a sequence found by the immediate re-probe behind a match has no backward part)
                k = 0  # what the gather sees: equal bytes in front of probe / candidate, up to 4, whatever the anchor
                cpos = match + (pp - ip)
                while k < 5 and pp - 1 - k >= 0 and cpos - 1 - k >= 0 and d[pp - 1 - k] == d[cpos - 1 - k]:
                    k += 1
                seqinfo.append((pp, ip + m, m - (pp - ip) - 4, k, cpos))  # probe, end, forward bytes beyond the 4, backward seen, candidate
                probe_pos = -1
                ip += m
                anchor = ip
                if ip >= mfl1:
                    done = True
                    break
                put(ip - 2)
                match = table.get(H(ip), 0)
                probes.append(ip)
                cands.append((ip, match))
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
    replay.cands, replay.matches, replay.seqinfo = cands, matches, seqinfo
    return bytes(out), inserted, probes, H


for name, gen in (("terasort", lambda: datagen.skew_block(32768 * 4, "terasort", seed=5)[0]),
                  ("wide rows", lambda: datagen.tpcds_wide_map_output(160000, 1, seed=3)[0][:32768 * 4])):
    data = np.ascontiguousarray(gen())
    hops_all, ins_frac = [], []
    ring_c, ring_m = [], []
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
        ring_c.extend(replay.cands); ring_m.extend(replay.matches)
    h = np.array(hops_all)
    print(f"{name}: {len(h)} probes with a candidate; positions inserted {100 * np.mean(ins_frac):.0f} %; link hops to the first "
          f"inserted same-hash position: mean {h.mean():.2f}, p50 {np.percentile(h, 50):.0f}, p90 {np.percentile(h, 90):.0f}, "
          f"max {h.max()}; probes needing > 1 hop {100 * (h > 1).mean():.0f} %")

    # round 4: would a ring of the most recent input bytes in LDS serve the window block's candidate
    # gather and its match extensions?  Distances probe -> candidate (what the gather reads) and sequence -> match source
    # (what an extension reads); per 64-byte window, how many probes have a candidate outside the ring.
    c = np.array(ring_c); dist = c[:, 0] - c[:, 1]
    m = np.array(ring_m); mdist = m[:, 0] - m[:, 1]
    print(f"  probes {len(c)}, sequences {len(m)} ({len(m) / (4 * 512):.2f} per window), mean match {m[:, 2].mean():.1f} B, "
          f"matches longer than 12 B (beyond the gather's 8 forward bytes + 4) {100 * (m[:, 2] > 12).mean():.0f} %")
    for ring in (1024, 2048, 4096, 8192):
        far = dist >= ring - 80
        win = np.bincount(c[far, 0] // 64 + 0, minlength=1)
        allw = np.bincount(c[:, 0] // 64)
        nwin = (allw > 0).sum()
        print(f"  ring {ring:5d}: candidate outside {100 * far.mean():5.1f} % of probes; windows with 0 far probes "
              f"{100 * (1 - (np.bincount(c[far, 0] // 64, minlength=len(allw)) > 0).sum() / nwin):5.1f} %, far probes per window "
              f"{far.sum() / nwin:.2f}; match sources outside {100 * (mdist >= ring - 80).mean():5.1f} % "
              f"(of matches > 12 B: {100 * (mdist[m[:, 2] > 12] >= ring - 80).mean():5.1f} %)")
    print("  candidate distance percentiles (10/50/90/99):", [int(np.percentile(dist, q)) for q in (10, 50, 90, 99)],
          " match distance:", [int(np.percentile(mdist, q)) for q in (10, 50, 90, 99)])

    # round 4: how much of a window's chain of sequences could be resolved lane-parallel?  A sequence is
    # SIMPLE if nothing about it needs the scalar run loop: its match is what the window's one gather saw (the probe is no
    # duplicate-hash suspect: no lower position of the same window has its hash), no suspect probe lies in the literal run in
    # front of it, and the match is settled by the gather's 16 bytes (<= 4 backward, <= 8 + 4 forward).  The lane-parallel chain
    # would take a window's leading run of simple sequences and hand the rest to the scalar loop.
    hashes = {}
    lead, total_seq, simple_seq, wins = [], 0, 0, 0
    for c_i in range(4):
        d = data[c_i * 32768:(c_i + 1) * 32768]
        blk, inserted, probes, H = replay(d.tobytes())
        hv = [H(p) for p in range(len(d) - 3)]
        pset = set(probes)
        by_win = {}
        for ip, mt, ml in replay.matches:
            by_win.setdefault(ip // 64, []).append((ip, mt, ml))
        prev_end = 0
        for w in sorted(by_win):
            wb = w * 64
            run, ok = 0, True
            for ip, mt, ml in by_win[w]:
                total_seq += 1
                # suspects among the probes of this sequence's literal run (window part) and its own probe lane
                lo = max(prev_end, wb)
                susp = False
                for p in range(lo, ip + 1):
                    if p in pset and any(hv[q] == hv[p] for q in range(wb, p) if q < len(hv)):
                        susp = True
                        break
                # the probe that found the match is at ip + (backward extension); backward <= 4 and forward <= 12 from the gather
                simple = (not susp) and ml <= 12
                if ok and simple:
                    run += 1
                    simple_seq += 1
                else:
                    ok = False
                prev_end = ip + ml
            lead.append(run)
            wins += 1
    lead = np.array(lead)
    print(f"  lane-parallel chains: {total_seq} sequences in {wins} windows with a sequence; leading simple run per window: mean {lead.mean():.2f}, "
          f"windows with a run >= 2: {100 * (lead >= 2).mean():.0f} %, >= 3: {100 * (lead >= 3).mean():.0f} %; sequences inside leading runs of >= 2: "
          f"{100 * lead[lead >= 2].sum() / max(total_seq, 1):.0f} % of all sequences")

    # round 4: the lane-parallel stage exactly as designed for the window block: chain = consecutive SIMPLE events from
    # the window's first event: match seen by the gather, probe lane in no duplicate-hash group of the window's live lanes, forward
    # length uncapped (< 8 beyond the 4), backward count exact (< 4, candidate >= 4), the match ends inside the window, at most 14
    # literals; the stage runs when the chain has at least 2 elements.
    cov = tot = stages = 0
    clen = []
    for c_i in range(4):
        d = data[c_i * 32768:(c_i + 1) * 32768]
        blk, inserted, probes, H = replay(d.tobytes())
        hv = [H(p) for p in range(len(d) - 3)]
        pset = set(probes)
        info = {pp: (end, fwd, k, cpos) for pp, end, fwd, k, cpos in replay.seqinfo}
        wins = {}
        for pp in sorted(info):
            wins.setdefault(pp // 64, []).append(pp)
        tot += len(info)
        prev_end = {}
        ends = sorted((pp, info[pp][0]) for pp in info)
        for w, pps in wins.items():
            wb = w * 64
            # live lanes: from the run start (the end of the last sequence that began before the window, or wb)
            live0 = wb
            for pp, e in ends:
                if pp < wb and e > live0: live0 = e
                if pp >= wb: break
            groups = {}
            for q in range(max(live0, wb), min(wb + 64, len(hv))):
                groups.setdefault(hv[q], []).append(q)
            dup = {q for g in groups.values() if len(g) > 1 for q in g}
            chain, start = 0, None
            for i, pp in enumerate(pps):
                end, fwd, k, cpos = info[pp]
                st = live0 if i == 0 else info[pps[i - 1]][0]
                # dup probes in the literal run in front of it make it not the next event
                lits_ok = not any((q in dup) for q in range(max(st, wb), pp) if q in pset)
                nb = min(k, pp - (st if i else st))
                lit = pp - nb - st
                simple = lits_ok and pp not in dup and fwd < 8 and k < 4 and cpos >= 4 and end < wb + 64 and lit <= 14
                if i == 0 and st < wb: lit = pp - nb - st  # literals from the previous window (anchor earlier) are fine up to 14
                if not simple: break
                chain += 1
            if chain >= 2:
                cov += chain; stages += 1
            clen.append(chain)
    clen = np.array(clen)
    print(f"  stage as designed: fires in {100 * stages / max(len(clen), 1):.0f} % of the windows with a sequence, covers {100 * cov / max(tot, 1):.0f} % of all sequences, "
          f"{cov / max(stages, 1):.2f} sequences per firing")
