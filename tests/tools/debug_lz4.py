"""GPU-vs-oracle LZ4 block diff: prints the first differing sequence for a set of corpora."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "spark-s3-shuffle_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import s3shuffle, corpus
from oracle import binding as o
from s3shuffle import datagen

def parse(c):
    i = 0; seqs = []; n = len(c); pos = 0
    while i < n:
        tok = c[i]; i += 1
        lit = tok >> 4
        if lit == 15:
            while True:
                b = c[i]; i += 1; lit += b
                if b != 255: break
        i += lit
        if i >= n:
            seqs.append((pos, lit, 0, 0)); break
        off = c[i] | (c[i + 1] << 8); i += 2
        ml = tok & 15
        if ml == 15:
            while True:
                b = c[i]; i += 1; ml += b
                if b != 255: break
        seqs.append((pos, lit, ml + 4, off)); pos += lit + ml + 4
    return seqs

def payload(img):
    clen = int.from_bytes(img[9:13].tobytes(), "little"); tok = img[8]
    return tok, img[21:21 + clen]

codec = s3shuffle.Codec(0)
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 2
codec.set_option(4, variant)
cases = []
for kind in (1, 2, 3, 4, 5, 7):
    for n in (255, 300, 1000, 4096, 32768):
        rng = np.random.default_rng(100 + kind)
        cases.append((f"kind{kind}", n, corpus.chunk_corpus(kind, n, rng)))
d, _ = datagen.skew_block(32768, "terasort", seed=5); cases.append(("terasort", 32768, d))
d, _ = datagen.tpcds_wide_map_output(200000, 1, seed=3); cases.append(("tpcds", 32768, d[:32768].copy()))
nbad = 0
for name, n, data in cases:
    try:
        img, index, sums = codec.compress_map_output(1, 0, data, [0, n])
    except Exception as e:
        print(name, n, "EXC", e); nbad += 1; continue
    rimg, rindex, _ = o.compress_map_output(1, 0, data, [0, n])
    if np.array_equal(img, rimg):
        print(name, n, "ok"); continue
    nbad += 1
    tg, pg = payload(img); tr, pr = payload(rimg)
    print(name, n, "MISMATCH tokens", hex(tg), hex(tr), "sizes", img.size, rimg.size)
    if tg != 0x25 or tr != 0x25: continue
    try:
        sg, sr = parse(pg.tolist()), parse(pr.tolist())
    except Exception as e:
        print("   parse failed", e); sr = parse(pr.tolist()); sg = []
    for i, (a, b) in enumerate(zip(sg, sr)):
        if a != b:
            print("   first differing sequence", i, "gpu (pos,lit,ml,off)=", a, "ref=", b, " window", b[0] >> 6, "lane", b[0] & 63)
            print("   prev:", sr[max(0, i - 2):i], " next ref:", sr[i + 1:i + 3], "next gpu:", sg[i + 1:i + 3])
            break
    else:
        m = min(pg.size, pr.size)
        diff = np.nonzero(pg[:m] != pr[:m])[0]
        print("   sequence lists equal up to", min(len(sg), len(sr)), "lens", len(sg), len(sr), "first differing payload byte", int(diff[0]) if diff.size else -1)
print("bad", nbad)
