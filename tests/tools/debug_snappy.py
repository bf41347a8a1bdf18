"""GPU-vs-oracle Snappy diff: first differing element of the first differing chunk."""
import os, sys, struct
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "spark-s3-shuffle_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import s3shuffle, corpus
from oracle import binding as o

def elements(c):
    i = 0; ulen = 0; sh = 0
    while True:
        b = c[i]; i += 1; ulen |= (b & 0x7f) << sh; sh += 7
        if not b & 0x80: break
    els = []; pos = 0
    while i < len(c):
        t = c[i]; i += 1
        if t & 3 == 0:
            n = (t >> 2) + 1
            if n > 60:
                nb = n - 60; n = int.from_bytes(bytes(c[i:i + nb]), "little") + 1; i += nb
            els.append((pos, "L", n, 0)); i += n; pos += n
        elif t & 3 == 1:
            n = 4 + ((t >> 2) & 7); off = ((t >> 5) << 8) | c[i]; i += 1
            els.append((pos, "C1", n, off)); pos += n
        elif t & 3 == 2:
            n = (t >> 2) + 1; off = c[i] | (c[i + 1] << 8); i += 2
            els.append((pos, "C2", n, off)); pos += n
        else:
            n = (t >> 2) + 1; off = int.from_bytes(bytes(c[i:i + 4]), "little"); i += 4
            els.append((pos, "C4", n, off)); pos += n
    return ulen, els

def chunks(stream):
    out = []; ip = 16
    while ip < len(stream):
        cl = struct.unpack(">i", bytes(stream[ip:ip + 4]))[0]; ip += 4
        out.append(stream[ip:ip + cl]); ip += cl
    return out

codec = s3shuffle.Codec(0)
nbad = 0
for algo in (1, 2, 0):
    rng = np.random.default_rng(17 + algo)
    for it in range(5):
        data, offsets = corpus.ragged_map_output(rng, n_parts=int(rng.integers(1, 30)), max_len=120_000)
        r0 = int(rng.integers(0, len(offsets) - 1)); r1 = int(rng.integers(r0 + 1, len(offsets)))
        img, index, sums = codec.compress_map_output(2, 0, data, offsets)
        rimg, rindex, _ = o.compress_map_output(2, 0, data, offsets)
        if np.array_equal(img, rimg): continue
        for p in range(len(offsets) - 1):
            g = img[index[p]:index[p + 1]].tolist(); r = rimg[rindex[p]:rindex[p + 1]].tolist()
            if g == r: continue
            nbad += 1
            cg, cr = chunks(g), chunks(r)
            for k, (a, b) in enumerate(zip(cg, cr)):
                if a != b:
                    ug, eg = elements(a); ur, er = elements(b)
                    print(f"algo {algo} it {it} part {p} len {offsets[p+1]-offsets[p]} chunk {k} ulen {ug}/{ur} sizes {len(a)}/{len(b)}")
                    for j, (x, y) in enumerate(zip(eg, er)):
                        if x != y:
                            print("   first differing element", j, "gpu", x, "ref", y, "prev", er[max(0, j - 2):j], "next ref", er[j + 1:j + 3], "next gpu", eg[j + 1:j + 3])
                            break
                    else:
                        print("   elements equal up to", min(len(eg), len(er)), len(eg), len(er))
                    break
            break
print("bad", nbad)
