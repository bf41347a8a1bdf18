"""(GPU) Mutated Zstandard partitions through the product's batched reduce-side call, many per launch (TEST INFRASTRUCTURE).

    python tests/tools/gpu_zstd_fuzz.py --seed 7 --rounds 20 --parts 400

Every round builds `--parts` partitions - libzstd frames of TeraSort / wide-row pieces of 1 .. 700 KB (one to six blocks), most
of them damaged (bit flips, random bytes, truncations, splices) - and decodes them as ONE range per partition in one
s3s_decompress_ranges_batch_device call, so that hundreds of workgroups with damaged streams run the literal wavefront, the lean
sequence loop and the compiled loop side by side on the real machine.  Per partition the verdict must be libzstd's: refused
(bad frame / capacity), or the same bytes (a few per cent differ in the two documented directions: the product refuses a sequence
stream that reads below its first bit; libzstd refuses offsets beyond its window, the product's history is the whole frame).  A hang would show as the call not returning (the polls are bounded: kPipeSpins)."""
import argparse
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
ROOT = os.path.dirname(TESTS)
for p in (ROOT, TESTS, os.path.join(ROOT, "spark-s3-shuffle_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--rounds", type=int, default=10)
    ap.add_argument("--parts", type=int, default=300)
    ap.add_argument("--host-model", action="store_true", help="decode with the host build of the same core instead (no GPU): the verdict counts must be the GPU run's")
    a = ap.parse_args()
    from oracle import zstd_ref as z
    from s3shuffle import datagen

    if a.host_model:
        import ctypes
        import subprocess

        so = os.path.join(TESTS, "model", "zstd_decode_model.so")
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", os.path.join(TESTS, "model", "zstd_decode_model.cpp"), "-o", so], check=True)
        model = ctypes.CDLL(so)
        model.zs_decode.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_int64)]
    else:
        import s3shuffle
        from hipdev import Dev

    rng = np.random.default_rng(a.seed)
    tera = datagen.terasort_map_output(8 << 20, 1, seed=a.seed)[0]
    wide = datagen.tpcds_wide_map_output(8 << 20, 1, seed=a.seed + 1)[0]
    codec = dev = None
    if not a.host_model:
        codec = s3shuffle.Codec(0)
        dev = Dev()
    n_ok = n_ref = n_strict = n_lenient = 0
    strict_by = {}
    t0 = time.time()
    for rnd in range(a.rounds):
        parts = []
        for k in range(a.parts):
            src = tera if rng.integers(0, 2) else wide
            n = int(rng.choice([1000, 20_000, 140_000, 300_000, 700_000], p=[0.2, 0.3, 0.2, 0.2, 0.1]))
            at = int(rng.integers(0, src.size - n))
            data = src[at:at + n]
            comp = z.compress_stream(data, 1, checksum=bool(rng.integers(0, 4) == 0)).copy()
            how = int(rng.integers(0, 6))
            if how == 0:
                pass
            elif how == 1:
                for _ in range(int(rng.integers(1, 4))):
                    comp[int(rng.integers(0, comp.size))] ^= 1 << int(rng.integers(0, 8))
            elif how == 2:
                i = int(rng.integers(0, comp.size))
                comp[i:i + 8] = rng.integers(0, 256, min(8, comp.size - i), dtype=np.uint8)
            elif how == 3:
                comp = comp[: int(rng.integers(1, comp.size))]
            elif how == 4 and parts:
                other = parts[int(rng.integers(0, len(parts)))][0]
                cut = int(rng.integers(1, min(comp.size, other.size)))
                comp = np.concatenate([comp[:cut], other[cut:]])
            else:
                comp[int(rng.integers(0, min(comp.size, 64)))] = int(rng.integers(0, 256))
            parts.append((comp, n))
        args, bufs = [], []
        if a.host_model:
            res, backs = [], []
            for comp, n in parts:
                cap = n + 4096
                out = np.full(cap + 32, 0x5A, np.uint8)
                c = np.ascontiguousarray(comp)
                tot = ctypes.c_int64(0)
                rc = model.zs_decode(c.ctypes.data, c.size, out.ctypes.data, cap, ctypes.byref(tot))
                res.append((rc, tot.value, -1))
                bufs.append((out, cap))
        else:
            for comp, n in parts:
                cap = n + 4096
                d_out = dev.upload(np.full(cap + 32, 0x5A, np.uint8))
                bufs.append((d_out, cap))
                args.append((dev.upload(np.ascontiguousarray(comp)), comp.size, np.array([0, comp.size], np.int64), None, d_out, cap))
            res = codec.decompress_ranges_batch_device(s3shuffle.CODEC_ZSTD, s3shuffle.CHECKSUM_NONE, args, raise_on_error=False)
        for (comp, n), (st, got, bad), (d_out, cap) in zip(parts, res, bufs):
            ref = z.decompress(comp, cap)
            back = d_out if a.host_model else dev.download(d_out, cap + 32)
            assert np.all(back[cap:] == 0x5A), "wrote behind the destination"
            if st == 0 and ref is None:
                # libzstd refuses through a limit the product does not share (its window-size bound on offsets: the product's
                # history is the whole frame) - as in tests/test_zstd_model.py; the destination is intact behind `cap`
                n_lenient += 1
            elif st == 0:
                assert got == ref.size and np.array_equal(back[:got], ref), "decoded differently from libzstd"
                n_ok += 1
            else:
                assert st in (-3, -2, -6), st
                if ref is not None:
                    n_strict += 1  # (stricter in two documented places: a sequence stream read below its first bit, -3; a frame
                    strict_by[st] = strict_by.get(st, 0) + 1  # header that names a dictionary id, -6, which libzstd ignores without one)
                n_ref += 1
        if dev:
            dev.free()  # (every buffer of the round)
        print(f"round {rnd}: {n_ok} decoded like libzstd, {n_ref} refused ({n_strict} of them accepted by libzstd), {n_lenient} decoded where libzstd refuses, {time.time() - t0:.0f} s", flush=True)
    assert n_strict <= 0.03 * (n_ok + n_ref) + 2 and n_lenient <= 0.03 * (n_ok + n_ref) + 2, (n_strict, n_lenient)
    print(f"DONE seed {a.seed}: {a.rounds * a.parts} partitions, {n_ok} decoded like libzstd, {n_ref} refused, {n_strict} strict {strict_by}, {n_lenient} lenient, 0 failures")


if __name__ == "__main__":
    main()
