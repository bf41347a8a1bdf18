"""Executed instructions per SOURCE line of a compiled kernel, from the gfx950 interpreter (TEST INFRASTRUCTURE).

The kernel source is compiled once more with -g (line tables only, same -O3 code), every instruction of the assembly is
attributed to the last `.loc` in front of it, and one block is run through the interpreter with a per-instruction
counter.  usage: python tests/tools/source_profile.py decode [lz4|snappy] [terasort|wide]   (default: decode lz4 terasort)
           python tests/tools/source_profile.py zstd [terasort|wide] [bytes]
"""
import collections
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "isa"), ROOT, os.path.join(ROOT, "spark-s3-shuffle_amd")):
    sys.path.insert(0, p)
import decode_kernel as dk  # noqa: E402
import gfx950_emu as emu  # noqa: E402
import lz4_kernel as lk  # noqa: E402
from oracle import binding as O  # noqa: E402
from s3shuffle import datagen  # noqa: E402


def zstd_main():
    """python tests/tools/source_profile.py zstd [terasort|wide] [bytes]: pass 2 of the compiled Zstandard decoder"""
    import zstd_kernel as zk
    from oracle import zstd_ref

    gen = datagen.tpcds_wide_map_output if len(sys.argv) > 2 and sys.argv[2] == "wide" else datagen.terasort_map_output
    nbytes = int(sys.argv[3]) if len(sys.argv) > 3 else 40000
    orig = lk.compile_asm
    lk.compile_asm = lambda name, flags=(), cache_dir=None: orig(name, tuple(flags) + ("-gline-tables-only",), cache_dir)
    zk._PROG = None
    prog, entry, text, lds = zk.program()
    files, loc, line_src = {}, None, {}
    for ln, raw in enumerate(text.splitlines(), 1):
        t = raw.strip()
        m = re.match(r'\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', t)
        if m:
            files[int(m.group(1))] = m.group(3) or m.group(2)
            continue
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
        if m:
            loc = (int(m.group(1)), int(m.group(2)))
            continue
        line_src[ln] = loc
    counts = collections.Counter()
    armed = [False]
    for i in prog.insts:
        if i.fn is not None:
            f = i.fn

            def g(w_, i_, f=f):
                if armed[0]:
                    counts[i_.line] += 1
                return f(w_, i_)
            i.fn = g
    data, _ = gen(1 << 20, 10, seed=3)
    blk = data[:nbytes]
    comp = bytes(zstd_ref.compress_stream(blk, level=1))
    orig_launch = emu.launch
    calls = [0]

    def launch(*a, **k):
        calls[0] += 1
        armed[0] = calls[0] == 2  # pass 2 only
        return orig_launch(*a, **k)
    emu.launch = launch
    out, rcs, waves = zk.decode_partitions([(comp, blk.size)])
    assert rcs == [0] and out[0] == blk.tobytes()
    by_src = collections.Counter()
    for ln, c in counts.items():
        by_src[line_src.get(ln)] += c
    tot = sum(by_src.values())
    print("pass 2: %d instructions with a handler for %d bytes (%d compressed)" % (tot, blk.size, len(comp)))
    srcs = {}
    for key, c in sorted(((k, v) for k, v in by_src.items()), key=lambda kv: -kv[1])[:int(os.environ.get("SP_TOP", "60"))]:
        if key is None:
            print("%8d %5.1f%%  (no line)" % (c, 100.0 * c / tot))
            continue
        fid, sl = key
        fn = files.get(fid, "?")
        if fn not in srcs:
            path = fn if os.path.isabs(fn) else os.path.join(ROOT, "spark-s3-shuffle_amd", "csrc", os.path.basename(fn))
            srcs[fn] = open(path).read().splitlines() if os.path.exists(path) else []
        code = srcs[fn][sl - 1].strip() if 0 < sl <= len(srcs[fn]) else ""
        print("%8d %5.1f%%  %s:%d  %s" % (c, 100.0 * c / tot, os.path.basename(fn), sl, code[:105]))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "zstd":
        return zstd_main()
    fmt = 1 if len(sys.argv) > 2 and sys.argv[2] == "snappy" else 0
    gen = datagen.tpcds_wide_map_output if len(sys.argv) > 3 and sys.argv[3] == "wide" else datagen.terasort_map_output
    src = "lz4_decode_batch.hip"
    orig = lk.compile_asm
    lk.compile_asm = lambda name, flags=(), cache_dir=None: orig(name, tuple(flags) + ("-gline-tables-only",), cache_dir)
    text = lk.compile_asm(src)
    # instruction line -> source line
    files, loc = {}, None
    line_src = {}
    for ln, raw in enumerate(text.splitlines(), 1):
        t = raw.strip()
        m = re.match(r'\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', t)
        if m:
            files[int(m.group(1))] = m.group(3) or m.group(2)
            continue
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
        if m:
            loc = (int(m.group(1)), int(m.group(2)))
            continue
        line_src[ln] = loc
    counts = collections.Counter()
    orig_run = emu.run_wave

    def run_wave(prog, w, entry=None, **kw):
        insts = prog.insts
        fns = {}
        for i in insts:  # wrap every instruction's handler once (counts by assembly line)
            if i.fn is not None and id(i) not in fns:
                f = i.fn
                fns[id(i)] = f

                def g(w_, i_, f=f):
                    counts[i_.line] += 1
                    return f(w_, i_)
                i.fn = g
        try:
            return orig_run(prog, w, entry, **kw)
        finally:
            for i in insts:
                if id(i) in fns:
                    i.fn = fns[id(i)]
    emu.run_wave = run_wave
    dk._PROGS.clear()
    data, _ = gen(1 << 20, 10, seed=3)
    cf = O.lz4_compress_block if fmt == 0 else O.snappy_compress_block
    blk = data[:32768]
    res, st, waves = dk.decode_blocks([(bytes(cf(blk)), 32768)], fmt=fmt)
    assert st == 0 and res[0] == blk.tobytes()
    by_src = collections.Counter()
    for ln, c in counts.items():
        by_src[line_src.get(ln)] += c
    tot = sum(by_src.values())
    print("instructions with a handler (branches and s_endpgm are not counted): %d" % tot)
    srcs = {}
    for (fid, sl), c in sorted(((k, v) for k, v in by_src.items() if k), key=lambda kv: -kv[1])[:45]:
        fn = files.get(fid, "?")
        if fn not in srcs:
            path = fn if os.path.isabs(fn) else os.path.join(ROOT, "spark-s3-shuffle_amd", "csrc", os.path.basename(fn))
            srcs[fn] = open(path).read().splitlines() if os.path.exists(path) else []
        code = srcs[fn][sl - 1].strip() if 0 < sl <= len(srcs[fn]) else ""
        print("%7d %5.1f%%  %s:%d  %s" % (c, 100.0 * c / tot, os.path.basename(fn), sl, code[:110]))


if __name__ == "__main__":
    main()
