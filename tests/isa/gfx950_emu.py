"""A small gfx950 (CDNA4, wave64) ISA interpreter for the integer kernels of this repository.

TEST INFRASTRUCTURE ONLY.  It executes the assembly text `hipcc -S --cuda-device-only` prints for a kernel
(including the inline-asm blocks) one wavefront at a time on the CPU, so that the *compiled* HIP kernels can be
checked bit for bit against the oracle without a GPU, every global / LDS access is bounds-checked, and the
instructions, taken branches and memory round trips of a path can be counted.  It models what the kernels here
use: the scalar and vector integer ALU, SDWA / DPP operand forms, LDS, global memory, SMEM loads, EXEC / VCC /
SCC.  It does not model timing, the hardware's hazards (tests/isa/hazards.py checks the hand-written blocks for
those), floating point, MFMA, or more than one wavefront per workgroup.
"""
import copy
import re
import numpy as np

M32 = 0xFFFFFFFF
M64 = 0xFFFFFFFFFFFFFFFF
LANES = np.arange(64, dtype=np.uint64)
VCC = 106  # vcc lives in the SGPR array at 106/107
_FLOAT_CONST = {"0.5": 0x3F000000, "1.0": 0x3F800000, "2.0": 0x40000000, "4.0": 0x40800000,
                "-0.5": 0xBF000000, "-1.0": 0xBF800000, "-2.0": 0xC0000000, "-4.0": 0xC0800000}


class EmuError(Exception):
    pass


class MemFault(EmuError):
    pass


def s32(x):
    x &= M32
    return x - (1 << 32) if x & 0x80000000 else x


def s64(x):
    x &= M64
    return x - (1 << 64) if x & (1 << 63) else x


class Memory:
    """Flat 64-bit address space made of named regions; every access is bounds-checked."""

    def __init__(self):
        self.regions = []
        self.next_base = 0x7F0000000000

    def map(self, data, name, writable=True, guard=1 << 24):
        arr = data if isinstance(data, np.ndarray) else np.frombuffer(bytearray(data), dtype=np.uint8)
        arr = arr.view(np.uint8).reshape(-1)
        base = self.next_base
        self.next_base += ((arr.size + guard + 0xFFFF) >> 16) << 16
        self.regions.append((base, arr, name, writable))
        return base

    def find(self, addr):
        for base, arr, name, wr in self.regions:
            if base <= addr < base + max(arr.size, 1):
                return base, arr, name, wr
        raise MemFault("address 0x%x is not mapped" % addr)

    def _locate(self, addrs, active, n, write):
        idx = np.flatnonzero(active)
        if idx.size == 0:
            return None, None
        base, arr, name, wr = self.find(int(addrs[idx[0]]))
        if write and not wr:
            raise MemFault("write to read-only region %s" % name)
        offs = addrs.astype(np.int64) - base
        a = offs[idx]
        if a.min() < 0 or a.max() + n > arr.size:
            bad = idx[(a < 0) | (a + n > arr.size)][0]
            raise MemFault("lane %d: %d-byte %s at %s%+d (region size %d)" % (
                bad, n, "store" if write else "load", name, int(offs[bad]), arr.size))
        return arr, np.where(active, offs, 0)

    def _groups(self, addrs, active):
        """the active lanes grouped by the region their address falls into (one group in nearly every access: the fast path
        of _locate; kernels that follow per-lane descriptors — the batched frame discovery — touch several regions at once)"""
        idx = np.flatnonzero(active)
        if idx.size == 0:
            return
        base, arr, _, _ = self.find(int(addrs[idx[0]]))
        a = addrs.astype(np.int64)[idx] - base
        if a.min() >= 0 and a.max() < max(arr.size, 1):
            yield active
            return
        left = active.copy()
        while left.any():
            first = int(np.flatnonzero(left)[0])
            base, arr, _, _ = self.find(int(addrs[first]))
            o = addrs.astype(np.int64) - base
            grp = left & (o >= 0) & (o < max(arr.size, 1))
            yield grp
            left = left & ~grp

    def load(self, addrs, active, n):
        """-> uint8[64, n] (zeros for inactive lanes)"""
        out = np.zeros((64, n), dtype=np.uint8)
        for grp in self._groups(addrs, active):
            arr, offs = self._locate(addrs, grp, n, False)
            got = arr[offs[:, None] + np.arange(n)]
            out[grp] = got[grp]
        return out

    def store(self, addrs, active, data):
        n = data.shape[1]
        for grp in self._groups(addrs, active):
            arr, offs = self._locate(addrs, grp, n, True)
            for lane in np.flatnonzero(grp):  # lane order: the highest lane wins a same-address race
                o = int(offs[lane])
                arr[o:o + n] = data[lane]

    def load_scalar(self, addr, n):
        base, arr, name, wr = self.find(addr)
        o = addr - base
        if o + n > arr.size:
            raise MemFault("scalar load of %d bytes at %s+%d (size %d)" % (n, name, o, arr.size))
        return bytes(arr[o:o + n])


# ---------------------------------------------------------------------------------------------------------------
# operands
_REG_RE = re.compile(r"^([sva])(\d+)$")
_RANGE_RE = re.compile(r"^([sva])\[(\d+):(\d+)\]$")


def parse_operand(tok):
    tok = tok.strip()
    neg = False
    m = _REG_RE.match(tok)
    if m:
        return (m.group(1), int(m.group(2)), 1)
    m = _RANGE_RE.match(tok)
    if m:
        return (m.group(1), int(m.group(2)), int(m.group(3)) - int(m.group(2)) + 1)
    if tok == "vcc":
        return ("s", VCC, 2)
    if tok == "vcc_lo":
        return ("s", VCC, 1)
    if tok == "vcc_hi":
        return ("s", VCC + 1, 1)
    if tok in ("exec", "exec_lo", "exec_hi", "scc", "m0", "off", "null"):
        return (tok, 0, 2 if tok == "exec" else 1)
    if tok in _FLOAT_CONST:
        return ("lit", _FLOAT_CONST[tok], 1)
    try:
        return ("lit", int(tok, 0) & M64, 1)
    except ValueError:
        pass
    m = re.match(r"^([A-Za-z_.$][\w.$]*)@rel32@(lo|hi)(\+\d+)?$", tok)
    if m:
        # s_getpc_b64 + `sym@rel32@lo+4` / `sym@rel32@hi+12` is the address of sym itself; other addends shift it
        return ("sym_" + m.group(2), (m.group(1), int(m.group(3) or "+0")), 1)
    m = re.match(r"^(sext)\((.*)\)$", tok)
    if m:
        inner = parse_operand(m.group(2))
        return ("sext", inner, 1)
    if len(tok) > 2 and tok[0] == "|" and tok[-1] == "|":
        return ("fabs", parse_operand(tok[1:-1]), 1)
    if len(tok) > 1 and tok[0] == "-" and (tok[1] in "sv|" or tok[1:].startswith("vcc")):
        return ("fneg", parse_operand(tok[1:]), 1)
    return ("label", tok, 0)


class Inst:
    __slots__ = ("op", "ops", "mods", "text", "line", "fn", "kind", "target")

    def __init__(self, op, ops, mods, text, line):
        self.op, self.ops, self.mods, self.text, self.line = op, ops, mods, text, line
        self.fn = None
        self.kind = None
        self.target = None


_MOD_RE = re.compile(r"\b(offset0|offset1|offset|dst_sel|dst_unused|src0_sel|src1_sel|row_shr|row_shl|row_ror|wave_shr|wave_shl|wave_ror|"
                     r"wave_rol|row_mask|bank_mask|bound_ctrl|quad_perm|row_bcast|op_sel|op_sel_hi|bitop3)\s*:\s*(\[[^\]]*\]|\S+)")
_FLAG_RE = re.compile(r"\b(glc|slc|sc0|sc1|nt|row_mirror|row_half_mirror|clamp|lds)\b")


def parse_program(text, entry=None):
    """assembly text -> (list of Inst, {label: index}).  With `entry` (a symbol name), only that function."""
    insts, labels = [], {}
    active = entry is None
    for ln, raw in enumerate(text.splitlines(), 1):
        line = raw.split(";")[0].strip() if not raw.strip().startswith(";;#") else ""
        if not line:
            continue
        if line.endswith(":") and not line.startswith("."):
            name = line[:-1]
            if entry is not None:
                if name == entry:
                    active = True
                    labels[name] = len(insts)
                    continue
                if active and not name.startswith(".L"):
                    break
            if active:
                labels[name] = len(insts)
            continue
        if not active:
            continue
        if line.startswith(".L") and line.endswith(":"):
            labels[line[:-1]] = len(insts)
            continue
        if line.startswith("."):
            if line.startswith(".section") and insts and entry is not None:
                break
            continue
        parts = line.split(None, 1)
        op = parts[0]
        rest = parts[1] if len(parts) > 1 else ""
        mods = {}
        for m in _MOD_RE.finditer(rest):
            mods[m.group(1)] = m.group(2)
        rest = _MOD_RE.sub("", rest)
        for m in _FLAG_RE.finditer(rest):
            mods[m.group(1)] = True
        rest = _FLAG_RE.sub("", rest)
        if op == "s_waitcnt":
            ops = []
        else:
            # split on commas that are not inside [...]
            toks, depth, cur = [], 0, ""
            for ch in rest:
                if ch == "[":
                    depth += 1
                elif ch == "]":
                    depth -= 1
                if ch == "," and depth == 0:
                    toks.append(cur)
                    cur = ""
                else:
                    cur += ch
            if cur.strip():
                toks.append(cur)
            ops = [parse_operand(t) for t in toks if t.strip()]
        insts.append(Inst(op, ops, mods, line, ln))
    return insts, labels


# ---------------------------------------------------------------------------------------------------------------
def _ff1(x, bits):
    if x == 0:
        return M32  # -1
    return (x & -x).bit_length() - 1


def _flbit(x, bits):  # count leading zeros, -1 for 0
    if x == 0:
        return M32
    return bits - x.bit_length()


def _bfe_u32(v, spec):
    off = spec & 31
    width = (spec >> 16) & 0x7F
    if width == 0:
        return 0
    return (v >> off) & ((1 << width) - 1) if width < 32 else (v >> off)


class Wave:
    def __init__(self, mem, lds_bytes=65536):
        self.s = [0] * 128
        self.v = np.zeros((512, 64), dtype=np.uint32)
        self.scc = 0
        self.exec = M64
        self.m0 = 0
        self.mem = mem
        self.lds = np.zeros(lds_bytes, dtype=np.uint8)
        self._em_cache = {}
        self.count = {}
        self.n_inst = 0
        self.n_taken = 0
        self.n_salu = self.n_valu = self.n_lds = self.n_vmem = self.n_smem = 0
        self.clock = 0
        self.sgpr_ready = {}  # SGPR -> cycle at which a VALU-produced value can be read by the scalar unit (Cost.CROSS)
        self.trace = None
        self.region_counts = None
        # outstanding memory operations, oldest first: lists of pending destination VGPR numbers ([] for stores)
        self.q_vm = []
        self.q_lgkm = []
        self.pending = {}  # VGPR -> text of the load that has not been waited for
        self.t_vm, self.t_lgkm = [], []  # cycle model: return times of the outstanding operations (same order as q_vm / q_lgkm)
        self.n_wait_events = 0

    # ---- helpers
    def em(self, mask=None):
        mask = self.exec if mask is None else mask
        r = self._em_cache.get(mask)
        if r is None:
            r = ((np.uint64(mask) >> LANES) & np.uint64(1)).astype(bool)
            if len(self._em_cache) > 4096:
                self._em_cache.clear()
            self._em_cache[mask] = r
        return r

    @staticmethod
    def mask_of(boolarr):
        return int(np.bitwise_or.reduce(np.where(boolarr, np.uint64(1) << LANES, np.uint64(0))))

    def rs32(self, o):
        k = o[0]
        if k == "s":
            return self.s[o[1]]
        if k == "lit":
            return o[1] & M32
        if k == "exec_lo":
            return self.exec & M32
        if k == "exec_hi":
            return self.exec >> 32
        if k == "exec":
            return self.exec & M32
        if k == "scc":
            return self.scc
        if k == "m0":
            return self.m0
        if k == "sym_lo":
            return (self.symbols[o[1][0]] + o[1][1] - 4) & M32
        if k == "sym_hi":  # (the carry of the low add comes through s_addc_u32)
            return ((self.symbols[o[1][0]] + o[1][1] - 12) >> 32) & M32
        raise EmuError("bad scalar source %r" % (o,))

    def rs64(self, o):
        k = o[0]
        if k == "s":
            if o[2] == 1:
                raise EmuError("64-bit read of a single SGPR %r" % (o,))
            return self.s[o[1]] | (self.s[o[1] + 1] << 32)
        if k == "lit":
            v = o[1]
            # 32-bit inline constants / literals are sign-extended for integers in 64-bit operands
            if v > M32:
                return v & M64
            return v  # positive literal (hex masks are written in full by the assembler output)
        if k == "exec":
            return self.exec
        raise EmuError("bad 64-bit scalar source %r" % (o,))

    def ws32(self, o, val):
        k = o[0]
        val &= M32
        if k == "s":
            self.s[o[1]] = val
        elif k == "exec_lo":
            self.exec = (self.exec & ~M32 & M64) | val
        elif k == "exec_hi":
            self.exec = (self.exec & M32) | (val << 32)
        elif k == "m0":
            self.m0 = val
        elif k == "null":
            pass
        else:
            raise EmuError("bad scalar destination %r" % (o,))

    def ws64(self, o, val):
        k = o[0]
        val &= M64
        if k == "s":
            self.s[o[1]] = val & M32
            self.s[o[1] + 1] = val >> 32
        elif k == "exec":
            self.exec = val
        elif k == "null":
            pass
        else:
            raise EmuError("bad 64-bit scalar destination %r" % (o,))

    def rv32(self, o):
        """vector source -> np.uint32[64]"""
        k = o[0]
        if k == "v":
            return self.v[o[1]]
        if k == "sext":
            raise EmuError("sext() outside SDWA")
        return np.full(64, self.rs32(o), dtype=np.uint32)

    def rv64(self, o):
        if o[0] == "v":
            return self.v[o[1]].astype(np.uint64) | (self.v[o[1] + 1].astype(np.uint64) << np.uint64(32))
        return np.full(64, self.rs64(o) if (o[0] != "lit") else (o[1] if o[1] <= M32 else o[1]), dtype=np.uint64)

    def wv32(self, o, val, mask=None):
        if o[0] != "v":
            raise EmuError("bad vector destination %r" % (o,))
        e = self.em(mask)
        self.v[o[1]] = np.where(e, val.astype(np.uint32) if isinstance(val, np.ndarray) else np.uint32(val & M32),
                                self.v[o[1]])

    def wv64(self, o, val):
        e = self.em()
        val = val.astype(np.uint64)
        self.v[o[1]] = np.where(e, (val & np.uint64(M32)).astype(np.uint32), self.v[o[1]])
        self.v[o[1] + 1] = np.where(e, (val >> np.uint64(32)).astype(np.uint32), self.v[o[1] + 1])

    def wmask(self, o, boolarr):
        """result of a vector compare: only active lanes can set bits, inactive lanes read as 0"""
        m = self.mask_of(boolarr & self.em())
        self.ws64(o, m)


def _i32(a):
    return a.astype(np.int32)


def _sdwa_sel(arr, sel, sext=False):
    if sel is None or sel == "DWORD":
        return arr
    if sel.startswith("BYTE_"):
        r = (arr >> np.uint32(8 * int(sel[5]))) & np.uint32(0xFF)
        if sext:
            r = r.astype(np.uint8).astype(np.int8).astype(np.int32).astype(np.uint32)
        return r
    if sel.startswith("WORD_"):
        r = (arr >> np.uint32(16 * int(sel[5]))) & np.uint32(0xFFFF)
        if sext:
            r = r.astype(np.uint16).astype(np.int16).astype(np.int32).astype(np.uint32)
        return r
    raise EmuError("sdwa sel " + sel)


def _rv_sdwa(w, o, sel):
    """SDWA source: the selected byte / word of the register, sign-extended if written sext(vN)"""
    if o[0] == "sext":
        return _sdwa_sel(w.rv32(o[1]), sel, sext=True)
    return _sdwa_sel(w.rv32(o), sel)


def _sdwa_dst(old, new, sel, unused):
    if sel is None or sel == "DWORD":
        return new
    if sel.startswith("BYTE_"):
        sh, msk = 8 * int(sel[5]), 0xFF
    else:
        sh, msk = 16 * int(sel[5]), 0xFFFF
    field = (new & np.uint32(msk)) << np.uint32(sh)
    if unused == "UNUSED_PRESERVE":
        return (old & np.uint32(~(msk << sh) & M32)) | field
    if unused == "UNUSED_SEXT":
        raise EmuError("UNUSED_SEXT not modelled")
    return field  # UNUSED_PAD


def _dpp_src(w, inst, arr, old):
    """apply the DPP lane permutation of `inst` to arr; returns (values, valid lanes)"""
    mods = inst.mods
    lane = np.arange(64)
    src = lane.copy()
    valid = np.ones(64, dtype=bool)
    row = lane & ~15
    if "row_shr" in mods:
        n = int(mods["row_shr"])
        src = lane - n
        valid = (lane & 15) >= n
    elif "row_shl" in mods:
        n = int(mods["row_shl"])
        src = lane + n
        valid = (lane & 15) + n < 16
    elif "row_ror" in mods:
        n = int(mods["row_ror"])
        src = row | ((lane - n) & 15)
    elif "wave_shr" in mods:
        src = lane - 1
        valid = lane >= 1
    elif "wave_shl" in mods:
        src = lane + 1
        valid = lane < 63
    elif "wave_ror" in mods:
        src = (lane - 1) & 63
    elif "wave_rol" in mods:
        src = (lane + 1) & 63
    elif "row_mirror" in mods:
        src = row | (15 - (lane & 15))
    elif "row_half_mirror" in mods:
        src = (lane & ~7) | (7 - (lane & 7))
    elif "quad_perm" in mods:
        q = [int(x) for x in mods["quad_perm"].strip("[]").split(",")]
        src = (lane & ~3) | np.array(q)[lane & 3]
    elif "row_bcast" in mods:
        n = int(mods["row_bcast"])
        if n == 15:
            src = (lane & ~15) - 1
            valid = lane >= 16
        else:
            src = (lane & ~31) - 1
            valid = lane >= 32
    else:
        raise EmuError("dpp control not modelled: " + inst.text)
    srcc = np.clip(src, 0, 63)
    vals = arr[srcc]
    # a source lane that is disabled in EXEC is invalid as well
    valid = valid & w.em()[srcc]
    rm = int(mods.get("row_mask", "0xf"), 0)
    bm = int(mods.get("bank_mask", "0xf"), 0)
    wr = (((rm >> (lane >> 4)) & 1) == 1) & (((bm >> ((lane >> 2) & 3)) & 1) == 1)
    bound = "bound_ctrl" in mods
    if bound:
        vals = np.where(valid, vals, np.uint32(0))
        ok = wr
    else:
        ok = wr & valid
    return vals, ok


# ---- vector ALU tables ------------------------------------------------------------------------------------------
def _shl(a, n):
    return (a.astype(np.uint64) << (n & np.uint32(31)).astype(np.uint64)).astype(np.uint32)


V2 = {  # name -> f(src0, src1) on uint32 arrays
    "v_add_u32": lambda a, b: a + b,
    "v_sub_u32": lambda a, b: a - b,
    "v_subrev_u32": lambda a, b: b - a,
    "v_and_b32": lambda a, b: a & b,
    "v_or_b32": lambda a, b: a | b,
    "v_xor_b32": lambda a, b: a ^ b,
    "v_xnor_b32": lambda a, b: ~(a ^ b),
    "v_lshlrev_b32": lambda a, b: _shl(b, a),
    "v_bfm_b32": lambda a, b: _shl((np.uint64(1) << (a & np.uint32(31)).astype(np.uint64)).astype(np.uint32) - np.uint32(1), b),
    "v_lshrrev_b32": lambda a, b: b >> (a & np.uint32(31)),
    "v_ashrrev_i32": lambda a, b: (_i32(b) >> (a & np.uint32(31)).astype(np.int32)).astype(np.uint32),
    "v_min_u32": np.minimum,
    "v_max_u32": np.maximum,
    "v_min_i32": lambda a, b: np.minimum(_i32(a), _i32(b)).astype(np.uint32),
    "v_max_i32": lambda a, b: np.maximum(_i32(a), _i32(b)).astype(np.uint32),
    "v_mul_lo_u32": lambda a, b: (a.astype(np.uint64) * b.astype(np.uint64)).astype(np.uint32),
    "v_mul_hi_u32": lambda a, b: ((a.astype(np.uint64) * b.astype(np.uint64)) >> np.uint64(32)).astype(np.uint32),
    "v_mul_u32_u24": lambda a, b: ((a & np.uint32(0xFFFFFF)).astype(np.uint64) * (b & np.uint32(0xFFFFFF)).astype(np.uint64)).astype(np.uint32),
    "v_mul_i32_i24": lambda a, b: (_sx24(a) * _sx24(b)).astype(np.uint64).astype(np.uint32),
    "v_mul_hi_i32_i24": lambda a, b: ((_sx24(a) * _sx24(b)) >> np.int64(32)).astype(np.uint64).astype(np.uint32),
    "v_mul_hi_u32_u24": lambda a, b: (((a & np.uint32(0xFFFFFF)).astype(np.uint64) * (b & np.uint32(0xFFFFFF)).astype(np.uint64)) >> np.uint64(32)).astype(np.uint32),
    "v_lshlrev_b16": lambda a, b: (b << (a & np.uint32(15))) & np.uint32(0xFFFF),
    "v_lshrrev_b16": lambda a, b: (b & np.uint32(0xFFFF)) >> (a & np.uint32(15)),
    "v_add_u16": lambda a, b: (a + b) & np.uint32(0xFFFF),
    "v_sub_u16": lambda a, b: (a - b) & np.uint32(0xFFFF),
    "v_and_b16": lambda a, b: (a & b) & np.uint32(0xFFFF),
    "v_or_b16": lambda a, b: (a | b) & np.uint32(0xFFFF),
    "v_bcnt_u32_b32": lambda a, b: np.array([bin(int(x)).count("1") for x in a], dtype=np.uint32) + b,
}


def _sx24(a):
    """low 24 bits, sign-extended, as int64"""
    v = (a & np.uint32(0xFFFFFF)).astype(np.int64)
    return np.where(v & 0x800000, v - 0x1000000, v)


def _ffbl(a):
    out = np.full(64, M32, dtype=np.uint32)
    nz = a != 0
    low = a & (~a + np.uint32(1))
    out[nz] = np.log2(low[nz].astype(np.float64)).astype(np.uint32)
    return out


def _ffbh(a):
    out = np.full(64, M32, dtype=np.uint32)
    nz = a != 0
    bl = np.zeros(64, dtype=np.uint32)
    x = a.astype(np.uint64)
    bl[nz] = np.floor(np.log2(x[nz].astype(np.float64))).astype(np.uint32)
    # float64 represents every uint32 exactly, so floor(log2) is exact
    out[nz] = np.uint32(31) - bl[nz]
    return out


def _bfrev(a):
    r = np.zeros(64, dtype=np.uint32)
    x = a.copy()
    for _ in range(32):
        r = (r << np.uint32(1)) | (x & np.uint32(1))
        x = x >> np.uint32(1)
    return r


V1 = {
    "v_mov_b32": lambda a: a,
    "v_not_b32": lambda a: ~a,
    "v_ffbl_b32": _ffbl,
    "v_ffbh_u32": _ffbh,
    "v_bfrev_b32": _bfrev,
    "v_mov_b16": lambda a: a & np.uint32(0xFFFF),
}


def _alignbyte(a, b, c):
    sh = ((c & np.uint32(3)) * np.uint32(8)).astype(np.uint64)
    return (((a.astype(np.uint64) << np.uint64(32)) | b.astype(np.uint64)) >> sh).astype(np.uint32)


def _alignbit(a, b, c):
    sh = (c & np.uint32(31)).astype(np.uint64)
    return (((a.astype(np.uint64) << np.uint64(32)) | b.astype(np.uint64)) >> sh).astype(np.uint32)


def _v_bfe_u32(a, b, c):
    off = (b & np.uint32(31)).astype(np.uint64)
    width = (c & np.uint32(31)).astype(np.uint64)
    return ((a.astype(np.uint64) >> off) & ((np.uint64(1) << width) - np.uint64(1))).astype(np.uint32)


def _perm(a, b, sel):
    src = (a.astype(np.uint64) << np.uint64(32)) | b.astype(np.uint64)
    out = np.zeros(64, dtype=np.uint32)
    for i in range(4):
        s = (sel >> np.uint32(8 * i)) & np.uint32(0xFF)
        byte = np.zeros(64, dtype=np.uint32)
        lo = s < 8
        byte[lo] = ((src[lo] >> (s[lo].astype(np.uint64) * np.uint64(8))) & np.uint64(0xFF)).astype(np.uint32)
        byte[s == 0x0C] = 0
        byte[s >= 0x0D] = 0xFF
        for k, bit in ((8, 15), (9, 31), (10, 47), (11, 63)):
            m = s == k
            byte[m] = np.where((src[m] >> np.uint64(bit)) & np.uint64(1), 0xFF, 0).astype(np.uint32)
        out |= byte << np.uint32(8 * i)
    return out


def _dot4_u32_u8(a, b, c):
    r = c.copy()
    for k in range(4):
        r = r + ((a >> np.uint32(8 * k)) & np.uint32(0xFF)) * ((b >> np.uint32(8 * k)) & np.uint32(0xFF))
    return r


V3 = {
    "v_dot4_u32_u8": _dot4_u32_u8,
    "v_lshl_add_u32": lambda a, b, c: _shl(a, b) + c,
    "v_add_lshl_u32": lambda a, b, c: _shl(a + b, c),
    "v_lshl_or_b32": lambda a, b, c: _shl(a, b) | c,
    "v_and_or_b32": lambda a, b, c: (a & b) | c,
    "v_or3_b32": lambda a, b, c: a | b | c,
    "v_add3_u32": lambda a, b, c: a + b + c,
    "v_xad_u32": lambda a, b, c: (a ^ b) + c,
    "v_bfi_b32": lambda a, b, c: (a & b) | (~a & c),
    "v_alignbyte_b32": _alignbyte,
    "v_alignbit_b32": _alignbit,
    "v_bfe_u32": _v_bfe_u32,
    "v_perm_b32": _perm,
    "v_mad_u32_u24": lambda a, b, c: ((a & np.uint32(0xFFFFFF)).astype(np.uint64) * (b & np.uint32(0xFFFFFF)).astype(np.uint64)).astype(np.uint32) + c,
    "v_med3_i32": lambda a, b, c: np.sort(np.stack([_i32(a), _i32(b), _i32(c)]), axis=0)[1].astype(np.uint32),
    "v_med3_u32": lambda a, b, c: np.sort(np.stack([a, b, c]), axis=0)[1],
    "v_min3_u32": lambda a, b, c: np.minimum(np.minimum(a, b), c),
    "v_max3_u32": lambda a, b, c: np.maximum(np.maximum(a, b), c),
    "v_min3_i32": lambda a, b, c: np.minimum(np.minimum(_i32(a), _i32(b)), _i32(c)).astype(np.uint32),
    "v_max3_i32": lambda a, b, c: np.maximum(np.maximum(_i32(a), _i32(b)), _i32(c)).astype(np.uint32),
    "v_mad_u32_u16": lambda a, b, c: (a & np.uint32(0xFFFF)) * (b & np.uint32(0xFFFF)) + c,
}

_CMP = {"eq": np.equal, "ne": np.not_equal, "lg": np.not_equal, "gt": np.greater, "ge": np.greater_equal,
        "lt": np.less, "le": np.less_equal}
_SCMP = {"eq": lambda a, b: a == b, "lg": lambda a, b: a != b, "gt": lambda a, b: a > b, "ge": lambda a, b: a >= b,
         "lt": lambda a, b: a < b, "le": lambda a, b: a <= b}

_S2_32 = {
    "s_and_b32": lambda a, b: a & b, "s_or_b32": lambda a, b: a | b, "s_xor_b32": lambda a, b: a ^ b,
    "s_andn2_b32": lambda a, b: a & ~b, "s_orn2_b32": lambda a, b: a | (~b & M32),
    "s_nand_b32": lambda a, b: ~(a & b), "s_nor_b32": lambda a, b: ~(a | b), "s_xnor_b32": lambda a, b: ~(a ^ b),
    "s_lshl_b32": lambda a, b: a << (b & 31), "s_lshr_b32": lambda a, b: a >> (b & 31),
    "s_ashr_i32": lambda a, b: s32(a) >> (b & 31),
}
_S2_64 = {
    "s_and_b64": lambda a, b: a & b, "s_or_b64": lambda a, b: a | b, "s_xor_b64": lambda a, b: a ^ b,
    "s_andn2_b64": lambda a, b: a & ~b, "s_orn2_b64": lambda a, b: a | (~b & M64),
    "s_nand_b64": lambda a, b: ~(a & b), "s_nor_b64": lambda a, b: ~(a | b), "s_xnor_b64": lambda a, b: ~(a ^ b),
}


def _lit64(o, w):
    """64-bit scalar source where small negative inline constants are sign-extended"""
    if o[0] == "lit":
        v = o[1]
        if v > M32:  # parsed from a negative number (already masked to 64 bits) or a long hex
            return v & M64
        return v
    return w.rs64(o)


CODE_BASE = 0x7C0DE000_0000  # where the interpreter pretends the code lives: instruction k is at CODE_BASE + 4 k


class Program:
    def __init__(self, text, entry=None, callees=()):
        """callees: device functions the kernel CALLS (s_getpc_b64 + sym@rel32 + s_swappc_b64; hipcc keeps a rare path out
        of line): their instructions are appended behind the kernel's, their names become code addresses (code_symbols)."""
        self.insts, self.labels = parse_program(text, entry)
        self.code_symbols = {}
        for name in callees:
            more, labels = parse_program(text, name)
            base = len(self.insts)
            self.insts += more
            for k, v in labels.items():
                self.labels[k] = base + v
            self.code_symbols[name] = CODE_BASE + 4 * self.labels[name]
        for i in self.insts:
            self._bind(i)

    def _bind(self, i):
        op = i.op
        base = op
        for suf in ("_e32", "_e64", "_sdwa", "_dpp"):
            if base.endswith(suf):
                base = base[: -len(suf)]
                break
        fn = getattr(self, "x_" + base, None)
        if fn is None:
            fn = self._generic(i, base)
        if fn is None:
            i.fn = None  # raised when executed (unknown opcodes on paths that never run are harmless)
        else:
            i.fn = fn
        if op.startswith("s_cbranch") or op == "s_branch":
            i.target = i.ops[0][1]
        c = op[0]
        if op.startswith("ds_"):
            i.kind = "lds"
        elif op.startswith(("global_", "flat_", "buffer_", "scratch_")):
            i.kind = "vmem"
        elif op.startswith(("s_load", "s_buffer_load", "s_memtime", "s_memrealtime")):
            i.kind = "smem"
        elif c == "v":
            i.kind = "valu"
        else:
            i.kind = "salu"

    # ---- generic families ------------------------------------------------------------------------------------
    def _generic(self, i, base):
        if base in _S2_32:
            f = _S2_32[base]

            def run(w, i, f=f):
                r = f(w.rs32(i.ops[1]), w.rs32(i.ops[2])) & M32
                w.ws32(i.ops[0], r)
                w.scc = 1 if r else 0
            return run
        if base in _S2_64:
            f = _S2_64[base]

            def run(w, i, f=f):
                r = f(_lit64(i.ops[1], w), _lit64(i.ops[2], w)) & M64
                w.ws64(i.ops[0], r)
                w.scc = 1 if r else 0
            return run
        m = re.match(r"^s_cmp_(eq|lg|gt|ge|lt|le)_(i32|u32)$", base)
        if m:
            f, signed = _SCMP[m.group(1)], m.group(2) == "i32"

            def run(w, i, f=f, signed=signed):
                a, b = w.rs32(i.ops[0]), w.rs32(i.ops[1])
                if signed:
                    a, b = s32(a), s32(b)
                w.scc = 1 if f(a, b) else 0
            return run
        m = re.match(r"^s_cmpk_(eq|lg|gt|ge|lt|le)_(i32|u32)$", base)
        if m:
            f, signed = _SCMP[m.group(1)], m.group(2) == "i32"

            def run(w, i, f=f, signed=signed):
                a, b = w.rs32(i.ops[0]), i.ops[1][1] & 0xFFFF
                if signed:
                    a = s32(a)
                    b = b - 0x10000 if b & 0x8000 else b
                w.scc = 1 if f(a, b) else 0
            return run
        m = re.match(r"^s_cmp_(eq|lg)_u64$", base)
        if m:
            f = _SCMP[m.group(1)]

            def run(w, i, f=f):
                w.scc = 1 if f(_lit64(i.ops[0], w), _lit64(i.ops[1], w)) else 0
            return run
        m = re.match(r"^v_cmpx_(eq|ne|lg|gt|ge|lt|le)_(u32|i32)$", base)
        if m:  # (gfx9: the compare's result goes to the SGPR pair / VCC AND to EXEC; inactive lanes read as 0)
            f, signed = _CMP[m.group(1)], m.group(2) == "i32"

            def run(w, i, f=f, signed=signed):
                x, y = w.rv32(i.ops[1]), w.rv32(i.ops[2])
                if signed:
                    x, y = _i32(x), _i32(y)
                w.wmask(i.ops[0], f(x, y))
                w.exec = w.rs64(i.ops[0])
            return run
        m = re.match(r"^v_cmp_(eq|ne|lg|gt|ge|lt|le)_(u32|i32|u16|i16|u64|i64)$", base)
        if m:
            f, ty = _CMP[m.group(1)], m.group(2)
            e32 = i.op.endswith("_e32") or (len(i.ops) == 3 and i.ops[0] == ("s", VCC, 2) and not i.op.endswith("_e64"))

            def run(w, i, f=f, ty=ty):
                dst, a, b = i.ops[0], i.ops[1], i.ops[2]
                if ty in ("u64", "i64"):
                    x, y = w.rv64(a), w.rv64(b)
                    if ty == "i64":
                        x, y = x.astype(np.int64), y.astype(np.int64)
                else:
                    if "src0_sel" in i.mods or "src1_sel" in i.mods:
                        x = _rv_sdwa(w, a, i.mods.get("src0_sel"))
                        y = _rv_sdwa(w, b, i.mods.get("src1_sel"))
                    else:
                        x, y = w.rv32(a), w.rv32(b)
                    if ty == "i32":
                        x, y = _i32(x), _i32(y)
                    elif ty == "u16":
                        x, y = x & np.uint32(0xFFFF), y & np.uint32(0xFFFF)
                    elif ty == "i16":
                        x, y = x.astype(np.uint16).astype(np.int16), y.astype(np.uint16).astype(np.int16)
                w.wmask(dst, f(x, y))
            return run
        if base in V2:
            f = V2[base]

            def run(w, i, f=f):
                if i.op.endswith("_sdwa"):
                    a = _rv_sdwa(w, i.ops[1], i.mods.get("src0_sel"))
                    b = _rv_sdwa(w, i.ops[2], i.mods.get("src1_sel"))
                    r = _sdwa_dst(w.v[i.ops[0][1]], f(a, b), i.mods.get("dst_sel"), i.mods.get("dst_unused"))
                    w.wv32(i.ops[0], r)
                    return
                a, b = w.rv32(i.ops[1]), w.rv32(i.ops[2])
                if i.op.endswith("_dpp"):
                    a, ok = _dpp_src(w, i, a, None)
                    r = f(a, b)
                    w.wv32(i.ops[0], np.where(ok, r, w.v[i.ops[0][1]]))
                elif i.mods.get("clamp"):
                    # integer clamp: the result saturates instead of wrapping (hipcc: `cond ? 0 : x - 4` -> v_sub_u32 ... clamp)
                    if base == "v_sub_u32":
                        w.wv32(i.ops[0], np.where(a >= b, a - b, np.uint32(0)))
                    elif base == "v_subrev_u32":
                        w.wv32(i.ops[0], np.where(b >= a, b - a, np.uint32(0)))
                    elif base == "v_add_u32":
                        r = a.astype(np.uint64) + b.astype(np.uint64)
                        w.wv32(i.ops[0], np.minimum(r, np.uint64(0xFFFFFFFF)).astype(np.uint32))
                    else:
                        raise EmuError("clamp not modelled for " + base)
                else:
                    w.wv32(i.ops[0], f(a, b))
            return run
        if base in V1:
            f = V1[base]

            def run(w, i, f=f):
                a = w.rv32(i.ops[1])
                if i.op.endswith("_sdwa"):
                    a = _sdwa_sel(a, i.mods.get("src0_sel"))
                    r = _sdwa_dst(w.v[i.ops[0][1]], f(a), i.mods.get("dst_sel"), i.mods.get("dst_unused"))
                    w.wv32(i.ops[0], r)
                elif i.op.endswith("_dpp"):
                    a, ok = _dpp_src(w, i, a, None)
                    w.wv32(i.ops[0], np.where(ok, f(a), w.v[i.ops[0][1]]))
                else:
                    w.wv32(i.ops[0], f(a))
            return run
        if base in V3:
            f = V3[base]

            def run(w, i, f=f):
                w.wv32(i.ops[0], f(w.rv32(i.ops[1]), w.rv32(i.ops[2]), w.rv32(i.ops[3])))
            return run
        m = re.match(r"^(global|flat)_load_(ubyte|sbyte|ushort|sshort|dword|dwordx2|dwordx3|dwordx4|ubyte_d16|ubyte_d16_hi|short_d16|short_d16_hi)$", base)
        if m:
            return self._mk_gload(m.group(2))
        m = re.match(r"^(global|flat)_store_(byte|short|dword|dwordx2|dwordx3|dwordx4)$", base)
        if m:
            return self._mk_gstore(m.group(2))
        m = re.match(r"^ds_read_(u8|i8|u16|i16|b32|b64|b96|b128)$", base)
        if m:
            return self._mk_dsread(m.group(1))
        m = re.match(r"^ds_write_(b8|b16|b32|b64|b96|b128)$", base)
        if m:
            return self._mk_dswrite(m.group(1))
        m = re.match(r"^scratch_(load|store)_(dword|dwordx2|dwordx3|dwordx4)$", base)
        if m:
            return self._mk_scratch(m.group(1) == "load", {"dword": 1, "dwordx2": 2, "dwordx3": 3, "dwordx4": 4}[m.group(2)])
        m = re.match(r"^ds_(read|write)2(st64)?_(b32|b64)$", base)
        if m:
            return self._mk_ds2(m.group(1) == "read", 4 if m.group(3) == "b32" else 8, 64 if m.group(2) else 1)
        m = re.match(r"^s_load_dword(x2|x4|x8|x16)?$", base)
        if m:
            n = {None: 1, "x2": 2, "x4": 4, "x8": 8, "x16": 16}[m.group(1)]

            def run(w, i, n=n):
                off = w.rs32(i.ops[2]) if i.ops[2][0] != "lit" else i.ops[2][1]
                if "offset" in i.mods:
                    off += int(i.mods["offset"], 0)
                raw = w.mem.load_scalar(w.rs64(i.ops[1]) + off, 4 * n)
                for k in range(n):
                    w.s[i.ops[0][1] + k] = int.from_bytes(raw[4 * k:4 * k + 4], "little")
            return run
        return None

    # ---- addresses
    @staticmethod
    def _gaddr(w, i, vaddr_op, saddr_op):
        off = int(i.mods.get("offset", "0"), 0)
        if off & 0x1000 and off < 0x2000:  # 13-bit signed, printed as positive by some tools (never by llvm)
            pass
        if saddr_op[0] == "off":
            a = w.rv64(vaddr_op)
        else:
            a = np.uint64(w.rs64(saddr_op)) + w.v[vaddr_op[1]].astype(np.uint64)
        return (a.astype(np.int64) + np.int64(s32(off) if off > 0x7FFFFFFF else off)).astype(np.uint64)

    def _mk_gload(self, ty):
        n = {"ubyte": 1, "sbyte": 1, "ushort": 2, "sshort": 2, "dword": 4, "dwordx2": 8, "dwordx3": 12, "dwordx4": 16,
             "ubyte_d16": 1, "ubyte_d16_hi": 1, "short_d16": 2, "short_d16_hi": 2}[ty]

        def run(w, i, n=n, ty=ty):
            addrs = self._gaddr(w, i, i.ops[1], i.ops[2] if len(i.ops) > 2 else ("off",))  # (flat_load: no saddr)
            e = w.em()
            raw = w.mem.load(addrs, e, n)
            d = i.ops[0][1]
            if n >= 4:
                words = raw.view("<u4")
                for k in range(n // 4):
                    w.v[d + k] = np.where(e, words[:, k], w.v[d + k])
                return
            if n == 1:
                val = raw[:, 0].astype(np.uint32)
                if ty == "sbyte":
                    val = raw[:, 0].astype(np.int8).astype(np.int32).astype(np.uint32)
            else:
                val = raw.view("<u2")[:, 0].astype(np.uint32)
                if ty == "sshort":
                    val = raw.view("<i2")[:, 0].astype(np.int32).astype(np.uint32)
            if ty.endswith("d16"):
                val = (w.v[d] & np.uint32(0xFFFF0000)) | val
            elif ty.endswith("d16_hi"):
                val = (w.v[d] & np.uint32(0xFFFF)) | (val << np.uint32(16))
            w.v[d] = np.where(e, val, w.v[d])
        return run

    def _mk_gstore(self, ty):
        n = {"byte": 1, "short": 2, "dword": 4, "dwordx2": 8, "dwordx3": 12, "dwordx4": 16}[ty]

        def run(w, i, n=n):
            addrs = self._gaddr(w, i, i.ops[0], i.ops[2] if len(i.ops) > 2 else ("off",))  # (flat_store: no saddr)
            e = w.em()
            d = i.ops[1][1]
            if n >= 4:
                data = np.stack([w.v[d + k] for k in range(n // 4)], axis=1).astype("<u4").view(np.uint8).reshape(64, n)
            else:
                data = w.v[d].astype("<u4").view(np.uint8).reshape(64, 4)[:, :n]
            w.mem.store(addrs, e, np.ascontiguousarray(data))
        return run

    def _mk_dsread(self, ty):
        n = {"u8": 1, "i8": 1, "u16": 2, "i16": 2, "b32": 4, "b64": 8, "b96": 12, "b128": 16}[ty]

        def run(w, i, n=n, ty=ty):
            e = w.em()
            addr = (w.v[i.ops[1][1]].astype(np.int64) + int(i.mods.get("offset", "0"), 0)) & 0xFFFFFFFF  # (32-bit add)
            a = np.where(e, addr, 0)
            if e.any() and (a[e].min() < 0 or a[e].max() + n > w.lds_limit):
                raise MemFault("LDS read out of bounds: %s" % i.text)
            raw = w.lds[a[:, None] + np.arange(n)]
            d = i.ops[0][1]
            if n >= 4:
                words = np.ascontiguousarray(raw).view("<u4")
                for k in range(n // 4):
                    w.v[d + k] = np.where(e, words[:, k], w.v[d + k])
                return
            if n == 1:
                val = raw[:, 0].astype(np.uint32) if ty == "u8" else raw[:, 0].astype(np.int8).astype(np.int32).astype(np.uint32)
            else:
                c = np.ascontiguousarray(raw)
                val = c.view("<u2")[:, 0].astype(np.uint32) if ty == "u16" else c.view("<i2")[:, 0].astype(np.int32).astype(np.uint32)
            w.v[d] = np.where(e, val, w.v[d])
        return run

    def _mk_scratch(self, is_load, k):
        """scratch_load_dword[xN] vdst, vaddr|off, saddr|off / scratch_store_dword[xN] vaddr|off, vdata, saddr|off: private
        memory, one copy per lane (4 KiB each here; the kernels of this repo use a few dwords)"""
        def run(w, i, is_load=is_load, k=k):
            if not hasattr(w, "scratch"):
                w.scratch = np.zeros((64, 4096), dtype=np.uint8)
            va, sa = (i.ops[1], i.ops[2]) if is_load else (i.ops[0], i.ops[2])
            addr = np.zeros(64, dtype=np.int64) + int(i.mods.get("offset", "0"), 0)
            if va[0] != "off":
                addr = addr + w.v[va[1]].astype(np.int64)
            if sa[0] != "off":
                addr = addr + w.rs32(sa)
            e = w.em()
            if e.any() and (addr[e].min() < 0 or addr[e].max() + 4 * k > w.scratch.shape[1]):
                raise MemFault("scratch access out of bounds: %s" % i.text)
            for lane in np.flatnonzero(e):
                a = int(addr[lane])
                if is_load:
                    words = np.ascontiguousarray(w.scratch[lane, a:a + 4 * k]).view("<u4")
                    for j in range(k):
                        w.v[i.ops[0][1] + j][lane] = words[j]
                else:
                    d = i.ops[1][1]
                    vals = np.array([w.v[d + j][lane] for j in range(k)], dtype="<u4")
                    w.scratch[lane, a:a + 4 * k] = vals.view(np.uint8)
        return run

    def _mk_ds2(self, is_read, n, stride=1):
        """ds_read2[st64]_b32 / _b64, ds_write2[st64]_b32 / _b64: two elements of n bytes at addr + offset0 * unit and
        addr + offset1 * unit, unit = n (st64: 64 * n)"""
        def run(w, i, is_read=is_read, n=n, unit=n * stride):
            e = w.em()
            base = w.v[i.ops[1 if is_read else 0][1]].astype(np.int64)
            k = n // 4
            for which, mod in enumerate(("offset0", "offset1")):
                addr = (base + int(i.mods.get(mod, "0"), 0) * unit) & 0xFFFFFFFF  # (32-bit add: a negative base wraps)
                a = np.where(e, addr, 0)
                if e.any() and (a[e].min() < 0 or a[e].max() + n > w.lds_limit):
                    raise MemFault("LDS access out of bounds: %s" % i.text)
                if is_read:
                    raw = np.ascontiguousarray(w.lds[a[:, None] + np.arange(n)]).view("<u4")
                    d = i.ops[0][1] + which * k
                    for j in range(k):
                        w.v[d + j] = np.where(e, raw[:, j], w.v[d + j])
                else:
                    d = i.ops[1 + which][1]
                    data = np.stack([w.v[d + j] for j in range(k)], axis=1).astype("<u4").view(np.uint8).reshape(64, n)
                    for lane in np.flatnonzero(e):
                        al = int(addr[lane])
                        w.lds[al:al + n] = data[lane]
        return run

    def _mk_dswrite(self, ty):
        n = {"b8": 1, "b16": 2, "b32": 4, "b64": 8, "b96": 12, "b128": 16}[ty]

        def run(w, i, n=n):
            e = w.em()
            addr = (w.v[i.ops[0][1]].astype(np.int64) + int(i.mods.get("offset", "0"), 0)) & 0xFFFFFFFF
            d = i.ops[1][1]
            if n >= 4:
                data = np.stack([w.v[d + k] for k in range(n // 4)], axis=1).astype("<u4").view(np.uint8).reshape(64, n)
            else:
                data = w.v[d].astype("<u4").view(np.uint8).reshape(64, 4)[:, :n]
            lanes = np.flatnonzero(e)
            if w.lds_order is not None:
                lanes = lanes[w.lds_order(len(lanes))]
            for lane in lanes:
                a = int(addr[lane])
                if a < 0 or a + n > w.lds_limit:
                    raise MemFault("LDS write out of bounds (lane %d, addr %d): %s" % (lane, a, i.text))
                w.lds[a:a + n] = data[lane]
        return run

    # ---- scalar singles ----------------------------------------------------------------------------------------
    def x_s_mov_b32(self, w, i):
        w.ws32(i.ops[0], w.rs32(i.ops[1]))

    def x_s_movk_i32(self, w, i):
        v = i.ops[1][1] & 0xFFFF
        w.ws32(i.ops[0], v - 0x10000 if v & 0x8000 else v)

    def x_s_mov_b64(self, w, i):
        w.ws64(i.ops[0], _lit64(i.ops[1], w))

    def x_s_cmov_b32(self, w, i):
        if w.scc:
            w.ws32(i.ops[0], w.rs32(i.ops[1]))

    def x_s_cmov_b64(self, w, i):
        if w.scc:
            w.ws64(i.ops[0], _lit64(i.ops[1], w))

    def x_s_not_b32(self, w, i):
        r = ~w.rs32(i.ops[1]) & M32
        w.ws32(i.ops[0], r)
        w.scc = 1 if r else 0

    def x_s_not_b64(self, w, i):
        r = ~_lit64(i.ops[1], w) & M64
        w.ws64(i.ops[0], r)
        w.scc = 1 if r else 0

    def x_s_add_i32(self, w, i):
        a, b = s32(w.rs32(i.ops[1])), s32(w.rs32(i.ops[2]))
        r = a + b
        w.scc = 1 if (r > 0x7FFFFFFF or r < -0x80000000) else 0
        w.ws32(i.ops[0], r)

    def x_s_sub_i32(self, w, i):
        a, b = s32(w.rs32(i.ops[1])), s32(w.rs32(i.ops[2]))
        r = a - b
        w.scc = 1 if (r > 0x7FFFFFFF or r < -0x80000000) else 0
        w.ws32(i.ops[0], r)

    def x_s_add_u32(self, w, i):
        r = w.rs32(i.ops[1]) + w.rs32(i.ops[2])
        w.scc = 1 if r > M32 else 0
        w.ws32(i.ops[0], r)

    def x_s_addc_u32(self, w, i):
        r = w.rs32(i.ops[1]) + w.rs32(i.ops[2]) + w.scc
        w.scc = 1 if r > M32 else 0
        w.ws32(i.ops[0], r)

    def x_s_sub_u32(self, w, i):
        a, b = w.rs32(i.ops[1]), w.rs32(i.ops[2])
        w.scc = 1 if b > a else 0
        w.ws32(i.ops[0], a - b)

    def x_s_subb_u32(self, w, i):
        a, b = w.rs32(i.ops[1]), w.rs32(i.ops[2]) + w.scc
        w.ws32(i.ops[0], a - b)
        w.scc = 1 if b > a else 0

    def x_s_addk_i32(self, w, i):
        k = i.ops[1][1] & 0xFFFF
        k = k - 0x10000 if k & 0x8000 else k
        r = s32(w.rs32(i.ops[0])) + k
        w.scc = 1 if (r > 0x7FFFFFFF or r < -0x80000000) else 0
        w.ws32(i.ops[0], r)

    def x_s_mulk_i32(self, w, i):
        k = i.ops[1][1] & 0xFFFF
        k = k - 0x10000 if k & 0x8000 else k
        w.ws32(i.ops[0], s32(w.rs32(i.ops[0])) * k)

    def x_s_mul_i32(self, w, i):
        w.ws32(i.ops[0], s32(w.rs32(i.ops[1])) * s32(w.rs32(i.ops[2])))

    def x_s_mul_hi_u32(self, w, i):
        w.ws32(i.ops[0], (w.rs32(i.ops[1]) * w.rs32(i.ops[2])) >> 32)

    def x_s_mul_hi_i32(self, w, i):
        w.ws32(i.ops[0], (s32(w.rs32(i.ops[1])) * s32(w.rs32(i.ops[2]))) >> 32)

    def x_s_min_i32(self, w, i):
        a, b = s32(w.rs32(i.ops[1])), s32(w.rs32(i.ops[2]))
        w.scc = 1 if a < b else 0
        w.ws32(i.ops[0], min(a, b))

    def x_s_max_i32(self, w, i):
        a, b = s32(w.rs32(i.ops[1])), s32(w.rs32(i.ops[2]))
        w.scc = 1 if a > b else 0
        w.ws32(i.ops[0], max(a, b))

    def x_s_min_u32(self, w, i):
        a, b = w.rs32(i.ops[1]), w.rs32(i.ops[2])
        w.scc = 1 if a < b else 0
        w.ws32(i.ops[0], min(a, b))

    def x_s_max_u32(self, w, i):
        a, b = w.rs32(i.ops[1]), w.rs32(i.ops[2])
        w.scc = 1 if a > b else 0
        w.ws32(i.ops[0], max(a, b))

    def x_s_cselect_b32(self, w, i):
        w.ws32(i.ops[0], w.rs32(i.ops[1]) if w.scc else w.rs32(i.ops[2]))

    def x_s_cselect_b64(self, w, i):
        w.ws64(i.ops[0], _lit64(i.ops[1], w) if w.scc else _lit64(i.ops[2], w))

    def x_s_lshl_b64(self, w, i):
        r = (_lit64(i.ops[1], w) << (w.rs32(i.ops[2]) & 63)) & M64
        w.ws64(i.ops[0], r)
        w.scc = 1 if r else 0

    def x_s_lshr_b64(self, w, i):
        r = _lit64(i.ops[1], w) >> (w.rs32(i.ops[2]) & 63)
        w.ws64(i.ops[0], r)
        w.scc = 1 if r else 0

    def x_s_ashr_i64(self, w, i):
        r = s64(_lit64(i.ops[1], w)) >> (w.rs32(i.ops[2]) & 63)
        w.ws64(i.ops[0], r)
        w.scc = 1 if (r & M64) else 0

    def x_s_bfe_u32(self, w, i):
        r = _bfe_u32(w.rs32(i.ops[1]), w.rs32(i.ops[2]))
        w.ws32(i.ops[0], r)
        w.scc = 1 if r else 0

    def _s_bfe64(self, w, i, signed):
        spec = w.rs32(i.ops[2])
        off, width = spec & 0x3F, (spec >> 16) & 0x7F
        v = _lit64(i.ops[1], w) & M64
        if width == 0:
            r = 0
        else:
            width = min(width, 64 - off) if off < 64 else 0
            r = (v >> off) & ((1 << width) - 1) if width else 0
            if signed and width and width < 64 and (r >> (width - 1)) & 1:
                r -= 1 << width
        w.ws64(i.ops[0], r & M64)
        w.scc = 1 if (r & M64) else 0

    def x_s_bfe_u64(self, w, i):
        self._s_bfe64(w, i, False)

    def x_s_bfe_i64(self, w, i):
        self._s_bfe64(w, i, True)

    def x_s_bfe_i32(self, w, i):
        spec = w.rs32(i.ops[2])
        width = (spec >> 16) & 0x7F
        r = _bfe_u32(w.rs32(i.ops[1]), spec)
        if width and width < 32 and (r >> (width - 1)) & 1:
            r -= 1 << width
        w.ws32(i.ops[0], r)
        w.scc = 1 if (r & M32) else 0

    def x_s_bfe_u64(self, w, i):
        spec = w.rs32(i.ops[2])
        off, width = spec & 63, (spec >> 16) & 0x7F
        r = (_lit64(i.ops[1], w) >> off) & ((1 << width) - 1) if width else 0
        w.ws64(i.ops[0], r)
        w.scc = 1 if r else 0

    def x_s_ff1_i32_b32(self, w, i):
        w.ws32(i.ops[0], _ff1(w.rs32(i.ops[1]), 32))

    def x_s_ff1_i32_b64(self, w, i):
        w.ws32(i.ops[0], _ff1(_lit64(i.ops[1], w), 64))

    def x_s_ff0_i32_b64(self, w, i):  # position of the first ZERO bit, -1 if there is none (SCC untouched)
        w.ws32(i.ops[0], _ff1(~_lit64(i.ops[1], w) & ((1 << 64) - 1), 64))

    def x_s_flbit_i32_b32(self, w, i):
        w.ws32(i.ops[0], _flbit(w.rs32(i.ops[1]), 32))

    def x_s_flbit_i32_b64(self, w, i):
        w.ws32(i.ops[0], _flbit(_lit64(i.ops[1], w), 64))

    def x_s_bcnt1_i32_b32(self, w, i):
        r = bin(w.rs32(i.ops[1])).count("1")
        w.ws32(i.ops[0], r)
        w.scc = 1 if r else 0

    def x_s_bcnt1_i32_b64(self, w, i):
        r = bin(_lit64(i.ops[1], w)).count("1")
        w.ws32(i.ops[0], r)
        w.scc = 1 if r else 0

    def x_s_brev_b32(self, w, i):
        w.ws32(i.ops[0], int("{:032b}".format(w.rs32(i.ops[1]))[::-1], 2))

    def x_s_bitset1_b32(self, w, i):
        w.ws32(i.ops[0], w.rs32(i.ops[0]) | (1 << (w.rs32(i.ops[1]) & 31)))

    def x_s_bitset0_b32(self, w, i):
        w.ws32(i.ops[0], w.rs32(i.ops[0]) & ~(1 << (w.rs32(i.ops[1]) & 31)))

    def x_s_bitset1_b64(self, w, i):
        w.ws64(i.ops[0], w.rs64(i.ops[0]) | (1 << (w.rs32(i.ops[1]) & 63)))

    def x_s_bitset0_b64(self, w, i):
        w.ws64(i.ops[0], w.rs64(i.ops[0]) & ~(1 << (w.rs32(i.ops[1]) & 63)))

    def x_s_bitcmp0_b32(self, w, i):
        w.scc = 1 if ((w.rs32(i.ops[0]) >> (w.rs32(i.ops[1]) & 31)) & 1) == 0 else 0

    def x_s_bitcmp1_b32(self, w, i):
        w.scc = (w.rs32(i.ops[0]) >> (w.rs32(i.ops[1]) & 31)) & 1

    def x_s_bitcmp0_b64(self, w, i):
        w.scc = 1 if ((_lit64(i.ops[0], w) >> (w.rs32(i.ops[1]) & 63)) & 1) == 0 else 0

    def x_s_bitcmp1_b64(self, w, i):
        w.scc = (_lit64(i.ops[0], w) >> (w.rs32(i.ops[1]) & 63)) & 1

    def x_s_sext_i32_i16(self, w, i):
        v = w.rs32(i.ops[1]) & 0xFFFF
        w.ws32(i.ops[0], v - 0x10000 if v & 0x8000 else v)

    def x_s_sext_i32_i8(self, w, i):
        v = w.rs32(i.ops[1]) & 0xFF
        w.ws32(i.ops[0], v - 0x100 if v & 0x80 else v)

    def x_s_abs_i32(self, w, i):
        r = abs(s32(w.rs32(i.ops[1])))
        w.ws32(i.ops[0], r)
        w.scc = 1 if (r & M32) else 0

    def x_s_pack_ll_b32_b16(self, w, i):
        w.ws32(i.ops[0], (w.rs32(i.ops[1]) & 0xFFFF) | ((w.rs32(i.ops[2]) & 0xFFFF) << 16))

    def x_s_lshl1_add_u32(self, w, i):
        self._lshl_add(w, i, 1)

    def x_s_lshl2_add_u32(self, w, i):
        self._lshl_add(w, i, 2)

    def x_s_lshl3_add_u32(self, w, i):
        self._lshl_add(w, i, 3)

    def x_s_lshl4_add_u32(self, w, i):
        self._lshl_add(w, i, 4)

    def _lshl_add(self, w, i, n):
        r = (w.rs32(i.ops[1]) << n) + w.rs32(i.ops[2])
        w.scc = 1 if r > M32 else 0
        w.ws32(i.ops[0], r)

    def _saveexec(self, w, i, f):
        old = w.exec
        src = _lit64(i.ops[1], w)  # (the destination may be the source register)
        w.ws64(i.ops[0], old)
        w.exec = f(src, old) & M64
        w.scc = 1 if w.exec else 0

    def x_s_and_saveexec_b64(self, w, i):
        self._saveexec(w, i, lambda s, e: s & e)

    def x_s_or_saveexec_b64(self, w, i):
        self._saveexec(w, i, lambda s, e: s | e)

    def x_s_xor_saveexec_b64(self, w, i):
        self._saveexec(w, i, lambda s, e: s ^ e)

    def x_s_andn2_saveexec_b64(self, w, i):
        self._saveexec(w, i, lambda s, e: s & ~e)

    def x_s_andn1_saveexec_b64(self, w, i):
        self._saveexec(w, i, lambda s, e: ~s & e)

    def x_s_orn2_saveexec_b64(self, w, i):
        self._saveexec(w, i, lambda s, e: s | ~e)

    def x_s_nop(self, w, i):
        pass

    x_s_waitcnt = x_s_nop
    def x_s_barrier(self, w, i):
        g = getattr(w, "group", None)
        if g is not None:
            g.barrier(w.wave_in_group)
        elif getattr(w, "multi_wave_group", False):
            raise EmuError("s_barrier in a multi-wavefront workgroup: the interpreter runs its wavefronts one after the other")

    def x_s_sleep(self, w, i):
        g = getattr(w, "group", None)
        if g is not None:  # a wavefront that polls for another one's progress lets the others run
            g.sleep(w.wave_in_group)

    x_s_setprio = x_s_nop
    x_s_waitcnt_vscnt = x_s_nop
    x_s_waitcnt_depctr = x_s_nop
    x_s_inst_prefetch = x_s_nop
    x_s_dcache_wb = x_s_nop
    x_buffer_wbl2 = x_s_nop
    x_buffer_inv = x_s_nop
    x_s_setreg_imm32_b32 = x_s_nop
    x_v_nop = x_s_nop

    def x_s_getpc_b64(self, w, i):
        w.ws64(i.ops[0], 0)  # symbol@rel32 operands resolve to absolute addresses here

    def x_s_memtime(self, w, i):
        w.ws64(i.ops[0], w.clock)

    x_s_memrealtime = x_s_memtime

    # ---- vector singles ----------------------------------------------------------------------------------------
    def x_v_mov_b64(self, w, i):
        w.wv64(i.ops[0], w.rv64(i.ops[1]))

    def x_v_cndmask_b32(self, w, i):
        a, b = w.rv32(i.ops[1]), w.rv32(i.ops[2])
        if i.op.endswith("_sdwa"):
            a = _sdwa_sel(a, i.mods.get("src0_sel"))
            b = _sdwa_sel(b, i.mods.get("src1_sel"))
        sel = w.em(_lit64(i.ops[3], w))
        r = np.where(sel, b, a)
        if i.op.endswith("_sdwa"):
            r = _sdwa_dst(w.v[i.ops[0][1]], r, i.mods.get("dst_sel"), i.mods.get("dst_unused"))
        w.wv32(i.ops[0], r)

    def x_s_set_gpr_idx_on(self, w, i):
        # VGPR indexing mode: until s_set_gpr_idx_off, vector instructions add the index to the VGPR number of the named
        # operands (the compiler's way to index a small array it keeps in registers)
        m = re.search(r"gpr_idx\(([A-Z0-9,]+)\)", i.text)
        w.gpr_idx = (w.rs32(i.ops[0]) & 0xFF, set(m.group(1).split(",")) if m else set())

    def x_s_set_gpr_idx_off(self, w, i):
        w.gpr_idx = None

    def x_v_bfe_i32(self, w, i):
        a, off, width = w.rv32(i.ops[1]), w.rv32(i.ops[2]) & np.uint32(31), w.rv32(i.ops[3]) & np.uint32(31)
        x = (a >> off).astype(np.int64) & ((np.int64(1) << width.astype(np.int64)) - 1)
        sign = (x >> np.maximum(width.astype(np.int64) - 1, 0)) & 1
        x = np.where((width > 0) & (sign == 1), x - (np.int64(1) << width.astype(np.int64)), x)
        w.wv32(i.ops[0], (np.where(width == 0, 0, x) & 0xFFFFFFFF).astype(np.uint32))

    @staticmethod
    def _pk_sel(i, name, default):
        """op_sel / op_sel_hi of a packed (VOP3P) instruction -> list of 0 / 1 per source"""
        v = i.mods.get(name)
        if v is None:
            return default
        return [int(x) for x in v.strip("[]").split(",")]

    def _pk_shift(self, w, i, left):
        # D.lo = S1.[lo|hi] shifted by S0.[lo|hi] & 15, D.hi likewise; op_sel picks the halves for the LOW result (default
        # low, low), op_sel_hi for the HIGH result (default high, high); an inline constant has its value in the low half
        sel_lo, sel_hi = self._pk_sel(i, "op_sel", [0, 0]), self._pk_sel(i, "op_sel_hi", [1, 1])
        a, b = w.rv32(i.ops[1]), w.rv32(i.ops[2])

        def half(x, hi):
            return (x >> np.uint32(16)) if hi else (x & np.uint32(0xFFFF))

        def sh(x, n):
            n = n & np.uint32(15)
            return ((x << n) if left else (x >> n)) & np.uint32(0xFFFF)

        lo = sh(half(b, sel_lo[1]), half(a, sel_lo[0]))
        hi = sh(half(b, sel_hi[1]), half(a, sel_hi[0]))
        w.wv32(i.ops[0], lo | (hi << np.uint32(16)))

    def x_v_pk_mov_b32(self, w, i):
        # D[0] = S0[op_sel[0]], D[1] = S1[op_sel_hi[1]] (dwords of the 64-bit sources; defaults op_sel [0,0], op_sel_hi [1,1])
        sel_lo, sel_hi = self._pk_sel(i, "op_sel", [0, 0]), self._pk_sel(i, "op_sel_hi", [1, 1])
        s0, s1 = w.rv64(i.ops[1]), w.rv64(i.ops[2])
        lo = ((s0 >> np.uint64(32 * sel_lo[0])) & np.uint64(M32)).astype(np.uint64)
        hi = ((s1 >> np.uint64(32 * sel_hi[1])) & np.uint64(M32)).astype(np.uint64)
        w.wv64(i.ops[0], lo | (hi << np.uint64(32)))

    def x_v_pk_lshlrev_b16(self, w, i):
        self._pk_shift(w, i, True)

    def x_v_pk_lshrrev_b16(self, w, i):
        self._pk_shift(w, i, False)

    def x_v_bitop3_b16(self, w, i):
        # as v_bitop3_b32 on the low halves (the high half of the destination is kept)
        tt = int(i.mods["bitop3"], 0)
        a, b, c = w.rv32(i.ops[1]), w.rv32(i.ops[2]), w.rv32(i.ops[3])
        r = np.zeros(64, dtype=np.uint32)
        for k in range(8):
            if (tt >> k) & 1:
                r |= (a if k & 4 else ~a) & (b if k & 2 else ~b) & (c if k & 1 else ~c)
        old = w.rv32(i.ops[0])
        w.wv32(i.ops[0], (old & np.uint32(0xFFFF0000)) | (r & np.uint32(0xFFFF)))

    def x_v_readlane_b32(self, w, i):
        lane = w.rs32(i.ops[2]) & 63
        w.ws32(i.ops[0], int(w.v[i.ops[1][1]][lane]))

    def x_v_readfirstlane_b32(self, w, i):
        lane = _ff1(w.exec, 64) if w.exec else 0
        w.ws32(i.ops[0], int(w.rv32(i.ops[1])[lane]))

    def x_v_writelane_b32(self, w, i):
        lane = w.rs32(i.ops[2]) & 63
        w.v[i.ops[0][1]][lane] = w.rs32(i.ops[1])

    def _carry_op(self, w, i, f):
        # dst, carry_out, src0, src1 [, carry_in]
        a, b = w.rv32(i.ops[2]).astype(np.int64), w.rv32(i.ops[3]).astype(np.int64)
        cin = w.em(_lit64(i.ops[4], w)).astype(np.int64) if len(i.ops) > 4 else 0
        r = f(a, b, cin)
        w.wmask(i.ops[1], (r > M32) | (r < 0))
        w.wv32(i.ops[0], (r & M32).astype(np.uint32))

    def x_v_add_co_u32(self, w, i):
        self._carry_op(w, i, lambda a, b, c: a + b)

    def x_v_sub_co_u32(self, w, i):
        self._carry_op(w, i, lambda a, b, c: a - b)

    def x_v_subrev_co_u32(self, w, i):
        self._carry_op(w, i, lambda a, b, c: b - a)

    def x_v_addc_co_u32(self, w, i):
        self._carry_op(w, i, lambda a, b, c: a + b + c)

    def x_v_subb_co_u32(self, w, i):
        self._carry_op(w, i, lambda a, b, c: a - b - c)

    def x_v_subbrev_co_u32(self, w, i):
        self._carry_op(w, i, lambda a, b, c: b - a - c)

    def x_v_lshl_add_u64(self, w, i):
        a, sh, c = w.rv64(i.ops[1]), w.rv32(i.ops[2]), w.rv64(i.ops[3])
        w.wv64(i.ops[0], (a << (sh & np.uint32(7)).astype(np.uint64)) + c)

    def x_v_lshlrev_b64(self, w, i):
        sh, a = w.rv32(i.ops[1]), w.rv64(i.ops[2])
        w.wv64(i.ops[0], a << (sh & np.uint32(63)).astype(np.uint64))

    def x_v_lshrrev_b64(self, w, i):
        sh, a = w.rv32(i.ops[1]), w.rv64(i.ops[2])
        w.wv64(i.ops[0], a >> (sh & np.uint32(63)).astype(np.uint64))

    def x_v_ashrrev_i64(self, w, i):
        sh, a = w.rv32(i.ops[1]), w.rv64(i.ops[2])
        w.wv64(i.ops[0], (a.astype(np.int64) >> (sh & np.uint32(63)).astype(np.int64)).astype(np.uint64))

    def x_v_mad_u64_u32(self, w, i):
        # vdst[2], sdst(carry), src0, src1, src2(64)
        a, b, c = w.rv32(i.ops[2]).astype(np.uint64), w.rv32(i.ops[3]).astype(np.uint64), w.rv64(i.ops[4])
        w.wv64(i.ops[0], a * b + c)

    def x_v_mad_i64_i32(self, w, i):
        a, b, c = _i32(w.rv32(i.ops[2])).astype(np.int64), _i32(w.rv32(i.ops[3])).astype(np.int64), w.rv64(i.ops[4]).astype(np.int64)
        w.wv64(i.ops[0], (a * b + c).astype(np.uint64))

    def x_v_mbcnt_lo_u32_b32(self, w, i):
        m = w.rs32(i.ops[1]) if i.ops[1][0] != "v" else None
        mask = np.array([(((m if m is not None else int(w.v[i.ops[1][1]][l])) & ((1 << min(l, 32)) - 1))) for l in range(64)], dtype=np.uint64)
        cnt = np.array([bin(int(x)).count("1") for x in mask], dtype=np.uint32)
        w.wv32(i.ops[0], cnt + w.rv32(i.ops[2]))

    def x_v_mbcnt_hi_u32_b32(self, w, i):
        m = w.rs32(i.ops[1]) if i.ops[1][0] != "v" else None
        cnt = np.zeros(64, dtype=np.uint32)
        for l in range(32, 64):
            mm = m if m is not None else int(w.v[i.ops[1][1]][l])
            cnt[l] = bin(mm & ((1 << (l - 32)) - 1)).count("1")
        w.wv32(i.ops[0], cnt + w.rv32(i.ops[2]))

    def x_ds_bpermute_b32(self, w, i):
        off = np.uint32(int(i.mods.get("offset", "0"), 0))  # (added to the byte address before the lane is taken)
        addr = ((w.v[i.ops[1][1]] + off) >> np.uint32(2)) & np.uint32(63)
        src = w.v[i.ops[2][1]]
        e = w.em()
        vals = np.where(e[addr], src[addr], np.uint32(0))
        w.wv32(i.ops[0], vals)

    def x_ds_permute_b32(self, w, i):
        addr = (w.v[i.ops[1][1]] >> np.uint32(2)) & np.uint32(63)
        src = w.v[i.ops[2][1]]
        out = np.zeros(64, dtype=np.uint32)
        for lane in np.flatnonzero(w.em()):
            out[addr[lane]] = src[lane]
        w.wv32(i.ops[0], out)

    def x_ds_swizzle_b32(self, w, i):
        raise EmuError("ds_swizzle not modelled")

    # ---- the few single-precision instructions of hipcc's integer-division sequences -------------------------
    @staticmethod
    def _f32(w, o):
        if o[0] == "fneg":
            return -Program._f32(w, o[1])
        if o[0] == "fabs":
            return np.abs(Program._f32(w, o[1]))
        return w.rv32(o).view(np.float32)

    def x_v_cvt_f32_u32(self, w, i):
        w.wv32(i.ops[0], w.rv32(i.ops[1]).astype(np.float32).view(np.uint32))

    def x_v_cvt_f32_i32(self, w, i):
        w.wv32(i.ops[0], w.rv32(i.ops[1]).astype(np.int32).astype(np.float32).view(np.uint32))

    def x_v_cvt_f32_ubyte0(self, w, i):
        w.wv32(i.ops[0], (w.rv32(i.ops[1]) & np.uint32(0xFF)).astype(np.float32).view(np.uint32))

    def x_v_cvt_u32_f32(self, w, i):
        f = self._f32(w, i.ops[1]).astype(np.float64)
        f = np.where(np.isnan(f), 0.0, np.clip(np.trunc(f), 0.0, 4294967295.0))
        w.wv32(i.ops[0], f.astype(np.uint64).astype(np.uint32))

    def x_v_rcp_iflag_f32(self, w, i):
        with np.errstate(divide="ignore", over="ignore", invalid="ignore"):
            w.wv32(i.ops[0], (np.float32(1.0) / self._f32(w, i.ops[1])).astype(np.float32).view(np.uint32))

    x_v_rcp_f32 = x_v_rcp_iflag_f32

    def x_v_mul_f32(self, w, i):
        w.wv32(i.ops[0], (self._f32(w, i.ops[1]) * self._f32(w, i.ops[2])).astype(np.float32).view(np.uint32))

    def x_v_trunc_f32(self, w, i):
        w.wv32(i.ops[0], np.trunc(self._f32(w, i.ops[1])).astype(np.float32).view(np.uint32))

    def x_v_fma_f32(self, w, i):
        a, b, c = (self._f32(w, o).astype(np.float64) for o in i.ops[1:4])
        w.wv32(i.ops[0], (a * b + c).astype(np.float32).view(np.uint32))  # (double product + add, rounded once)

    x_v_mad_f32 = x_v_fma_f32

    def x_v_cmp_ge_f32(self, w, i):
        w.wmask(i.ops[0], self._f32(w, i.ops[1]) >= self._f32(w, i.ops[2]))

    def x_global_atomic_swap(self, w, i):
        # (no return value used here): global_atomic_swap vaddr, vdata, saddr
        addrs = self._gaddr(w, i, i.ops[0], i.ops[2])
        data = w.v[i.ops[1][1]].astype("<u4").view(np.uint8).reshape(64, 4)
        w.mem.store(addrs, w.em(), np.ascontiguousarray(data))

    def x_flat_atomic_swap(self, w, i):
        # (no return value used here): flat_atomic_swap vaddr(64), vdata
        addrs = self._gaddr(w, i, i.ops[0], ("off",))
        data = w.v[i.ops[1][1]].astype("<u4").view(np.uint8).reshape(64, 4)
        w.mem.store(addrs, w.em(), np.ascontiguousarray(data))

    def x_global_atomic_add(self, w, i):
        # with return: global_atomic_add vdst, vaddr, vdata, saddr sc0 — lanes take their turn in lane order
        ret = len(i.ops) == 4
        vaddr, vdata, saddr = (i.ops[1], i.ops[2], i.ops[3]) if ret else (i.ops[0], i.ops[1], i.ops[2])
        addrs = self._gaddr(w, i, vaddr, saddr)
        e = w.em()
        for lane in np.flatnonzero(e):
            one = np.zeros(64, dtype=bool)
            one[lane] = True
            old = w.mem.load(addrs, one, 4).view("<u4")[lane, 0]
            new = np.zeros((64, 4), dtype=np.uint8)
            new[lane] = np.array([(int(old) + int(w.v[vdata[1]][lane])) & 0xFFFFFFFF], dtype="<u4").view(np.uint8)
            w.mem.store(addrs, one, new)
            if ret:
                w.v[i.ops[0][1]][lane] = old

    def x_global_atomic_xor(self, w, i):
        # global_atomic_xor [vdst,] vaddr, vdata, saddr — as x_global_atomic_add
        ret = len(i.ops) == 4
        vaddr, vdata, saddr = (i.ops[1], i.ops[2], i.ops[3]) if ret else (i.ops[0], i.ops[1], i.ops[2])
        addrs = self._gaddr(w, i, vaddr, saddr)
        for lane in np.flatnonzero(w.em()):
            one = np.zeros(64, dtype=bool)
            one[lane] = True
            old = w.mem.load(addrs, one, 4).view("<u4")[lane, 0]
            new = np.zeros((64, 4), dtype=np.uint8)
            new[lane] = np.array([(int(old) ^ int(w.v[vdata[1]][lane])) & 0xFFFFFFFF], dtype="<u4").view(np.uint8)
            w.mem.store(addrs, one, new)
            if ret:
                w.v[i.ops[0][1]][lane] = old

    def x_v_bitop3_b32(self, w, i):
        # result bit = table[(a << 2) | (b << 1) | c]   (a, b, c = the bits of src0, src1, src2)
        tt = int(i.mods["bitop3"], 0)
        a, b, c = w.rv32(i.ops[1]), w.rv32(i.ops[2]), w.rv32(i.ops[3])
        r = np.zeros(64, dtype=np.uint32)
        for k in range(8):
            if (tt >> k) & 1:
                r |= (a if k & 4 else ~a) & (b if k & 2 else ~b) & (c if k & 1 else ~c)
        w.wv32(i.ops[0], r)

    def x_v_accvgpr_write_b32(self, w, i):
        w.v[256 + i.ops[0][1]] = np.where(w.em(), w.rv32(i.ops[1]), w.v[256 + i.ops[0][1]])

    def x_v_accvgpr_read_b32(self, w, i):
        w.wv32(i.ops[0], w.v[256 + i.ops[1][1]])


_WAIT_RE = re.compile(r"(vmcnt|lgkmcnt|expcnt)\((\d+)\)")


def _regs_of(o):
    return range(o[1], o[1] + o[2]) if o[0] == "v" else ()


def scoreboard(w, i):
    """Models the wait counters: a VGPR written by a load is unusable until an s_waitcnt covers the load (LDS
    operations and vector loads return in order; with stores in flight only vmcnt(0) is safe on gfx9)."""
    op = i.op
    if op == "s_waitcnt":
        for name, n in _WAIT_RE.findall(i.text):
            n = int(n)
            q = w.q_vm if name == "vmcnt" else w.q_lgkm if name == "lgkmcnt" else None
            if q is None:
                continue
            if name == "vmcnt" and n > 0 and len(q) > n and any(not d for d in q) and any(d for d in q):
                w.n_mixed_vmcnt = getattr(w, "n_mixed_vmcnt", 0) + 1
            tq = w.t_vm if name == "vmcnt" else w.t_lgkm
            while len(q) > n:
                for r in q.pop(0):
                    w.pending.pop(r, None)
                if tq:  # cycle model: the wave sleeps until the operation it waits for has returned
                    w.clock = max(w.clock, tq.pop(0))
        return
    if w.pending and i.kind != "salu":
        is_load = (i.kind == "vmem" and "_load_" in op) or (i.kind == "lds" and op.startswith("ds_read"))
        for k, o in enumerate(i.ops):
            if k == 0 and is_load:
                continue  # a later load may target a pending destination: loads return in order
            for r in _regs_of(o):
                if r in w.pending:
                    raise EmuError("v%d is used before the wait for `%s`" % (r, w.pending[r]))
    if i.kind == "lds":
        dest = list(_regs_of(i.ops[0])) if op.startswith("ds_read") or "permute" in op else []
        if "permute" not in op:
            w.q_lgkm.append(dest)
            w.t_lgkm.append(max(w.clock + Cost.LDS, w.t_lgkm[-1] if w.t_lgkm else 0))  # in-order return
            for r in dest:
                w.pending[r] = i.text
    elif i.kind == "vmem":
        dest = list(_regs_of(i.ops[0])) if "_load_" in op else []
        w.q_vm.append(dest)
        w.t_vm.append(max(w.clock + (Cost.VMEM if dest else Cost.VMEM_STORE), w.t_vm[-1] if w.t_vm else 0))
        for r in dest:
            w.pending[r] = i.text
    elif i.kind == "smem" and op.startswith("s_load"):
        w.q_lgkm.append([])
        w.t_lgkm.append(max(w.clock + Cost.SMEM, w.t_lgkm[-1] if w.t_lgkm else 0))


class Cost:
    """crude per-wave cycle model (tools/probe/issue_probe.hip): used for A/B estimates only.
    It is ONE wavefront alone: what the wavefronts of a CU share is not in it — the scalar and vector issue ports (one
    instruction of each kind per cycle per CU: the batch decoder is bound by them, and there the instruction COUNT
    predicted every hardware result of round 3 while this model's cycles did not), the LDS pipe (an access that is not
    aligned to its width occupies it for 65 cycles, tools/probe/lds_align_probe.hip) and the L1 / L2 queues."""
    ISSUE = 5
    TAKEN = 25
    NOT_TAKEN = 11
    LDS = 128
    VMEM = 700
    VMEM_STORE = 300
    SMEM = 200
    CROSS = 25  # VALU result in an SGPR -> first scalar instruction that may read it (issue_probe: 30 per producer + consumer pair)


def run_wave(prog, w, entry=None, max_inst=50_000_000, profile=None, hooks=None):
    """Runs until s_endpgm.  profile: optional dict label -> [instructions, taken branches] keyed by the most
    recent label passed (a cheap path profiler)."""
    insts, labels = prog.insts, prog.labels
    pc = labels[entry] if entry else 0
    n = 0
    cur = "<entry>"
    t_label = w.clock
    idx_label = {}
    for name, k in labels.items():
        idx_label.setdefault(k, name)
    hook_at = {labels[k]: f for k, f in (hooks or {}).items() if k in labels}
    while True:
        if profile is not None and pc in idx_label:
            if cur in profile:  # model cycles spent since the previous label
                pr = profile[cur]
                if len(pr) > 2:
                    pr[2] += w.clock - t_label
                else:
                    pr.append(w.clock - t_label)
            t_label = w.clock
            cur = idx_label[pc]
        if hook_at and pc in hook_at:
            hook_at[pc](w)
        i = insts[pc]
        op = i.op
        n += 1
        if n > max_inst:
            raise EmuError("instruction limit reached at line %d: %s" % (i.line, i.text))
        if profile is not None:
            p = profile.setdefault(cur, [0, 0])
            p[0] += 1
        k = i.kind
        if k == "salu":
            w.n_salu += 1
        elif k == "valu":
            w.n_valu += 1
        elif k == "lds":
            w.n_lds += 1
        elif k == "vmem":
            w.n_vmem += 1
        else:
            w.n_smem += 1
        # VALU -> SGPR -> SALU crossing (round 5, tools/probe/issue_probe.hip on the MI355X: a v_readlane / v_cmp whose SGPR
        # result the NEXT scalar instruction reads costs the pair 30 cycles instead of 9.4): the scalar consumer waits until
        # Cost.CROSS cycles after the producer issued; independent instructions in between hide it.
        if k == "valu":
            d = i.ops[0] if i.ops else None
            if d is not None and d[0] == "s":
                for r in range(d[1], d[1] + d[2]):
                    w.sgpr_ready[r] = w.clock + Cost.CROSS
            elif op.startswith("v_cmp") and op.endswith("_e32"):
                w.sgpr_ready[VCC] = w.sgpr_ready[VCC + 1] = w.clock + Cost.CROSS
        elif k == "salu" and w.sgpr_ready:
            t = 0
            for o in (i.ops[1:] if len(i.ops) > 1 else i.ops):
                if o[0] == "s":
                    for r in range(o[1], o[1] + o[2]):
                        t = max(t, w.sgpr_ready.get(r, 0))
            if op.startswith("s_cbranch_vcc"):
                t = max(t, w.sgpr_ready.get(VCC, 0), w.sgpr_ready.get(VCC + 1, 0))
            if t > w.clock:
                w.n_cross_wait = getattr(w, "n_cross_wait", 0) + (t - w.clock)
                w.clock = t
        w.clock += Cost.ISSUE
        if i.target is not None:
            if op == "s_branch":
                take = True
            elif op == "s_cbranch_scc0":
                take = w.scc == 0
            elif op == "s_cbranch_scc1":
                take = w.scc == 1
            elif op == "s_cbranch_vccz":
                take = (w.s[VCC] | w.s[VCC + 1]) == 0
            elif op == "s_cbranch_vccnz":
                take = (w.s[VCC] | w.s[VCC + 1]) != 0
            elif op == "s_cbranch_execz":
                take = w.exec == 0
            elif op == "s_cbranch_execnz":
                take = w.exec != 0
            else:
                raise EmuError("branch " + op)
            if take:
                w.n_taken += 1
                w.clock += Cost.TAKEN - Cost.ISSUE
                if profile is not None:
                    p[1] += 1
                pc = labels[i.target]
            else:
                w.clock += Cost.NOT_TAKEN - Cost.ISSUE
                pc += 1
            continue
        if op == "s_endpgm":
            break
        if op == "s_swappc_b64" or op == "s_setpc_b64":  # call / return (see Program.callees)
            dst = w.rs64(i.ops[-1])
            if op == "s_swappc_b64":
                w.ws64(i.ops[0], CODE_BASE + 4 * (pc + 1))
            if dst < CODE_BASE or (dst - CODE_BASE) % 4 or (dst - CODE_BASE) // 4 >= len(insts):
                raise EmuError("jump to %#x, which is not code: %s" % (dst, i.text))
            pc = (dst - CODE_BASE) // 4
            continue
        if i.fn is None:
            raise EmuError("opcode not modelled (line %d): %s" % (i.line, i.text))
        try:
            scoreboard(w, i)
            gi = getattr(w, "gpr_idx", None)
            if gi is not None and k == "valu":  # VGPR indexing mode: shifted operands for this instruction only
                idx, modes = gi
                j = copy.copy(i)
                ops = list(i.ops)
                for name, pos in (("DST", 0), ("SRC0", 1), ("SRC1", 2), ("SRC2", 3)):
                    if name in modes and pos < len(ops) and ops[pos][0] == "v":
                        ops[pos] = ("v", ops[pos][1] + idx) + tuple(ops[pos][2:])
                j.ops = ops
                i.fn(w, j)
            else:
                i.fn(w, i)
        except EmuError as e:
            raise type(e)("%s   [line %d: %s]" % (e, i.line, i.text)) from None
        pc += 1
    w.n_inst += n
    return n


def parse_objects(text):
    """data objects of the assembly (`name:` followed by .long / .short / .byte / .quad / .zero) -> {name: bytes}"""
    out, cur, buf = {}, None, None
    fmt = {".long": 4, ".int": 4, ".short": 2, ".byte": 1, ".quad": 8}
    def unescape(body):  # the assembler's string syntax: \ooo octal, \b \t \n \f \r \" \\
        b, k = bytearray(), 0
        simple = {"b": 8, "t": 9, "n": 10, "f": 12, "r": 13, '"': 34, "\\": 92}
        while k < len(body):
            ch = body[k]
            if ch != "\\":
                b.append(ord(ch))
                k += 1
            elif body[k + 1] in "01234567":
                j = k + 1
                while j < len(body) and j < k + 4 and body[j] in "01234567":
                    j += 1
                b.append(int(body[k + 1:j], 8) & 0xFF)
                k = j
            else:
                b.append(simple[body[k + 1]])
                k += 2
        return bytes(b)

    for raw in text.splitlines():
        st = raw.strip()
        if cur is not None and (st.startswith(".ascii") or st.startswith(".asciz")):
            q0, q1 = st.index('"'), st.rindex('"')
            buf += unescape(st[q0 + 1:q1]) + (b"\0" if st.startswith(".asciz") else b"")
            continue
        line = raw.split(";")[0].strip()
        if not line:
            continue
        if line.endswith(":") and not line.startswith(".L"):
            if cur is not None and buf:
                out[cur] = bytes(buf)
            cur, buf = line[:-1], bytearray()
            continue
        if cur is None:
            continue
        parts = line.split(None, 1)
        if parts[0] in fmt and len(parts) > 1:
            try:
                for v in parts[1].split(","):
                    buf += (int(v.strip(), 0) & ((1 << (8 * fmt[parts[0]])) - 1)).to_bytes(fmt[parts[0]], "little")
            except ValueError:
                cur, buf = None, None
        elif parts[0] == ".zero" and len(parts) > 1:
            buf += bytes(int(parts[1].split(",")[0], 0))
        elif parts[0] in (".size", ".p2align", ".type", ".globl", ".section"):
            continue
        else:
            if buf:
                out[cur] = bytes(buf)
            cur, buf = None, None
    if cur is not None and buf:
        out[cur] = bytes(buf)
    return out


def new_wave(mem, lds_bytes, lds_order=None):
    w = Wave(mem, max(lds_bytes, 4))
    w.symbols = {}
    w.lds_limit = lds_bytes
    w.lds_order = lds_order
    return w


class _WaveGroup:
    """The wavefronts of ONE workgroup that talk to each other (shared LDS, s_barrier, polling with s_sleep): each runs in its
    own thread, exactly one at a time (a baton); the baton moves at s_barrier, at s_sleep and when a wavefront ends."""

    def __init__(self, n):
        import threading
        self.n = n
        self.cv = threading.Condition()
        self.turn = 0
        self.alive = [True] * n
        self.waiting = [False] * n  # at the barrier
        self.error = None
        self.sleeps = 0

    def _pass(self, me):  # (cv held) the next wavefront that can run; a barrier everybody alive has reached opens
        if all(self.waiting[k] for k in range(self.n) if self.alive[k]):
            self.waiting = [False] * self.n
        for d in range(1, self.n + 1):
            k = (me + d) % self.n
            if self.alive[k] and not self.waiting[k]:
                self.turn = k
                self.cv.notify_all()
                return
        if any(self.alive):
            self.error = EmuError("workgroup deadlock: every live wavefront waits at s_barrier")
            self.cv.notify_all()

    def _wait_turn(self, me):
        while not (self.turn == me and not self.waiting[me]):
            if self.error is not None:
                raise self.error
            self.cv.wait(0.5)
        if self.error is not None:
            raise self.error

    def start(self, me):
        with self.cv:
            self._wait_turn(me)

    def barrier(self, me):
        with self.cv:
            self.waiting[me] = True
            self._pass(me)
            self._wait_turn(me)

    def sleep(self, me):
        with self.cv:
            self.sleeps += 1
            if self.sleeps > 2_000_000:
                self.error = EmuError("workgroup livelock: wavefronts keep polling")
                self.cv.notify_all()
                raise self.error
            self._pass(me)
            self._wait_turn(me)

    def end(self, me, exc=None):
        with self.cv:
            self.alive[me] = False
            if exc is not None and self.error is None:
                self.error = exc
                self.cv.notify_all()
                return
            self._pass(me)


def _launch_group(prog, entry, waves, profile, hooks):
    import threading
    g = _WaveGroup(len(waves))

    def body(k, w):
        try:
            g.start(k)
            run_wave(prog, w, entry, profile=profile, hooks=hooks)
        except BaseException as e:  # noqa: BLE001 - handed to the launching thread
            g.end(k, e)
            return
        g.end(k)

    for k, w in enumerate(waves):
        w.group, w.wave_in_group = g, k
        w.lds = waves[0].lds
    threads = [threading.Thread(target=body, args=(k, w), daemon=True) for k, w in enumerate(waves)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if g.error is not None:
        raise g.error


def launch(prog, entry, mem, kernarg, grid_x, lds_bytes, user_sgprs=2, block_x=64, profile=None, lds_order=None,
           on_wave=None, hooks=None, objects=None, cooperative=False):
    """One 64-thread workgroup per block id (the kernels here use single-wave workgroups).  ABI as hipcc emits it
    for these kernels: s[0:1] = kernarg segment, s2 = workgroup id x, v0 = thread id x."""
    if block_x % 64 != 0:
        raise EmuError("workgroups are whole wavefronts")
    # block_x > 64: the workgroup's wavefronts run one after the other, each with its own LDS image — only right for kernels
    # whose wavefronts do not talk to each other (no s_barrier, no shared LDS): run_wave refuses s_barrier then
    kbase = mem.map(np.frombuffer(bytearray(kernarg), dtype=np.uint8), "kernarg", writable=False)
    symbols = {name: mem.map(np.frombuffer(bytearray(data), dtype=np.uint8), name, writable=False)
               for name, data in (objects or {}).items()}
    symbols.update(getattr(prog, "code_symbols", {}))
    stats = []
    for bx in (grid_x if not isinstance(grid_x, int) else range(grid_x)):
        if cooperative and block_x > 64:  # wavefronts that share LDS and wait for each other: run together (see _WaveGroup)
            waves = []
            for wv in range(block_x // 64):
                w = new_wave(mem, lds_bytes, lds_order)
                w.symbols = symbols
                w.s[0], w.s[1] = kbase & M32, kbase >> 32
                w.s[user_sgprs] = bx
                w.v[0] = np.arange(64, dtype=np.uint32) + np.uint32(64 * wv)
                waves.append(w)
            _launch_group(prog, entry, waves, profile, hooks)
            for w in waves:
                stats.append(w)
                if on_wave:
                    on_wave(bx, w)
            continue
        for wv in range(block_x // 64):
            w = new_wave(mem, lds_bytes, lds_order)
            w.symbols = symbols
            w.multi_wave_group = block_x > 64
            w.s[0], w.s[1] = kbase & M32, kbase >> 32
            w.s[user_sgprs] = bx
            w.v[0] = np.arange(64, dtype=np.uint32) + np.uint32(64 * wv)  # work-item id x (y = z = 0 in the packed register)
            run_wave(prog, w, entry, profile=profile, hooks=hooks)
            stats.append(w)
            if on_wave:
                on_wave(bx, w)
    return stats
