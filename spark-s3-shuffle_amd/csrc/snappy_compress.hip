// snappy_compress.hip — raw Snappy compression of 32 KiB shuffle chunks on CDNA4, byte-exact
// with the fragment compressor of Google snappy 1.1.8 (what oracle/s3s_oracle_snappy.c restates
// and pins against libsnappy 1.1.8; the JVM's snappy-java 1.1.10.x bundles snappy 1.1.10 whose
// heuristics differ — "parity unpinned" vs the JVM).
//
// Replaces the [EXT] SnappyOutputStream.compressInput() stage (snappy-java -> JNI ->
// snappy::RawCompress) that produces the bytes arriving at S3ShuffleMapOutputWriter.scala:182-188
// when spark.io.compression.codec=snappy.
//
// Same wave64 scheme as lz4_compress.hip: one wavefront per chunk, the sequential probe loop is
// evaluated 64 probes at a time with one speculative table insert + read-back, the clean prefix
// decides which lanes saw the true candidate, the first matching lane wins, later lanes roll
// back.  What differs from LZ4:
//   * hash table: up to 16384 x u16 (32 KiB of LDS, 5 wavefronts per CU), size chosen from the
//     fragment length exactly like snappy does, hash = (bytes * 0x1e35a7bd) >> shift;
//   * skip schedule: skip starts at 32 per search, step = skip >> 5, skip += step;
//   * no backward extension; matches extend to the very end of the input;
//   * after a copy: insert ip-1, probe ip (probe index 0), then the search restarts at ip+1;
//   * element encoding: literal tags with 1-2 length bytes, 2- and 3-byte copies, long matches
//     are split into 64/60-byte copies.
//
// In front of the batch sits the EXACT WINDOW path (tests/model/snappy_window_model.cpp is its
// lock-step CPU model).  Shuffle rows are match-dense (a 32 KiB block of wide rows holds ~3500
// copies of ~8 bytes), and the batch finds one copy per memory round trip.  The window path cuts
// the fragment into aligned 64-byte windows, lane i <-> position 64k+i, and prepares a whole
// window with one table read and ONE memory round trip:
//     cp   = T[h]                 table candidate of every lane (nothing of the window is in T yet)
//     grp  = lane shares its hash with another live lane (two speculative store passes, rolled back)
//     fl   = number of equal bytes at p / cp (68 bytes of each loaded speculatively), em = fl >= 4
// The probes a run makes are a fixed pattern of the distance d to the run's base (d <= 33 every
// byte, 35..65 every 2nd, 68..98 every 3rd), so all runs of a window are resolved with scalar mask
// arithmetic: first event lane of the run; a grp lane's true candidate is the highest kept lane
// below it with the same hash (its bytes are in the window's registers), else cp.  K collects
// what the sequential code inserts (probes, and ip-1 after every copy); the highest kept lane of
// every hash commits with one store at the end of the window.
#include "s3s_internal.h"
#include "snappy_window_engine.inc"

namespace s3s {
namespace {

constexpr int kSnMaxTable = 1 << 14;
constexpr int kSnMinTable = 1 << 8;
constexpr int kSnInputMargin = 15;
constexpr int kSnSchedLen = 272;  // probes needed to skip across 64 KiB

// cumulative skip schedule: S[t] = sum of the first t steps of one search (step = skip >> 5)
struct SnSched {
  int32_t s[kSnSchedLen];
};
constexpr SnSched make_sched() {
  SnSched r{};
  uint32_t skip = 32;
  int32_t pos = 0;
  for (int t = 0; t < kSnSchedLen; t++) {
    r.s[t] = pos;
    const uint32_t step = skip >> 5;
    skip += step;
    pos += (int32_t)step;
  }
  return r;
}
__device__ const SnSched g_sn_sched = make_sched();

// position of probe u of a run relative to the run's base: u = 0 is the probe right after a
// copy (at the base itself), u >= 1 is search probe u-1 at base + 1 + S[u-1]
__device__ __forceinline__ int sn_Q(int u) {
  if (u <= 33) return u;
  const int t = u - 1;
  return 1 + g_sn_sched.s[t < kSnSchedLen ? t : kSnSchedLen - 1] + (t < kSnSchedLen ? 0 : (1 << 20));
}

__device__ __forceinline__ uint32_t sn_rd32(const uint8_t* base, int pos) {
  uint32_t v;
  __builtin_memcpy(&v, base + pos, 4);
  return v;
}

// n bytes global -> global, dword-vectorised on the destination alignment
__device__ __forceinline__ void sn_copy(uint8_t* dst, const uint8_t* src, int n, int lane) {
  int head = (int)((4u - (uint32_t)(uintptr_t)dst) & 3u);
  head = head < n ? head : n;
  if (lane < head) dst[lane] = src[lane];
  const int body = (n - head) >> 2;
  uint32_t* d32 = reinterpret_cast<uint32_t*>(dst + head);
  for (int j = lane; j < body; j += kWave) d32[j] = sn_rd32(src, head + 4 * j);
  const int done = head + 4 * body;
  if (lane < n - done) dst[done + lane] = src[done + lane];
}

// snappy EmitLiteral: tag | 0-2 length bytes | data.  len >= 1.
__device__ __forceinline__ int sn_emit_literal(uint8_t* out, int op, const uint8_t* in, int start,
                                               int len, int lane) {
  const int n = len - 1;
  int hdr;
  if (n < 60) {
    hdr = 1;
    if (lane == 0) out[op] = (uint8_t)(n << 2);
  } else {
    const int count = n < 256 ? 1 : 2;  // (Log2Floor(n) >> 3) + 1 for n < 65536
    hdr = 1 + count;
    if (lane == 0) out[op] = (uint8_t)((59 + count) << 2);
    if (lane >= 1 && lane <= count) out[op + lane] = (uint8_t)((uint32_t)n >> (8 * (lane - 1)));
  }
  sn_copy(out + op + hdr, in + start, len, lane);
  return op + hdr + len;
}

// snappy EmitCopy<len_less_than_12>: splits long matches into 64/60-byte 3-byte copies
__device__ __forceinline__ int sn_emit_copy(uint8_t* out, int op, int offset, int len, bool lt12,
                                            int lane) {
  if (lt12) {
    if (offset < 2048) {
      if (lane == 0) out[op] = (uint8_t)(1 + ((len - 4) << 2) + ((offset >> 3) & 0xe0));
      if (lane == 1) out[op + 1] = (uint8_t)offset;
      return op + 2;
    }
    if (lane < 3) out[op + lane] = lane == 0 ? (uint8_t)(2 + ((len - 1) << 2)) : (uint8_t)((uint32_t)offset >> (8 * (lane - 1)));
    return op + 3;
  }
  const int k = len >= 68 ? (len - 68) / 64 + 1 : 0;  // copies of 64
  int rem = len - 64 * k;                              // 4..67
  const bool has60 = rem > 64;
  if (has60) rem -= 60;
  // elements 0..k-1: len 64; element k: len 60 (if has60); last: rem (2 bytes iff rem < 12 && offset < 2048)
  const int n3 = k + (has60 ? 1 : 0);
  for (int j = lane; j < n3; j += kWave) {
    const int l = j < k ? 64 : 60;
    uint8_t* o = out + op + 3 * j;
    o[0] = (uint8_t)(2 + ((l - 1) << 2));
    o[1] = (uint8_t)offset;
    o[2] = (uint8_t)((uint32_t)offset >> 8);
  }
  op += 3 * n3;
  if (rem < 12 && offset < 2048) {
    if (lane == 0) out[op] = (uint8_t)(1 + ((rem - 4) << 2) + ((offset >> 3) & 0xe0));
    if (lane == 1) out[op + 1] = (uint8_t)offset;
    return op + 2;
  }
  if (lane < 3) out[op + lane] = lane == 0 ? (uint8_t)(2 + ((rem - 1) << 2)) : (uint8_t)((uint32_t)offset >> (8 * (lane - 1)));
  return op + 3;
}

// FindMatchLength(candidate + 4 + extra, ip + 4 + extra, ip_end), 256 bytes per round; the first
// `extra` bytes are known to be equal.  Returns the total number of bytes beyond the first four.
__device__ __forceinline__ int sn_find_match(const uint8_t* in, int ip0, int cand, int len, int last4,
                                             int extra, int lane) {
  for (;;) {
    const int avail = len - (ip0 + 4 + extra);
    if (avail <= 0) break;
    // the match may run to the very last byte: a lane whose dword would cross the end reads the
    // last dword of the chunk instead and drops the bytes in front of its own position
    const int fpi = ip0 + 4 + extra + 4 * lane;
    const int fp = fpi < last4 ? fpi : last4;
    uint32_t x = sn_rd32(in, fp) ^ sn_rd32(in, fp - (ip0 - cand));
    const int over = fpi - fp;
    x = over >= 4 ? 0u : (x >> (8 * over));
    const uint64_t D = __ballot(x != 0u);
    int got = 4 * kWave;
    if (D) {
      const int f = __builtin_ctzll(D);
      const uint32_t xf = __builtin_amdgcn_readlane(x, f);
      got = 4 * f + (__builtin_ctz(xf) >> 3);
    }
    got = got < avail ? got : avail;
    extra += got;
    if (got < 4 * kWave) break;
  }
  return extra;
}

typedef __attribute__((address_space(3))) uint16_t lds_u16;

__device__ __forceinline__ uint4 sn_ld16(const uint8_t* base, int pos) {
  uint4 x;
  __builtin_memcpy(&x, base + pos, 16);  // unaligned global_load_dwordx4
  return x;
}

// index of the first non-zero byte of x (16 if none)
__device__ __forceinline__ int sn_first_diff16(uint4 x) {
  int r = 16;
  r = x.w ? 12 + (__builtin_ctz(x.w) >> 3) : r;
  r = x.z ? 8 + (__builtin_ctz(x.z) >> 3) : r;
  r = x.y ? 4 + (__builtin_ctz(x.y) >> 3) : r;
  r = x.x ? (__builtin_ctz(x.x) >> 3) : r;
  return r;
}

// The parse of one fragment (chunk <= 32 KiB).  Returns the number of bytes written.
// kWin: exact windows in front of the general batch (header comment).
template <bool kWin>
__device__ int snappy_compress_wave(const uint8_t* in, lds_u16* table, int len, uint8_t* out, int lane) {
  volatile lds_u16* T = table;
  int op = 0;
  // varint32 preamble: uncompressed length
  {
    const int nb = len < 128 ? 1 : (len < 16384 ? 2 : 3);
    if (lane < nb) out[lane] = (uint8_t)(((uint32_t)len >> (7 * lane)) & 0x7f) | (lane + 1 < nb ? 0x80 : 0);
    op = nb;
  }
  if (len == 0) return op;
  int tsize = kSnMinTable;
  while (tsize < kSnMaxTable && tsize < len) tsize <<= 1;
  const int shift = 32 - (31 - __builtin_clz((uint32_t)tsize));
  const int last4 = len - 4;
  int next_emit = 0;

  if (len >= kSnInputMargin) {
    const int ip_limit = len - kSnInputMargin;
    int rbase = 0, u0 = 1;  // run base and first probe index of the next batch
    // windows: far enough from the end that no probe of a window meets ip_limit and the
    // speculative 68-byte loads stay inside the fragment
    const int fast_limit = len - 192;
    int kn = -1, kp = -1;     // window bases whose dwords are held in vn / vp
    uint32_t vn = 0, vp = 0;  // dwords of the next and of the previous window
    bool finished = false;
    for (;;) {
      if constexpr (kWin) {
        // position of probe u0 (closed form of sn_Q for u0 <= 60)
        const int q0 = u0 <= 33 ? u0 : (u0 <= 49 ? 2 * u0 - 33 : 3 * u0 - 82);
        int wbase = (rbase + q0) & ~63;
        if (u0 <= 60 && wbase + 63 - rbase <= 98 && wbase <= fast_limit) {
          // window state: set by the preparation below, or by the hand-written block when it hands over in mid-window
          uint32_t v, h, cp, info;
          uint64_t Dp, ED, K, runmask;
          int rt, pend_q;
          int cap_extra = 64;  // "at least" value of a record's length field
          bool resumed = false;
#ifndef S3S_NO_WINDOW_ENGINE
          if ((uint32_t)reinterpret_cast<uintptr_t>(table) == 0u) {  // (the block addresses the table at LDS offset 0)
            // Whole windows in one hand-written gfx950 block (snappy_window_engine.inc): it returns at a window boundary
            // (code 0), when a copy reaches ip_limit (3), or at the first event it does not handle (2: the loop resumes).
            int code, pk_e = -1000;
            const int rbase_in = rbase, u0_in = u0;
            uint32_t vcur, vnext;
            rbase = __builtin_amdgcn_readfirstlane(rbase);  // (uniform already; hipcc cannot always prove it)
            u0 = __builtin_amdgcn_readfirstlane(u0);
            next_emit = __builtin_amdgcn_readfirstlane(next_emit);
            op = __builtin_amdgcn_readfirstlane(op);
            kp = __builtin_amdgcn_readfirstlane(kp);
            asm volatile(S3S_SNAPPY_ENGINE_ASM
                         : [code] "=&s"(code), [h] "=&v"(h), [cp] "=&v"(cp), [info] "=&v"(info), [vcur] "=&v"(vcur),
                           [vnext] "=&v"(vnext), [rbase] "+s"(rbase), [u0] "+s"(u0), [ne] "+s"(next_emit), [op] "+s"(op),
                           [pk] "+s"(pk_e), [kp] "+s"(kp), [K] "=&s"(K), [ED] "=&s"(ED), [Dp] "=&s"(Dp), [rm] "=&s"(runmask),
                           [pendq] "=&s"(pend_q), [rt] "=&s"(rt), [vp] "+v"(vp)
                         : [len] "s"(len), [inp] "s"(in), [outp] "s"(out), [shift] "s"(__builtin_amdgcn_readfirstlane(shift))
                         : "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77",
                           "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91",
                           "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99", "m0", "v97", "v98", "v99", "v100", "v101", "v102",
                           "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115",
                           "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124",
                           "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136",
                           "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148",
                           "v149", "v150", "v151", "v152", "v153", "vcc", "scc", "memory");
            if (pk_e >= 0) {  // the block ran: its stream registers replace the cached dwords
              vn = vnext;
              kn = pk_e + 64;
            }
            if (code == 3) {
              finished = true;
              break;
            }
            if (code == 0 && (rbase != rbase_in || u0 != u0_in)) continue;  // re-dispatch at the next window
            if (code == 2) {
              resumed = true;
              cap_extra = 12;
              wbase = pk_e;
              v = vcur;
              if (kp != wbase - 64) vp = sn_rd32(in, wbase >= 64 ? wbase - 64 + lane : wbase + lane);
            }
          }
#endif
          const int p = wbase + lane;
          if (!resumed) {
          v = (kn == wbase) ? vn : sn_rd32(in, p);
          if (kp != wbase - 64) vp = sn_rd32(in, wbase >= 64 ? p - 64 : p);
          vn = sn_rd32(in, p + 64);
          kn = wbase + 64;
          h = (v * 0x1e35a7bdu) >> shift;
          const int rs0 = rbase + q0 - wbase;
          const bool live = lane >= rs0;
          cp = T[h];
          // 68 bytes at the position and at its table candidate, one round trip
          const uint4 b0 = sn_ld16(in, (int)cp), b1 = sn_ld16(in, (int)cp + 16);
          const uint4 b2 = sn_ld16(in, (int)cp + 32), b3 = sn_ld16(in, (int)cp + 48);
          const uint32_t b4 = sn_rd32(in, (int)cp + 64);
          const uint4 a0 = sn_ld16(in, p), a1 = sn_ld16(in, p + 16);
          const uint4 a2 = sn_ld16(in, p + 32), a3 = sn_ld16(in, p + 48);
          // duplicate-hash groups among the live lanes: two speculative store passes, rolled back
          // (lanes of a group of >= 2 either lose pass 1 or see pass 2's winner)
          bool grp = false;
          if (live) {
            T[h] = (uint16_t)p;
            const uint32_t r1 = T[h];
            const bool lost1 = r1 != (uint32_t)p;
            if (lost1) T[h] = (uint16_t)p;
            const uint32_t r2 = T[h];
            grp = lost1 || (r2 != (uint32_t)p);
            if (r2 == (uint32_t)p) T[h] = (uint16_t)cp;  // the slot's current owner restores it
          }
          // runs entering the window probe a fixed pattern of the distance to their base
          const int dd = p - rbase;
          const bool isprobe = live && (dd <= 33 || (dd <= 65 ? (dd & 1) != 0 : (dd >= 68 && (dd + 1) % 3 == 0)));
          runmask = __ballot(isprobe);
          // equal bytes at p / cp: 0..68
          const int f0 = sn_first_diff16(make_uint4(a0.x ^ b0.x, a0.y ^ b0.y, a0.z ^ b0.z, a0.w ^ b0.w));
          const int f1 = sn_first_diff16(make_uint4(a1.x ^ b1.x, a1.y ^ b1.y, a1.z ^ b1.z, a1.w ^ b1.w));
          const int f2 = sn_first_diff16(make_uint4(a2.x ^ b2.x, a2.y ^ b2.y, a2.z ^ b2.z, a2.w ^ b2.w));
          const int f3 = sn_first_diff16(make_uint4(a3.x ^ b3.x, a3.y ^ b3.y, a3.z ^ b3.z, a3.w ^ b3.w));
          const uint32_t x4 = vn ^ b4;  // the dword at p + 64 is the next window's own dword
          const int f4 = x4 ? (__builtin_ctz(x4) >> 3) : 4;
          int fl = 64 + f4;
          fl = f3 < 16 ? 48 + f3 : fl;
          fl = f2 < 16 ? 32 + f2 : fl;
          fl = f1 < 16 ? 16 + f1 : fl;
          fl = f0 < 16 ? f0 : fl;
          const bool em = live && fl >= 4;
          // per-lane record: [15:0] table candidate, [22:16] equal bytes beyond the first four (64 =
          // at least), [29] candidate matches, [31] member of a duplicate-hash group
          info = cp | ((uint32_t)(em ? fl - 4 : 0) << 16) | (em ? 0x20000000u : 0u) | (grp ? 0x80000000u : 0u);
          asm volatile("" : "+v"(vn));  // landed: keep later uses from draining the in-order vmcnt queue
          const uint64_t Ecp = __ballot(em);
          Dp = __ballot(grp);
          ED = Ecp | Dp;
          // ================= runs (scalar work) =====================================================
          K = 0;  // lanes the sequential code inserts: probes, and ip-1 after every copy
          rt = u0;
          pend_q = -1;
          }
          const uint64_t PM = 0xAAAAAAAA00000000ull | ((1ull << 34) - 1ull);  // d: 0..33, 35, 37, .., 63
          // ---- straight-line steps: up to six "plain" copies (the run's first event has its true candidate in the
          // table with an exact length, one literal tag + one copy element, literals in the registers of this or the
          // previous window, not the end of the fragment) without the generic loop's control flow: one branch
          // decides, everything else is arithmetic.  Anything else falls through to the generic loop, which continues
          // from whatever state the steps left (lz4_compress.hip does the same).
          bool left_window = false;
#pragma unroll
          for (int step = 0; step < (resumed ? 0 : 6); step++) {
            const uint64_t cm = ED & runmask;
            if (cm == 0ull) break;
            const int m = __builtin_ctzll(cm);
            const uint64_t bit = 1ull << m;
            const uint32_t inf = __builtin_amdgcn_readlane(info, m);
            const uint32_t hv = __builtin_amdgcn_readlane(h, m);
            const uint64_t dk = __ballot(h == hv) & (bit - 1ull) & (K | runmask);  // earlier kept lanes, same hash
            const int ip0 = wbase + m;
            const int cand = (int)(inf & 0xffffu), extra = (int)((inf >> 16) & 0x7fu);
            const int lit = ip0 - next_emit, offset = ip0 - cand, matched = 4 + extra;
            const int ipe = ip0 + matched;
            int over = extra - 60;                                         // capped, or more than one copy element
            over = over > lit - 60 ? over : lit - 60;                      // literal tag needs length bytes
            over = over > wbase - 64 - next_emit ? over : wbase - 64 - next_emit;  // literals older than the registers
            over = over > ipe - ip_limit + 1 ? over : ipe - ip_limit + 1;  // end of the fragment
            if ((inf & 0x20000000u) == 0u || dk != 0ull || over > 0) break;
            K |= runmask & ((bit << 1) - 1ull);
            {
              const bool two = matched < 12 && offset < 2048;
              const int hdr = lit > 0 ? 1 : 0;
              const int total = hdr + lit + (two ? 2 : 3);
              const uint32_t c0 = two ? (uint32_t)(1 + ((matched - 4) << 2) + ((offset >> 8) << 5))
                                      : (uint32_t)(2 + ((matched - 1) << 2));
              const int rel = (lane - next_emit) & 63;
              const int k = rel - lit;  // 0: literal tag (if any), then the copy bytes
              const int j = k - hdr;
              uint32_t bv = ((next_emit + rel < wbase) ? vp : v) & 0xffu;
              int idx = 1 + rel;
              if (k >= 0) {
                idx = (k == 0) ? 0 : lit + k;
                bv = (j < 0) ? (uint32_t)((lit - 1) << 2) : (j == 0 ? c0 : (j == 1 ? (uint32_t)offset : (uint32_t)offset >> 8));
              }
              if (rel < total) out[op + idx] = (uint8_t)bv;
              op += total;
            }
            next_emit = ipe;
            const int q = ipe - 1 - wbase;  // table[Hash(ip - 1)] = ip - 1
            K |= q < kWave ? (1ull << (q & 63)) : 0ull;
            pend_q = q < kWave ? pend_q : ipe - 1;
            rbase = ipe;
            u0 = 0;
            rt = 0;
            if (ipe >= wbase + kWave) {
              left_window = true;
              break;
            }
            runmask = PM << (ipe - wbase);
          }
          if (!left_window) for (;;) {
            const uint64_t cm = ED & runmask;
            if (cm == 0ull) {  // the run leaves the window without a match
              K |= runmask;
              u0 = rt + __popcll(runmask);
              break;
            }
            const int m = __builtin_ctzll(cm);
            const uint64_t bit = 1ull << m;
            const uint32_t inf = __builtin_amdgcn_readlane(info, m);
            const int ip0 = wbase + m;
            int cand = (int)(inf & 0xffffu);
            int extra = (int)((inf >> 16) & 0x7fu);
            bool capped = extra >= cap_extra;
            if (__builtin_expect((int)inf < 0, 0)) {
              // ---- another live lane has the same hash: the candidate may be inside the window ----
              bool is_match = (inf & 0x20000000u) != 0u;
              const uint32_t hv = __builtin_amdgcn_readlane(h, m);
              const uint64_t dk = __ballot(h == hv) & (bit - 1ull) & (K | runmask);
              if (dk) {
                const int d = 63 - __builtin_clzll(dk);
                is_match = __builtin_amdgcn_readlane(v, d) == __builtin_amdgcn_readlane(v, m);
                if (is_match) {
                  // bytes behind both positions are in the registers of this and the next window
                  const int oa = d + 4 + lane, ob = m + 4 + lane;
                  const uint32_t a_lo = __shfl(v, oa & 63), a_hi = __shfl(vn, oa & 63);
                  const uint32_t b_lo = __shfl(v, ob & 63), b_hi = __shfl(vn, ob & 63);
                  const uint32_t ba = (oa < 64 ? a_lo : a_hi) & 0xffu, bb = (ob < 64 ? b_lo : b_hi) & 0xffu;
                  const uint64_t E = __ballot(ba != bb || ob > 127);
                  extra = E ? __builtin_ctzll(E) : 64;
                  capped = m + 4 + extra > 127 || extra >= 64;
                  cand = wbase + d;
                }
              }
              if (!is_match) {
                ED &= ~bit;  // a plain no-match probe: the run goes on behind it
                continue;
              }
            }
            if (__builtin_expect(capped, 0)) extra = sn_find_match(in, ip0, cand, len, last4, extra, lane);
            K |= runmask & ((bit << 1) - 1ull);
            const int lit = ip0 - next_emit, offset = ip0 - cand, matched = 4 + extra;
            if (__builtin_expect(lit <= 60 && next_emit >= wbase - 64 && matched <= 64, 1)) {
              // literal tag | literals | copy in ONE store: every literal is the low byte of a lane's
              // dword of this or the previous window; otherwise idle lanes carry the tag bytes
              const bool two = matched < 12 && offset < 2048;
              const int hdr = lit > 0 ? 1 : 0;
              const int total = hdr + lit + (two ? 2 : 3);
              const uint32_t c0 = two ? (uint32_t)(1 + ((matched - 4) << 2) + ((offset >> 8) << 5))
                                      : (uint32_t)(2 + ((matched - 1) << 2));
              const uint32_t c1 = (uint32_t)offset;
              const int rel = (lane - next_emit) & 63;
              const int k = rel - lit;  // 0: literal tag (if any), then the copy bytes
              const int j = k - hdr;
              uint32_t bv = ((next_emit + rel < wbase) ? vp : v) & 0xffu;
              int idx = 1 + rel;
              if (k >= 0) {
                idx = (k == 0) ? 0 : lit + k;
                bv = (j < 0) ? (uint32_t)((lit - 1) << 2) : (j == 0 ? c0 : (j == 1 ? c1 : (uint32_t)offset >> 8));
              }
              if (rel < total) out[op + idx] = (uint8_t)bv;
              op += total;
            } else {
              if (lit > 0) op = sn_emit_literal(out, op, in, next_emit, lit, lane);
              op = sn_emit_copy(out, op, offset, matched, extra < 8, lane);
            }
            const int ipe = ip0 + matched;
            next_emit = ipe;
            if (__builtin_expect(ipe >= ip_limit, 0)) {
              finished = true;
              break;
            }
            const int q = ipe - 1 - wbase;  // table[Hash(ip - 1)] = ip - 1
            if (q < kWave) K |= 1ull << q;
            else pend_q = ipe - 1;
            rbase = ipe;
            u0 = 0;
            if (ipe >= wbase + kWave) break;
            rt = 0;
            runmask = PM << (ipe - wbase);
          }
          if (finished) break;
          // ================= commit: the highest kept lane of every hash writes ====================
          // All kept lanes store at once; which lane wins a same-address store is the hardware's choice, so
          // the lanes read back and every kept lane that finds a LOWER position in its slot stores again,
          // until none does (the owner only moves up: at most group-size rounds, one or two in practice).
          // This replaces a scalar loop over the kept group lanes (~10 SALU each, dozens per window).
          const bool kept = ((K >> lane) & 1ull) != 0ull;
          if (kept) T[h] = (uint16_t)p;
          if (K & Dp) {
            for (;;) {
              const bool redo = kept && (uint32_t)T[h] < (uint32_t)p;
              if (!__ballot(redo)) break;
              if (redo) T[h] = (uint16_t)p;
            }
          }
          if (pend_q >= 0) {
            const uint32_t vq = pend_q < wbase + 2 * kWave ? __builtin_amdgcn_readlane(vn, pend_q - wbase - kWave)
                                                          : sn_rd32(in, pend_q);
            T[(vq * 0x1e35a7bdu) >> shift] = (uint16_t)pend_q;
          }
          vp = v;
          kp = wbase;
          continue;
        }
      }
      // ---- one batch: lane i evaluates probe u0+i of the current run ------------------------------
      int nl = kWave;               // lanes offered
      if (u0 <= 1) nl = 34 - u0;    // the consecutive part of a fresh run needs no schedule lookup
      const int u = u0 + lane;
      const int pos = rbase + sn_Q(u);
      const int nextpos = rbase + sn_Q(u + 1);
      const bool ok = lane < nl && (u == 0 || nextpos <= ip_limit);
      int nvalid = __popcll(__ballot(ok));  // valid lanes form a prefix
      const bool valid = lane < nvalid;
      const uint32_t v = sn_rd32(in, pos < last4 ? pos : last4);
      const uint32_t h = (v * 0x1e35a7bdu) >> shift;
      uint32_t c = 0, r = (uint32_t)pos;
      if (valid) {
        c = T[h];
        T[h] = (uint16_t)pos;
        r = T[h];
      }
      const uint32_t w = sn_rd32(in, (int)c);
      const uint32_t vprev = __builtin_amdgcn_update_dpp(~v, v, 0x138 /*wave_shr:1*/, 0xf, 0xf, false);
      const uint64_t L = __ballot(r != (uint32_t)pos);
      const uint64_t M = __ballot(valid && w == v);
      const uint64_t A = __ballot(valid && lane > 0 && v == vprev);
      int B = kWave, c0 = -1;
      bool clean0 = false;
      if (L) {
        c0 = __builtin_ctzll(L);
        const uint32_t rc0 = __builtin_amdgcn_readlane(r, c0);
        const uint32_t pc0 = __builtin_amdgcn_readlane((uint32_t)pos, c0);
        clean0 = rc0 > pc0;
        B = c0 + (clean0 ? 1 : 0);
      }
      const int lim = B < nvalid ? B : nvalid;
      const uint64_t Mv = lim >= kWave ? M : (M & ((1ull << lim) - 1ull));
      int m = -1, keep = lim;
      bool adj = false;
      if (Mv) {
        m = __builtin_ctzll(Mv);
        keep = m + 1;
      } else if (lim < nvalid && ((A >> lim) & 1ull)) {
        m = lim;  // first non-clean lane repeats its clean predecessor: its candidate is that probe
        adj = true;
        keep = lim + 1;
      }
      if (valid && lane >= keep && r == (uint32_t)pos) T[h] = (uint16_t)c;
      const bool redo_c0 = clean0 && c0 < keep && !(adj && c0 == m - 1);
      if ((redo_c0 && lane == c0) || (adj && lane == m)) T[h] = (uint16_t)pos;

      if (m < 0) {
        if (lim == nvalid && nvalid < nl) break;  // next_ip > ip_limit: emit the remainder
        u0 += lim;                                // the run goes on
        continue;
      }
      // ---- copy at lane m ------------------------------------------------------------------------
      const int ip0 = (int)__builtin_amdgcn_readlane((uint32_t)pos, m);
      const int cand = adj ? (int)__builtin_amdgcn_readlane((uint32_t)pos, m - 1)
                           : (int)__builtin_amdgcn_readlane(c, m);
      const int extra = sn_find_match(in, ip0, cand, len, last4, 0, lane);
      if (ip0 > next_emit) op = sn_emit_literal(out, op, in, next_emit, ip0 - next_emit, lane);
      const int matched = 4 + extra;
      op = sn_emit_copy(out, op, ip0 - cand, matched, extra < 8, lane);
      const int ipe = ip0 + matched;
      next_emit = ipe;
      if (ipe >= ip_limit) break;
      T[(sn_rd32(in, ipe - 1) * 0x1e35a7bdu) >> shift] = (uint16_t)(ipe - 1);
      rbase = ipe;
      u0 = 0;
    }
  }
  if (next_emit < len) op = sn_emit_literal(out, op, in, next_emit, len - next_emit, lane);
  return op;
}

template <bool kWin>
__global__ __launch_bounds__(kWave) void snappy_compress_kernel(
    const uint8_t* __restrict__ src, const Item* __restrict__ items, int32_t n_items,
    uint8_t* __restrict__ slots, int64_t slot_stride, uint32_t* __restrict__ item_size) {
  __shared__ __attribute__((aligned(16))) uint16_t table[kSnMaxTable];
  const int it = blockIdx.x;
  if (it >= n_items) return;
  const Item item = items[it];
  const int kind = item.kind & 0xff;
  const int lane = threadIdx.x;
  if (kind != kItemSnappyChunk) {
    if (lane == 0 && kind == kItemSnappyHeader) item_size[it] = kSnappyStreamHeader;
    return;
  }
  {
    uint4* tz = reinterpret_cast<uint4*>(table);
    for (int i = lane; i < (int)(sizeof(table) / 16); i += kWave) tz[i] = make_uint4(0, 0, 0, 0);
  }
  __syncthreads();
  // touch the whole block first (32 independent 1 KiB rows, one wait), as the LZ4 kernel does: the parse's dependent
  // gathers then hit in this XCD's L2 (wide rows 30.6 -> 32.0 GB/s)
  {
    const uint8_t* g = src + item.src_off;
    uint32_t acc = 0;
    for (int i = lane * 16; i + 16 <= item.len; i += kWave * 16) {
      uint4 x;
      __builtin_memcpy(&x, g + i, 16);
      acc ^= x.x ^ x.y ^ x.z ^ x.w;
    }
    if (acc == 0x12345678u && item.len < 0) table[0] = 1;  // (never taken: keeps the loads)
  }
  uint8_t* slot = slots + (size_t)item.chunk * (size_t)slot_stride;
  const int clen = snappy_compress_wave<kWin>(src + item.src_off, (lds_u16*)table, item.len, slot + kSlotHeader, lane);
  // SnappyOutputStream.dumpOutput(): i32 BE compressed length in front of the raw block
  if (lane < 4) slot[kSlotHeader - 4 + lane] = (uint8_t)((uint32_t)clen >> (8 * (3 - lane)));
  if (lane == 0) item_size[it] = 4u + (uint32_t)clen;
}

}  // namespace

bool snappy_compress_available() { return true; }

void launch_snappy_compress(const uint8_t* d_src, const Item* d_items, int32_t n_items,
                            uint8_t* d_slots, int64_t slot_stride, uint32_t* d_item_size,
                            int variant, hipStream_t st) {
  if (n_items <= 0) return;
  if (variant == 0)
    hipLaunchKernelGGL(snappy_compress_kernel<false>, dim3((unsigned)n_items), dim3(kWave), 0, st, d_src,
                       d_items, n_items, d_slots, slot_stride, d_item_size);
  else
    hipLaunchKernelGGL(snappy_compress_kernel<true>, dim3((unsigned)n_items), dim3(kWave), 0, st, d_src,
                       d_items, n_items, d_slots, slot_stride, d_item_size);
}

}  // namespace s3s
