// lz4_compress.hip — bit-exact LZ4 block compression of 32 KiB shuffle chunks on CDNA4.
//
// Replaces the [EXT] LZ4BlockOutputStream.flushBufferedData() stage (lz4-java 1.8.0 ->
// liblz4 1.9.3 LZ4_compress_default + xxHash32) that produces the bytes arriving at
// S3ShuffleMapOutputWriter.scala:182-188 in the reference.  Output must equal the JVM path
// byte for byte, so this is NOT a "GPU-friendly LZ4 variant": it reproduces the greedy
// single-pass parse of LZ4_compress_generic(byU16, acceleration 1) exactly, including its
// hash-table update order, skip schedule, backward catch-up and end-of-block rules.
//
// One wavefront per chunk.  The sequential probe loop of the CPU code is evaluated 64
// positions at a time: lane i takes the i-th position of the deterministic "no match yet"
// schedule, hashes it, reads the 8192 x u16 table (LDS), and ALL lanes insert speculatively
// with one ds_write.  A readback tells every lane whether it lost a same-slot race (=> some
// lane in the batch shares its hash); the first such lane bounds the prefix in which the
// pre-batch table entries are the true candidates ("cut").  The first lane below the cut
// whose candidate matches wins (ballot + ctz); lanes after it undo their inserts, so the
// table state is exactly the sequential one.  Match extension is cooperative: 256 B forward /
// 64 B backward per memory round trip.  Which lane wins a same-address LDS store is
// irrelevant to the result (tests/model/lz4_wave_model.cpp proves it under adversarial
// orders).
//
// Two placements of the chunk bytes, same parse (template parameter):
//   kInLds   chunk staged in LDS (32 KiB + 16 KiB table -> 3 wavefronts per CU)
//   kInL2    chunk read in place through L1/L2 (16 KiB table only -> 10 wavefronts per CU);
//            the parse is issue/latency bound, so resident wavefronts are what buy throughput
// The frame's xxHash32 is computed by a separate streaming kernel (16 chunks per wavefront).
// Output: token/literal/offset bytes go straight to the chunk's slot in HBM; compressed output
// never exceeds the chunk length (anything longer is stored RAW by the frame rule
// compressedLength >= originalLength), so a slot is 32 B + 32 KiB.
#include "s3s_internal.h"

#ifdef S3S_LZ4_TIMING
__device__ unsigned long long g_lz4_dbg[32];
#define DBG_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define DBG_ADD(slot, val) dbg[slot] += (val)
#else
#define DBG_T(var)
#define DBG_ADD(slot, val)
#endif

namespace s3s {
namespace {

constexpr int kLdsSlack = 320;  // cooperative compares over-read at most 4*63+3+7 bytes
constexpr int kMfLimit = 12, kLastLiterals = 5, kMinMatch = 4;

__device__ __forceinline__ uint32_t hash13(uint32_t v) { return (v * 2654435761u) >> 19; }

// Cumulative LZ4 skip schedule.  Probe index t (t = 0 is the "test next position" probe right
// after a match, t >= 1 the search loop with searchMatchNb starting at 64) sits S(t) bytes
// after the run's first probe:  step_0 = step_1 = 1, step_t = (62 + t) >> 6 for t >= 2.
__device__ __forceinline__ int sched_S(int t) {
  const int X = 62 + t;
  const int q = X >> 6, r = X & 63;
  return t < 2 ? t : 2 + 32 * q * (q - 1) + q * r;
}

// ---- chunk byte sources -------------------------------------------------------------------
struct SrcLds {  // chunk staged in LDS
  static constexpr bool kClamp = false;  // the LDS copy has kLdsSlack bytes of slack
  const uint8_t* base;
  __device__ __forceinline__ uint32_t rd32(int pos) const {
    uint32_t v;
    __builtin_memcpy(&v, base + pos, 4);  // gfx950: unaligned ds_read_b32
    return v;
  }
  __device__ __forceinline__ uint32_t rd8(int pos) const { return base[pos]; }
};
struct SrcGlobal {  // chunk read in place (L1/L2)
  static constexpr bool kClamp = true;  // never read past the chunk: it may end the allocation
  const uint8_t* base;
  // positions are never negative: the unsigned index lets hipcc address with saddr + 32-bit voffset
  __device__ __forceinline__ uint32_t rd32(int pos) const {
    uint32_t v;
    __builtin_memcpy(&v, base + (uint32_t)pos, 4);  // unaligned global_load_dword
    return v;
  }
  __device__ __forceinline__ uint32_t rd8(int pos) const { return base[(uint32_t)pos]; }
  __device__ __forceinline__ uint4 ld16(int pos) const {
    uint4 x;
    __builtin_memcpy(&x, base + (uint32_t)pos, 16);  // unaligned global_load_dwordx4
    return x;
  }
  __device__ __forceinline__ uint2 ld8(int pos) const {
    uint2 x;
    __builtin_memcpy(&x, base + (uint32_t)pos, 8);
    return x;
  }
};

// n bytes chunk -> global, dword-vectorised on the destination alignment.
template <typename Src>
__device__ __forceinline__ void copy_to_global(uint8_t* dst, const Src& in, int src_pos, int n,
                                               int lane) {
  int head = (int)((4u - (uint32_t)(uintptr_t)dst) & 3u);
  head = head < n ? head : n;
  if (lane < head) dst[lane] = (uint8_t)in.rd8(src_pos + lane);
  const int body = (n - head) >> 2;
  uint32_t* d32 = reinterpret_cast<uint32_t*>(dst + head);
  for (int j = lane; j < body; j += kWave) d32[j] = in.rd32(src_pos + head + 4 * j);
  const int done = head + 4 * body;
  if (lane < n - done) dst[done + lane] = (uint8_t)in.rd8(src_pos + done + lane);
}

// Writes one LZ4 sequence (token, literal-length bytes, literals and — if has_match — offset
// and match-length bytes) at out+op.  All scalar arguments are wave-uniform.  Returns the new
// op, or -1 when the sequence would not fit in cap bytes (=> the frame is stored RAW).
// lit_reg: when >= 0x100 the literal run is NOT available in registers; otherwise every lane k
// holds in `lit_byte` the literal that belongs at output byte 1+... (see caller).
template <typename Src>
__device__ __forceinline__ int emit_sequence(uint8_t* out, int cap, int op, const Src& in,
                                             int anchor, int lit, bool has_match, int offset,
                                             int mcode, bool lit_in_regs, uint32_t lit_byte,
                                             int lane) {
  uint8_t* o = out + op;
  if (has_match && lit < 15 && mcode < 15 + 255) {
    // short form (the common case): token | literals | offset | [one match-length byte]
    const int total = 3 + lit + (mcode >= 15 ? 1 : 0);
    if (op + total > cap) return -1;
    uint32_t b = lit_byte;
    if (!lit_in_regs) {
      int lp = anchor + lane - 1;  // lanes 1..lit carry the literals
      lp = lp < anchor ? anchor : lp;
      b = in.rd8(lp);
    }
    b = (lane == 0) ? ((uint32_t)(lit << 4) | (uint32_t)(mcode < 15 ? mcode : 15)) : b;
    b = (lane == lit + 1) ? (uint32_t)offset : b;
    b = (lane == lit + 2) ? ((uint32_t)offset >> 8) : b;
    b = (lane == lit + 3) ? (uint32_t)(mcode - 15) : b;
    if (lane < total) o[lane] = (uint8_t)b;
    return op + total;
  }
  const int le = lit >= 15 ? (lit - 15) / 255 + 1 : 0;
  const int me = (has_match && mcode >= 15) ? (mcode - 15) / 255 + 1 : 0;
  const int total = 1 + le + lit + (has_match ? 2 + me : 0);
  if (op + total > cap) return -1;
  const uint32_t token =
      (uint32_t)((lit < 15 ? lit : 15) << 4) | (uint32_t)(has_match ? (mcode < 15 ? mcode : 15) : 0);
  const int lit0 = 1 + le;      // first literal byte
  const int off0 = lit0 + lit;  // offset low byte
  const uint32_t lrem = (uint32_t)((lit - 15) % 255), mrem = (uint32_t)((mcode - 15) % 255);
  if (lane == 0) o[0] = (uint8_t)token;
  for (int j = lane; j < le; j += kWave) o[1 + j] = (uint8_t)(j < le - 1 ? 255u : lrem);
  copy_to_global(o + lit0, in, anchor, lit, lane);
  if (has_match) {
    if (lane < 2) o[off0 + lane] = (uint8_t)((uint32_t)offset >> (8 * lane));
    for (int j = lane; j < me; j += kWave) o[off0 + 2 + j] = (uint8_t)(j < me - 1 ? 255u : mrem);
  }
  return op + total;
}


// lanes [lo,hi) as a 64-bit mask (0 <= lo <= 63, 0 <= hi <= 64)
__device__ __forceinline__ uint64_t lane_range(int lo, int hi) {
  const uint64_t top = hi >= 64 ? ~0ull : ((1ull << hi) - 1ull);
  return hi <= lo ? 0ull : (top & ~((1ull << lo) - 1ull));
}

// index of the first differing byte of two 16-byte blocks given x = a ^ b (16 if equal)
__device__ __forceinline__ int first_diff16(uint4 x) {
  int r = 16;
  r = x.w ? 12 + (__builtin_ctz(x.w) >> 3) : r;
  r = x.z ? 8 + (__builtin_ctz(x.z) >> 3) : r;
  r = x.y ? 4 + (__builtin_ctz(x.y) >> 3) : r;
  r = x.x ? (__builtin_ctz(x.x) >> 3) : r;
  return r;
}

// Cooperative match extension (LZ4 catch-up + LZ4_count), all arguments wave-uniform:
// backward over the pending literals 64 bytes per round, forward 256 bytes per round.
// Returns the number of bytes beyond MINMATCH; nback = bytes the match start moves back.
template <typename Src>
__device__ __forceinline__ int extend_match(const Src& in, int ip0, int match0, int anchor,
                                            int matchlimit, int last4, int lane, int& nback) {
  int ip = ip0, match = match0;
  int maxback = ip - anchor < match ? ip - anchor : match;
  uint32_t ba = 0, bb = 1;
  if (lane < maxback) {
    ba = in.rd8(ip - 1 - lane);
    bb = in.rd8(match - 1 - lane);
  }
  int fp = ip0 + kMinMatch + 4 * lane;
  if (Src::kClamp) fp = fp < last4 ? fp : last4;
  uint32_t x = in.rd32(fp) ^ in.rd32(fp - (ip0 - match0));
  while (maxback > 0) {
    const uint64_t E = __ballot(lane < maxback && ba == bb);
    const int nbk = (~E == 0ull) ? kWave : __builtin_ctzll(~E);
    ip -= nbk;
    match -= nbk;
    if (nbk < kWave) break;
    maxback -= kWave;
    if (lane < maxback) {
      ba = in.rd8(ip - 1 - lane);
      bb = in.rd8(match - 1 - lane);
    }
  }
  int fwd = 0;
  for (;;) {
    const int avail = matchlimit - (ip0 + kMinMatch + fwd);
    if (avail <= 0) break;
    const uint64_t D = __ballot(x != 0u);
    int got = 4 * kWave;
    if (D) {
      const int f = __builtin_ctzll(D);
      const uint32_t xf = __builtin_amdgcn_readlane(x, f);
      got = 4 * f + (__builtin_ctz(xf) >> 3);
    }
    got = got < avail ? got : avail;
    fwd += got;
    if (got < 4 * kWave) break;
    fp = ip0 + kMinMatch + fwd + 4 * lane;
    if (Src::kClamp) fp = fp < last4 ? fp : last4;
    x = in.rd32(fp) ^ in.rd32(fp - (ip0 - match0));
  }
  nback = ip0 - ip;
  return fwd;
}

typedef __attribute__((address_space(3))) uint16_t lds_u16;

#ifndef S3S_FAST_STEPS
#define S3S_FAST_STEPS 5  // measured: 3 -> 45.3, 5 -> 45.4 GB/s (TeraSort, two task threads), 21.2 vs 20.6 on wide rows
#endif

// The parse.  Returns the compressed size, or -1 if it would exceed len.
//
// kFast adds the "exact window" path in front of the general batch (DESIGN.md §6,
// tests/model/lz4_window_model.cpp is its lock-step CPU model):
//   the chunk is cut into aligned 64-byte windows, lane i <-> position 64k+i.  While the probe
//   index of the current run is small, consecutive probes are consecutive bytes, so a whole
//   window can be prepared with ONE table read and two memory round trips:
//     cp    = T[h]                      table candidate of every lane (nothing of this window is
//                                       in the table yet)
//     Dp    ⊇ lanes that share their hash with an earlier lane of the window (found with two
//             speculative store passes that are rolled back; which lane wins a same-address
//             store only decides which superset we get)
//     Ecp   = lanes whose table candidate matches, with per-lane forward length (<= 64) and
//             backward equal count (<= 8) loaded and computed speculatively
//   and then resolved run by run with SCALAR mask arithmetic only.  K collects what the
//   sequential code inserts (probes and ip-2 positions); for a suspect probe the true candidate
//   is the highest lane of (same hash) & K below it, else cp.  One store commits K at the end.
template <typename Src, int kMode>
__device__ int lz4_compress_wave(const Src in, lds_u16* table, int len, uint8_t* out, int lane) {
  volatile lds_u16* T = table;  // every access is a real ds_read_u16 / ds_write_b16, in order
  const int mfl1 = len - kMfLimit + 1;  // mflimitPlusOne
  const int matchlimit = len - kLastLiterals;
  const int last4 = len - 4;
  int anchor = 0, op = 0;
#ifdef S3S_LZ4_TIMING
  unsigned long long dbg[20] = {0};
  struct DbgFlush {
    unsigned long long* d;
    int lane;
    __device__ ~DbgFlush() {
      if (lane == 0)
        for (int i = 0; i < 20; i++) atomicAdd(&g_lz4_dbg[i], d[i]);
    }
  } dbg_flush{dbg, lane};
#endif

  if (len >= kMfLimit + 1) {
    T[hash13(in.rd32(0))] = 0;  // LZ4_putPosition(ip = source); all lanes store the same value
    int base = 1, t0 = 1;
    uint32_t vpre = in.rd32(1 + lane < last4 ? 1 + lane : last4);  // prefetched v of the next batch
    bool have_pre = true;
    uint32_t vput = 0;  // the 4 bytes at base-2, to insert before the batch (after a match)
    bool put_pending = false;
    // fast windows: far enough from the end of the chunk that neither mflimit nor matchlimit nor
    // the end of the buffer can be met by a window's probes and speculative loads
    const int fast_limit = len - 224;
    const int pipe_limit = len - 448;  // pipelined windows prefetch up to 323 bytes ahead
    int kn = -1;         // first position of the window whose v is held in vn
    uint32_t vn = 0;
    int kp = -2;         // first position of the window whose v is held in vp (kMode 1)
    uint32_t vp = 0;
    int pk = -1000;      // first position of the window whose stream bytes are held in pw0 (kMode 4)
    uint4 pw0 = make_uint4(0, 0, 0, 0), pw1 = pw0, pw2 = pw0;
    bool force_general = false;
    for (;;) {
      if constexpr (kMode == 2) {
        // ===== pipelined exact windows: window k is resolved while the loads of k+1, k+2, k+3 fly =====
        //   stage A  v   = rd32(position)                 issued 3 windows ahead
        //   stage B  cp  = T[hash(v)], w = rd32(cp)       table snapshot + candidate bytes, 2 ahead
        //   stage C  em  = (w == v), 64+8 bytes at p and cp for the em lanes ("raw")   1 ahead
        //   stage D  info = lengths from raw                        at the start of the window's turn
        // A snapshot is validated when its window is resolved: cp' = T[h] again; a lane is stale iff
        // cp' != cp.  Everything inserted since the snapshot lies in [wbase-128, p), i.e. in the v
        // registers of this and the two previous windows, so a stale lane's match test is a
        // cross-lane read; its lengths come from the cooperative extension.
        if (!force_general && t0 <= 48 && (base & ~63) <= pipe_limit) {
          if (put_pending) T[hash13(vput)] = (uint16_t)(base - 2);  // LZ4_putPosition(ip - 2)
          put_pending = false;
          have_pre = false;
          int wbase = base & ~63;
          // ---- (re)start: fill the pipeline synchronously ---------------------------------------------
          uint32_t v0 = in.rd32(wbase + lane), v1 = in.rd32(wbase + 64 + lane);
          uint32_t v2 = in.rd32(wbase + 128 + lane), vA = in.rd32(wbase + 192 + lane);
          uint32_t vm1 = in.rd32((wbase >= 64 ? wbase - 64 : 0) + lane);
          uint32_t vm2 = in.rd32((wbase >= 128 ? wbase - 128 : 0) + lane);
          uint32_t cp0 = T[hash13(v0)], cp1 = T[hash13(v1)], cp2 = T[hash13(v2)];
          uint32_t wN;
          uint32_t info0;
          uint4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;  // raw: 64 bytes behind p+4 and behind cp+4
          uint2 rqa, rqb;                                // raw: 8 bytes in front of p and of cp
          bool rem;                                      // raw belongs to an em lane
#define S3S_STAGE_C(VV, CP, WW, WB)                                                   \
  {                                                                                   \
    rem = (WW) == (VV);                                                               \
    if (rem) {                                                                        \
      const int pa_ = (WB) + lane + kMinMatch, pb_ = (int)(CP) + kMinMatch;           \
      ra0 = in.ld16(pa_), rb0 = in.ld16(pb_);                                         \
      ra1 = in.ld16(pa_ + 16), rb1 = in.ld16(pb_ + 16);                               \
      ra2 = in.ld16(pa_ + 32), rb2 = in.ld16(pb_ + 32);                               \
      ra3 = in.ld16(pa_ + 48), rb3 = in.ld16(pb_ + 48);                               \
      const int qa_ = (WB) + lane >= 8 ? (WB) + lane - 8 : 0;                         \
      const int qb_ = (CP) >= 8u ? (int)(CP) - 8 : 0;                                 \
      rqa = in.ld8(qa_), rqb = in.ld8(qb_);                                           \
    }                                                                                 \
  }
#define S3S_STAGE_D(CP, WB, INFO)                                                     \
  {                                                                                   \
    INFO = 0u;                                                                        \
    if (rem) {                                                                        \
      uint32_t be_ = 9;                                                               \
      if ((CP) >= 8u && (WB) + lane >= 8) {                                           \
        const uint32_t xh_ = rqa.y ^ rqb.y, xl_ = rqa.x ^ rqb.x;                      \
        be_ = xh_ ? (uint32_t)(__builtin_clz(xh_) >> 3)                               \
                  : (xl_ ? 4u + (uint32_t)(__builtin_clz(xl_) >> 3) : 8u);            \
      }                                                                               \
      int fl_ = first_diff16(make_uint4(ra0.x ^ rb0.x, ra0.y ^ rb0.y, ra0.z ^ rb0.z, ra0.w ^ rb0.w)); \
      if (fl_ == 16) {                                                                \
        fl_ = 16 + first_diff16(make_uint4(ra1.x ^ rb1.x, ra1.y ^ rb1.y, ra1.z ^ rb1.z, ra1.w ^ rb1.w)); \
        if (fl_ == 32) {                                                              \
          fl_ = 32 + first_diff16(make_uint4(ra2.x ^ rb2.x, ra2.y ^ rb2.y, ra2.z ^ rb2.z, ra2.w ^ rb2.w)); \
          if (fl_ == 48)                                                              \
            fl_ = 48 + first_diff16(make_uint4(ra3.x ^ rb3.x, ra3.y ^ rb3.y, ra3.z ^ rb3.z, ra3.w ^ rb3.w)); \
        }                                                                             \
      }                                                                               \
      INFO = (CP) | ((uint32_t)fl_ << 16) | (be_ << 24) | 0x20000000u;                \
    }                                                                                 \
  }
          {
            const uint32_t w0 = in.rd32((int)cp0), w1 = in.rd32((int)cp1);
            wN = in.rd32((int)cp2);
            S3S_STAGE_C(v0, cp0, w0, wbase);
            S3S_STAGE_D(cp0, wbase, info0);
            S3S_STAGE_C(v1, cp1, w1, wbase + 64);  // raw(k+1) stays in flight
          }
          // ---- steady state: one window per iteration -------------------------------------------------
          int exit_kind = 0;  // 0: leave to the outer loop, 1: last literals, 2: the general batch takes over
          int pend_far = -1;
          for (;;) {
            DBG_ADD(8, 1);
            DBG_T(pa);
            const int p = wbase + lane;
            const uint32_t h = hash13(v0);
            const int rs0 = base - wbase;
            const bool live = lane >= rs0;
            // fresh candidates + duplicate-hash groups among the live lanes (rolled back)
            const uint32_t cpn = T[h];
            bool grp = false;
            if (live) {
              T[h] = (uint16_t)p;
              const uint32_t r1 = T[h];
              const bool lost1 = r1 != (uint32_t)p;
              if (lost1) T[h] = (uint16_t)p;
              const uint32_t r2 = T[h];
              grp = lost1 || (r2 != (uint32_t)p);
              if (r2 == (uint32_t)p) T[h] = (uint16_t)cpn;
            }
            // info: [15:0] candidate [22:16] forward 0..64 [27:24] backward 0..8 / 9 unknown
            //       [29] candidate matches [30] lengths unknown (stale snapshot) [31] suspect lane
            uint32_t info = info0;
            const bool stale = live && (cpn != cp0);
            if (__ballot(stale)) {
              const int idx = (int)(cpn & 63u);
              const int dwin = (wbase - (int)(cpn & ~63u)) >> 6;  // 0, 1 or 2 windows back
              uint32_t sv = __shfl(v0, idx);
              const uint32_t s1 = __shfl(vm1, idx), s2 = __shfl(vm2, idx);
              sv = dwin == 1 ? s1 : sv;
              sv = dwin == 2 ? s2 : sv;
              // (dwin <= 2 by construction: everything inserted since the snapshot is in these windows)
              if (stale) info = (sv == v0) ? (cpn | 0x60000000u) : 0u;
            }
            const bool em = live && ((info & 0x20000000u) != 0u);
            if (grp) info |= 0x80000000u;
            const uint64_t Ecp = __ballot(em);
            const uint64_t Dp = __ballot(grp);
            uint64_t ED = Ecp | Dp;
            DBG_T(pb);
            DBG_ADD(0, pb - pa);
            // ---- runs (scalar) ---------------------------------------------------------------------------
            uint64_t K = 0;
            int rs = rs0, rt = t0, pend_q = -1;
            const int e0 = rs0 + 66 - t0;
            int elim = e0 < kWave ? e0 : kWave;
            uint64_t runmask = e0 < kWave ? ((1ull << e0) - 1ull) : ~0ull;
            for (;;) {
              const uint64_t cm = ED & runmask & (~0ull << rs);
              if (cm == 0ull) {
                K |= runmask & (~0ull << rs);
                base = wbase + elim;
                t0 = rt + (elim - rs);
                exit_kind = elim < kWave ? 2 : 0;
                break;
              }
              const int m = __builtin_ctzll(cm);
              const uint32_t inf = __builtin_amdgcn_readlane(info, m);
              const int ip0 = wbase + m;
              int mpos = (int)(inf & 0xffffu);
              int fwd = (int)((inf >> 16) & 0x7fu);
              const int be = (int)((inf >> 24) & 0xfu);
              const int nbmax = ip0 - anchor;
              int nb = be < nbmax ? be : nbmax;
              bool need_ext = fwd >= 64 || (be >= 8 && nbmax > (be == 8 ? 8 : 0)) || (inf & 0x40000000u) != 0u;
              if (__builtin_expect((int)inf < 0, 0)) {
                const uint64_t bit = 1ull << m;
                bool is_match = (Ecp & bit) != 0ull;
                const uint32_t hv = __builtin_amdgcn_readlane(h, m);
                const uint64_t dk = __ballot(h == hv) & (bit - 1ull) & (K | (runmask & (~0ull << rs)));
                if (dk) {
                  const int d = 63 - __builtin_clzll(dk);
                  is_match = __builtin_amdgcn_readlane(v0, d) == __builtin_amdgcn_readlane(v0, m);
                  mpos = wbase + d;
                  need_ext = true;
                }
                if (!is_match) {
                  ED &= ~bit;
                  continue;
                }
              }
              if (__builtin_expect(need_ext, 0)) {
                DBG_ADD(10, 1);
                fwd = extend_match(in, ip0, mpos, anchor, matchlimit, last4, lane, nb);
              }
              DBG_ADD(9, 1);
              K |= ((2ull << m) - 1ull) & (~0ull << rs);
              const int lit = ip0 - nb - anchor, offset = ip0 - mpos, mcode = nb + fwd;
              if (__builtin_expect(anchor >= wbase && lit < 15 && mcode < 15 + 255, 1)) {
                const int total = 3 + lit + (mcode >= 15 ? 1 : 0);
                if (op + total > len) return -1;
                const int rel = (lane - (anchor - wbase)) & 63;
                uint32_t bv = v0 & 0xffu;
                int idx = rel < lit ? 1 + rel : rel;
                if (rel == lit) {
                  bv = (uint32_t)(lit << 4) | (uint32_t)(mcode < 15 ? mcode : 15);
                  idx = 0;
                }
                if (rel == lit + 1) bv = (uint32_t)offset;
                if (rel == lit + 2) bv = (uint32_t)offset >> 8;
                if (rel == lit + 3) bv = (uint32_t)(mcode - 15);
#ifndef S3S_ABL_NOSTORE
                if (rel < total) out[op + idx] = (uint8_t)bv;
#else
                asm volatile("" ::"v"(bv), "v"(idx));
#endif
                op += total;
              } else {
                op = emit_sequence(out, len, op, in, anchor, lit, true, offset, mcode, false, 0u, lane);
                if (op < 0) return -1;
              }
              const int ipe = ip0 + kMinMatch + fwd;
              anchor = ipe;
              if (__builtin_expect(ipe >= mfl1, 0)) {
                exit_kind = 1;
                break;
              }
              const int q = ipe - 2 - wbase;
              if (q < kWave) K |= 1ull << q;
              else pend_q = q + wbase;
              if (ipe >= wbase + kWave) {
                base = ipe;
                t0 = 0;
                exit_kind = 0;
                break;
              }
              rs = ipe - wbase;
              rt = 0;
              elim = kWave;
              runmask = ~0ull;
            }
            DBG_T(pc);
            DBG_ADD(3, pc - pb);
            if (exit_kind == 1) break;
            // ---- commit ---------------------------------------------------------------------------------
            uint64_t Wm = K, sus = K & Dp;
            while (sus) {
              const int i = __builtin_ctzll(sus);
              sus &= sus - 1ull;
              const uint32_t hv = __builtin_amdgcn_readlane(h, i);
              if (__ballot(h == hv) & K & ~((2ull << i) - 1ull)) Wm &= ~(1ull << i);
            }
            if ((Wm >> lane) & 1ull) T[h] = (uint16_t)p;
            if (pend_q >= 0) {
              if (pend_q < wbase + 3 * kWave) {
                const uint32_t vq = pend_q < wbase + 2 * kWave
                                        ? __builtin_amdgcn_readlane(v1, pend_q - wbase - kWave)
                                        : __builtin_amdgcn_readlane(v2, pend_q - wbase - 2 * kWave);
                T[hash13(vq)] = (uint16_t)pend_q;
                pend_q = -1;
              } else {
                pend_far = pend_q;  // beyond the pipeline: inserted after the loop (it ends here)
              }
            }
            DBG_T(pd);
            DBG_ADD(5, pd - pc);
            // ---- advance: only a step into the very next window keeps the pipeline -----------------------
            if (exit_kind != 0 || t0 > 48 || (base & ~63) != wbase + kWave || wbase + kWave > pipe_limit) break;
            wbase += kWave;
            S3S_STAGE_D(cp1, wbase, info0);              // raw(k+1) -> info of the new current window
            vm2 = vm1;
            vm1 = v0;
            v0 = v1;
            cp0 = cp1;
            v1 = v2;
            cp1 = cp2;
            S3S_STAGE_C(v1, cp1, wN, wbase + 64);        // w of the new k+1 landed long ago
            v2 = vA;
            cp2 = T[hash13(v2)];
            wN = in.rd32((int)cp2);
            vA = in.rd32(wbase + 192 + lane);
            DBG_T(pe);
            DBG_ADD(2, pe - pd);
          }
#undef S3S_STAGE_C
#undef S3S_STAGE_D
          if (exit_kind == 1) break;
          if (pend_far >= 0) T[hash13(in.rd32(pend_far))] = (uint16_t)pend_far;  // LZ4_putPosition(ip - 2)
          force_general = exit_kind == 2;
          continue;
        }
        force_general = false;
      }
      if constexpr (kMode == 3) {
        // ===== pipelined exact windows, VALU style (variant 4): as kMode 2, but every wave-uniform
        // quantity of the run loop lives in VGPRs and only branch conditions are made scalar — the CU's
        // single scalar ALU is what bounds these kernels =====
        //   stage A  v   = rd32(position)                 issued 3 windows ahead
        //   stage B  cp  = T[hash(v)], w = rd32(cp)       table snapshot + candidate bytes, 2 ahead
        //   stage C  em  = (w == v), 64+8 bytes at p and cp for the em lanes ("raw")   1 ahead
        //   stage D  info = lengths from raw                        at the start of the window's turn
        // A snapshot is validated when its window is resolved: cp' = T[h] again; a lane is stale iff
        // cp' != cp.  Everything inserted since the snapshot lies in [wbase-128, p), i.e. in the v
        // registers of this and the two previous windows, so a stale lane's match test is a
        // cross-lane read; its lengths come from the cooperative extension.
        if (!force_general && t0 <= 48 && (base & ~63) <= pipe_limit) {
          if (put_pending) T[hash13(vput)] = (uint16_t)(base - 2);  // LZ4_putPosition(ip - 2)
          put_pending = false;
          have_pre = false;
          int wbase = base & ~63;
          // ---- (re)start: fill the pipeline synchronously ---------------------------------------------
          uint32_t v0 = in.rd32(wbase + lane), v1 = in.rd32(wbase + 64 + lane);
          uint32_t v2 = in.rd32(wbase + 128 + lane), vA = in.rd32(wbase + 192 + lane);
          uint32_t vm1 = in.rd32((wbase >= 64 ? wbase - 64 : 0) + lane);
          uint32_t vm2 = in.rd32((wbase >= 128 ? wbase - 128 : 0) + lane);
          uint32_t cp0 = T[hash13(v0)], cp1 = T[hash13(v1)], cp2 = T[hash13(v2)];
          uint32_t wN;
          uint32_t info0;
          uint4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;  // raw: 64 bytes behind p+4 and behind cp+4
          uint2 rqa, rqb;                                // raw: 8 bytes in front of p and of cp
          bool rem;                                      // raw belongs to an em lane
#define S3S_STAGE_C(VV, CP, WW, WB)                                                   \
  {                                                                                   \
    rem = (WW) == (VV);                                                               \
    if (rem) {                                                                        \
      const int pa_ = (WB) + lane + kMinMatch, pb_ = (int)(CP) + kMinMatch;           \
      ra0 = in.ld16(pa_), rb0 = in.ld16(pb_);                                         \
      ra1 = in.ld16(pa_ + 16), rb1 = in.ld16(pb_ + 16);                               \
      ra2 = in.ld16(pa_ + 32), rb2 = in.ld16(pb_ + 32);                               \
      ra3 = in.ld16(pa_ + 48), rb3 = in.ld16(pb_ + 48);                               \
      const int qa_ = (WB) + lane >= 8 ? (WB) + lane - 8 : 0;                         \
      const int qb_ = (CP) >= 8u ? (int)(CP) - 8 : 0;                                 \
      rqa = in.ld8(qa_), rqb = in.ld8(qb_);                                           \
    }                                                                                 \
  }
#define S3S_STAGE_D(CP, WB, INFO)                                                     \
  {                                                                                   \
    INFO = 0u;                                                                        \
    if (rem) {                                                                        \
      uint32_t be_ = 9;                                                               \
      if ((CP) >= 8u && (WB) + lane >= 8) {                                           \
        const uint32_t xh_ = rqa.y ^ rqb.y, xl_ = rqa.x ^ rqb.x;                      \
        be_ = xh_ ? (uint32_t)(__builtin_clz(xh_) >> 3)                               \
                  : (xl_ ? 4u + (uint32_t)(__builtin_clz(xl_) >> 3) : 8u);            \
      }                                                                               \
      int fl_ = first_diff16(make_uint4(ra0.x ^ rb0.x, ra0.y ^ rb0.y, ra0.z ^ rb0.z, ra0.w ^ rb0.w)); \
      if (fl_ == 16) {                                                                \
        fl_ = 16 + first_diff16(make_uint4(ra1.x ^ rb1.x, ra1.y ^ rb1.y, ra1.z ^ rb1.z, ra1.w ^ rb1.w)); \
        if (fl_ == 32) {                                                              \
          fl_ = 32 + first_diff16(make_uint4(ra2.x ^ rb2.x, ra2.y ^ rb2.y, ra2.z ^ rb2.z, ra2.w ^ rb2.w)); \
          if (fl_ == 48)                                                              \
            fl_ = 48 + first_diff16(make_uint4(ra3.x ^ rb3.x, ra3.y ^ rb3.y, ra3.z ^ rb3.z, ra3.w ^ rb3.w)); \
        }                                                                             \
      }                                                                               \
      INFO = (CP) | ((uint32_t)fl_ << 16) | (be_ << 24) | 0x20000000u;                \
    }                                                                                 \
  }
          {
            const uint32_t w0 = in.rd32((int)cp0), w1 = in.rd32((int)cp1);
            wN = in.rd32((int)cp2);
            S3S_STAGE_C(v0, cp0, w0, wbase);
            S3S_STAGE_D(cp0, wbase, info0);
            S3S_STAGE_C(v1, cp1, w1, wbase + 64);  // raw(k+1) stays in flight
          }
          // ---- steady state: one window per iteration; wave-uniform state in VGPRs -------------------------
          int vbase = base, vt0 = t0, vanchor = anchor, vop = op;
          asm volatile("" : "+v"(vbase), "+v"(vt0), "+v"(vanchor), "+v"(vop));
          int exit_kind = 0;  // 0: leave to the outer loop, 1: last literals, 2: the general batch takes over
          int pend_far = -1;
          bool overflow = false;
          for (;;) {
            const int p = wbase + lane;
            const uint32_t h = hash13(v0);
            const int rs0 = vbase - wbase;
            const bool live = lane >= rs0;
            const uint32_t cpn = T[h];
            bool grp = false;
            if (live) {
              T[h] = (uint16_t)p;
              const uint32_t r1 = T[h];
              const bool lost1 = r1 != (uint32_t)p;
              if (lost1) T[h] = (uint16_t)p;
              const uint32_t r2 = T[h];
              grp = lost1 || (r2 != (uint32_t)p);
              if (r2 == (uint32_t)p) T[h] = (uint16_t)cpn;
            }
            uint32_t info = info0;
            const bool stale = live && (cpn != cp0);
            if (__ballot(stale)) {
              const int idx = (int)(cpn & 63u);
              const int dwin = (wbase - (int)(cpn & ~63u)) >> 6;
              uint32_t sv = __shfl(v0, idx);
              const uint32_t s1 = __shfl(vm1, idx), s2 = __shfl(vm2, idx);
              sv = dwin == 1 ? s1 : sv;
              sv = dwin == 2 ? s2 : sv;
              if (stale) info = (sv == v0) ? (cpn | 0x60000000u) : 0u;
            }
            if (grp) info |= 0x80000000u;
            bool ed = live && ((info & 0xa0000000u) != 0u);  // event lanes: candidate matches, or suspect
            // ---- runs ---------------------------------------------------------------------------------------
            bool kept = false;
            int rs = rs0, rt = vt0, pendq = -1;
            int elim = rs0 + 66 - vt0;
            elim = elim < kWave ? elim : kWave;
            for (;;) {
              const bool inrun = lane >= rs && lane < elim;
              const uint64_t cm = __ballot(ed && inrun);
              if (cm == 0ull) {  // the run leaves the window (or its consecutive part) without a match
                kept = kept || inrun;
                vbase = wbase + elim;
                vt0 = rt + (elim - rs);
                exit_kind = __builtin_amdgcn_readfirstlane((int)(elim < kWave)) ? 2 : 0;
                break;
              }
              const int m = __builtin_ctzll(cm);
              int mv = m;
              asm volatile("" : "+v"(mv));
              const uint32_t inf = (uint32_t)__shfl((int)info, mv);
              const int ip0 = wbase + mv;
              int mpos = (int)(inf & 0xffffu);
              int fwd = (int)((inf >> 16) & 0x7fu);
              const int be = (int)((inf >> 24) & 0xfu);
              const int nbmax = ip0 - vanchor;
              int nb = be < nbmax ? be : nbmax;
              int is_match = (int)((inf >> 29) & 1u);
              int need_ext = (fwd >= 64) | ((be >= 8) & (nbmax > (be == 8 ? 8 : 0))) | (int)((inf >> 30) & 1u);
              if (__builtin_amdgcn_readfirstlane((int)(inf >> 31))) {
                // suspect lane: does a kept (or earlier-in-run) lane of this window share its hash?
                const uint32_t hv = (uint32_t)__shfl((int)h, mv);
                const uint64_t dk = __ballot(h == hv && lane < m && (kept || lane >= rs));
                if (dk) {
                  const int d = 63 - __builtin_clzll(dk);
                  is_match = __builtin_amdgcn_readlane(v0, d) == __builtin_amdgcn_readlane(v0, m);
                  mpos = wbase + d;
                  need_ext = 1;
                }
              }
              if (!__builtin_amdgcn_readfirstlane(is_match)) {
                ed = ed && (lane != m);  // a plain no-match probe: the run goes on behind it
                continue;
              }
              bool ended = false;
              if (__builtin_amdgcn_readfirstlane(need_ext)) {
                const int anchors = __builtin_amdgcn_readfirstlane(vanchor);
                const int mposs = __builtin_amdgcn_readfirstlane(mpos);
                int nbs = 0;
                fwd = extend_match(in, wbase + m, mposs, anchors, matchlimit, last4, lane, nbs);
                nb = nbs;
                ended = wbase + m + kMinMatch + fwd >= mfl1;
              }
              kept = kept || (lane >= rs && lane <= m);
              const int lit = ip0 - nb - vanchor, offset = ip0 - mpos, mcode = nb + fwd;
              const int fe = (vanchor >= wbase) & (lit < 15) & (mcode < 15 + 255) & (vop + 20 <= len);
              if (__builtin_amdgcn_readfirstlane(fe)) {
                const int total = 3 + lit + (mcode >= 15 ? 1 : 0);
                const int rel = (lane - (vanchor - wbase)) & 63;
                uint32_t bv = v0 & 0xffu;
                int idx = rel < lit ? 1 + rel : rel;
                bv = rel == lit ? ((uint32_t)(lit << 4) | (uint32_t)(mcode < 15 ? mcode : 15)) : bv;
                idx = rel == lit ? 0 : idx;
                bv = rel == lit + 1 ? (uint32_t)offset : bv;
                bv = rel == lit + 2 ? ((uint32_t)offset >> 8) : bv;
                bv = rel == lit + 3 ? (uint32_t)(mcode - 15) : bv;
                if (rel < total) out[vop + idx] = (uint8_t)bv;
                vop += total;
              } else {
                const int ops = emit_sequence(out, len, __builtin_amdgcn_readfirstlane(vop), in,
                                              __builtin_amdgcn_readfirstlane(vanchor), __builtin_amdgcn_readfirstlane(lit),
                                              true, __builtin_amdgcn_readfirstlane(offset),
                                              __builtin_amdgcn_readfirstlane(mcode), false, 0u, lane);
                if (ops < 0) {
                  overflow = true;
                  break;
                }
                vop = ops;
              }
              const int ipe = ip0 + kMinMatch + fwd;
              vanchor = ipe;
              if (ended) {
                exit_kind = 1;
                break;
              }
              const int q = ipe - 2 - wbase;  // LZ4_putPosition(ip - 2)
              kept = kept || (lane == q);
              pendq = q >= kWave ? ipe - 2 : pendq;
              if (__builtin_amdgcn_readfirstlane((int)(ipe >= wbase + kWave))) {
                vbase = ipe;
                vt0 = 0;
                exit_kind = 0;
                break;
              }
              rs = ipe - wbase;
              rt = 0;
              elim = kWave;
            }
            if (overflow || exit_kind == 1) break;
            // ---- commit: the highest kept lane of every hash bucket must own the slot --------------------------
            if (kept) T[h] = (uint16_t)p;
            for (;;) {
              uint32_t r = (uint32_t)p;
              if (kept) r = T[h];
              const bool redo = kept && ((uint32_t)p > r);
              if (!__ballot(redo)) break;
              if (redo) T[h] = (uint16_t)p;
            }
            const int pq = __builtin_amdgcn_readfirstlane(pendq);
            if (pq >= 0) {
              if (pq < wbase + 3 * kWave) {
                const uint32_t vq = pq < wbase + 2 * kWave ? __builtin_amdgcn_readlane(v1, pq - wbase - kWave)
                                                           : __builtin_amdgcn_readlane(v2, pq - wbase - 2 * kWave);
                T[hash13(vq)] = (uint16_t)pq;
              } else {
                pend_far = pq;
              }
            }
            // ---- advance: only a step into the very next window keeps the pipeline ---------------------------
            const int keep_going = (vt0 <= 48) & ((vbase & ~63) == wbase + kWave);
            if (exit_kind != 0 || !__builtin_amdgcn_readfirstlane(keep_going) || wbase + kWave > pipe_limit) break;
            wbase += kWave;
            S3S_STAGE_D(cp1, wbase, info0);
            vm2 = vm1;
            vm1 = v0;
            v0 = v1;
            cp0 = cp1;
            v1 = v2;
            cp1 = cp2;
            S3S_STAGE_C(v1, cp1, wN, wbase + 64);
            v2 = vA;
            cp2 = T[hash13(v2)];
            wN = in.rd32((int)cp2);
            vA = in.rd32(wbase + 192 + lane);
          }
          base = __builtin_amdgcn_readfirstlane(vbase);
          t0 = __builtin_amdgcn_readfirstlane(vt0);
          anchor = __builtin_amdgcn_readfirstlane(vanchor);
          op = __builtin_amdgcn_readfirstlane(vop);
          if (overflow) return -1;
#undef S3S_STAGE_C
#undef S3S_STAGE_D
          if (exit_kind == 1) break;
          if (pend_far >= 0) T[hash13(in.rd32(pend_far))] = (uint16_t)pend_far;  // LZ4_putPosition(ip - 2)
          force_general = exit_kind == 2;
          continue;
        }
        force_general = false;
      }
      if constexpr (kMode == 4) {
        // ===== lean exact windows (variant 10) =====================================================================
        // Same resolution rules as kMode 1 (runs resolved with scalar mask arithmetic over K / ED), but the
        // window is prepared with ONE memory round trip instead of three:
        //   * every lane keeps the 16 bytes p-4 .. p+11 of its position for this window and the next two
        //     (linear, issued two windows ahead: never waited for in steady state);
        //   * after the table read each lane gathers the 16 bytes cp-4 .. cp+11 around its candidate; that one
        //     load decides "candidate matches", the forward length up to 8 bytes beyond MINMATCH and the
        //     backward count up to 4 — enough for the short matches of serialized rows; anything longer goes
        //     to the cooperative extension (one more round trip per LONG sequence only);
        //   * the duplicate-hash passes over the table run while that gather is in flight.
        const int wbase = base & ~63;
        if (!force_general && t0 <= 48 && wbase >= 64 && wbase <= fast_limit) {
          if (put_pending) T[hash13(vput)] = (uint16_t)(base - 2);  // LZ4_putPosition(ip - 2)
          put_pending = false;
          have_pre = false;
          DBG_T(t4a);
          DBG_ADD(8, 1);
          const int p = wbase + lane;
          if (pk != wbase) {
            if (pk + 64 == wbase) {
              vp = pw0.y;
              kp = pk;
              pw0 = pw1;
              pw1 = pw2;
              pw2 = in.ld16(p + 124);
            } else if (pk + 128 == wbase) {
              vp = pw1.y;
              kp = pk + 64;
              pw0 = pw2;
              pw1 = in.ld16(p + 60);
              pw2 = in.ld16(p + 124);
            } else {
              pw0 = in.ld16(p - 4);
              pw1 = in.ld16(p + 60);
              pw2 = in.ld16(p + 124);
            }
            pk = wbase;
          }
          const bool vp_ok = kp == wbase - 64;  // vp holds the previous window's dwords (literals may start there)
          const uint32_t v = pw0.y;
          const uint32_t h = hash13(v);
          const int rs0 = base - wbase;
          const bool live = lane >= rs0;
          const uint32_t cp = T[h];
#ifdef S3S_LZ4_TIMING
          asm volatile("" ::"v"(cp));
#endif
          DBG_T(t4b);
          const bool near0 = cp < 4u;  // (also every empty slot: it reads as position 0)
          const uint4 G = in.ld16(near0 ? 0 : (int)cp - 4);
#ifdef S3S_G2_PREFETCH
          // touch the candidate's next cache line too: a long match's cooperative extension then hits L1/L2
          const int g2p = (int)cp + 64 + 60;
          const uint32_t g2 = in.rd32(g2p < last4 ? g2p : last4);
#endif
          // duplicate-hash groups among the live lanes (two speculative store passes, rolled back) — while G flies
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
#ifdef S3S_LZ4_TIMING
          asm volatile("" ::"v"(grp));
#endif
          DBG_T(t4c0);
#ifdef S3S_LZ4_TIMING
          asm volatile("" ::"v"(G.x));
#endif
          DBG_T(t4c);
          const uint32_t w = near0 ? __builtin_amdgcn_alignbyte(G.y, G.x, cp) : G.y;
#ifdef S3S_G2_PREFETCH
          asm volatile("" ::"v"(g2));
#endif
          const bool em = live && (w == v);
          // per-lane event record: [15:0] table candidate, [19:16] forward length 0..8 (8 = at least),
          // [22:20] backward equal bytes 0..4 (4 = at least), [30] candidate matches, [31] suspect lane
          uint32_t info = grp ? 0x80000000u : 0u;
          if (em) {
            uint32_t fl = 8u, be = 0u;  // near0: lengths by the cooperative extension
            if (!near0) {
              const uint32_t xz = G.z ^ pw0.z, xw = G.w ^ pw0.w, xb = G.x ^ pw0.x;
              fl = xz ? (uint32_t)(__builtin_ctz(xz) >> 3) : (xw ? 4u + (uint32_t)(__builtin_ctz(xw) >> 3) : 8u);
              be = xb ? (uint32_t)(__builtin_clz(xb) >> 3) : 4u;
            }
            info |= cp | (fl << 16) | (be << 20) | 0x40000000u;
          }
          const uint64_t Ecp = __ballot(em);
          const uint64_t Dp = __ballot(grp);
          uint64_t ED = Ecp | Dp;  // lanes the run loop has to look at
          DBG_T(t4d);
          DBG_ADD(0, t4b - t4a);
          DBG_ADD(1, t4c0 - t4b);
          DBG_ADD(2, t4c - t4c0);
          DBG_ADD(6, t4d - t4c);
          // ================= runs (scalar work) ========================================================
          // One loop with a single hot path: next event -> (rare: suspect lane) -> (cooperative extension)
          // -> short-form emit -> advance.  Conditions are folded into sign tests of small integer
          // expressions so that hipcc does not materialise 64-bit lane masks for every boolean.
          uint64_t K = 0;      // lanes the sequential code inserts: probes and ip-2 positions
          int rs = rs0, rt = t0, pend_q = -1;
          int exit_kind = 0;   // 0: next window / batch, 1: last literals, 2: the general batch takes over
          const int e0 = rs0 + 66 - t0;  // the first run may continue an older one (t0 <= 48 => e0 >= rs0 + 18)
          int elim = e0 < kWave ? e0 : kWave;
          uint64_t valid = e0 < kWave ? ((1ull << e0) - 1ull) : ~0ull;  // consecutive probes of the current run
          const int lit_floor = vp_ok ? wbase - 64 : wbase;  // literals in registers start here
          // Cost model measured on gfx950 (tools/probe/issue_probe.hip): a plain SALU / VALU instruction costs a
          // wave ~5 cycles, a taken branch ~25, a not-taken one ~11, a VALU -> SGPR -> SALU crossing ~+20.  So the
          // hot path below is one fall-through chain: every rare case leaves the loop (exit codes) and is handled
          // behind it, then the loop is re-entered.
          enum { kNoEvent = 0, kLeft = 2 };
          int why;
          {
            DBG_T(tl0);
            for (;;) {
              const uint64_t live_m = valid & (~0ull << rs);
              const uint64_t cm = ED & live_m;
              if (cm == 0ull) { why = kNoEvent; break; }
              const int m = __builtin_ctzll(cm);
              const uint32_t inf = __builtin_amdgcn_readlane(info, m);
              const uint64_t bit = 1ull << m;
              // an earlier kept (or in-run) lane with the same table candidate may share the hash: rare path
              const uint64_t dk = __ballot(cp == (inf & 0xffffu)) & (bit - 1ull) & (K | live_m) & Dp;
              const int ip0 = wbase + m;
              int mpos = (int)(inf & 0xffffu);
              int fwd = (int)((inf >> 16) & 0xfu);
              const int be = (int)((inf >> 20) & 0x7u);
              const int nbmax = ip0 - anchor;
              int nb = be < nbmax ? be : nbmax;
              // need_ext <=> fwd >= 8 || (be >= 4 && nbmax > 4)   (a length hit its cap)
              int ext = (7 - fwd) | ((3 - be) & (4 - nbmax));
              if (__builtin_expect(dk != 0ull || (inf & 0x40000000u) == 0u, 0)) {
                // exact duplicate-hash resolution: the highest earlier kept / in-run lane with the same HASH
                DBG_ADD(15, 1);
                const uint32_t hv = __builtin_amdgcn_readlane(h, m);
                const uint64_t dh = __ballot(h == hv) & (bit - 1ull) & (K | live_m);
                bool is_match = (inf & 0x40000000u) != 0u;
                if (dh) {  // the sequential code's candidate is an earlier position of this window
                  const int d = 63 - __builtin_clzll(dh);
                  is_match = __builtin_amdgcn_readlane(v, d) == __builtin_amdgcn_readlane(v, m);
                  mpos = wbase + d;
                  ext = -1;
                }
                if (!is_match) {
                  DBG_ADD(16, 1);
                  ED &= ~bit;  // a plain no-match probe: the run goes on behind it
                  continue;
                }
              }
              if (ext < 0) {
                DBG_T(ts0);
                // cooperative extension, one round: 256 bytes forward, 64 backward (all loads issued together)
                int fp = ip0 + kMinMatch + 4 * lane;
                fp = fp < last4 ? fp : last4;
                const uint32_t x = in.rd32(fp) ^ in.rd32(fp - (ip0 - mpos));
                const int maxback = nbmax < mpos ? nbmax : mpos;
                uint32_t ba = 0, bb = 1;
                if (lane < maxback) {
                  ba = in.rd8(ip0 - 1 - lane);
                  bb = in.rd8(mpos - 1 - lane);
                }
                const uint64_t E = __ballot(ba == bb);
                const uint64_t D = __ballot(x != 0u);
                const int nbk = (int)__builtin_ctzll(~E | (1ull << 63));  // (lane 63 always stops the count)
                const int f = D ? (int)__builtin_ctzll(D) : 0;
                const uint32_t xf = __builtin_amdgcn_readlane(x, f);
                int got = D ? 4 * f + (int)(__builtin_ctz(xf) >> 3) : 4 * kWave;
                const int avail = matchlimit - (ip0 + kMinMatch);
                if (__builtin_expect((got >= 4 * kWave && avail > 4 * kWave) || nbk >= 63, 0)) {
                  fwd = extend_match(in, ip0, mpos, anchor, matchlimit, last4, lane, nb);  // longer than one round
                } else {
                  fwd = got < avail ? got : avail;
                  nb = nbk;
                }
                DBG_T(ts1);
                DBG_ADD(4, ts1 - ts0);
                DBG_ADD(10, 1);
              }
              DBG_ADD(9, 1);
              const int lit = nbmax - nb, offset = ip0 - mpos, mcode = nb + fwd;
              const int total = 3 + lit + (mcode >= 15 ? 1 : 0);
              const int ipe = ip0 + kMinMatch + fwd;
              // short form <=> anchor >= lit_floor && lit < 15 && mcode < 270 && op + total <= len
              if (__builtin_expect(((anchor - lit_floor) | (14 - lit) | (269 - mcode) | (len - op - total)) < 0, 0)) {
                op = emit_sequence(out, len, op, in, anchor, lit, true, offset, mcode, false, 0u, lane);
                if (op < 0) return -1;
              } else {
                // every literal is the low byte of a lane's v (this window) or vp (the previous one): lane L
                // stores its own byte, four otherwise idle lanes store token / offset / match-length byte
#ifdef S3S_ABL_NOEMIT
                if (false) {
#else
                {
#endif
                const uint32_t rel = (uint32_t)(lane - anchor) & 63u;
                const uint32_t tok = (uint32_t)(lit << 4) | (uint32_t)(mcode < 15 ? mcode : 15);
                const uint32_t d = rel - (uint32_t)lit;  // 0: token, 1/2: offset, 3: match-length byte
                uint32_t bv = (((int)(anchor + rel) < wbase) ? vp : v) & 0xffu;
                bv = d == 0u ? tok : bv;
                bv = d == 1u ? (uint32_t)offset : bv;
                bv = d == 2u ? (uint32_t)offset >> 8 : bv;
                bv = d == 3u ? (uint32_t)(mcode - 15) : bv;
                const uint32_t idx = d == 0u ? 0u : rel + (rel < (uint32_t)lit ? 1u : 0u);
#if defined(S3S_ABL_NOSTORE)
                asm volatile("" ::"v"(bv), "v"(idx));
#else
                if (rel < (uint32_t)total) out[(uint32_t)op + idx] = (uint8_t)bv;
#endif
                }
                op += total;
              }
              K |= live_m & ((bit << 1) - 1ull);
              anchor = ipe;
              const int q = ipe - 2 - wbase;  // LZ4_putPosition(ip - 2)
              K |= q < kWave ? (1ull << (q & 63)) : 0ull;
              pend_q = q < kWave ? pend_q : q + wbase;
              rs = ipe - wbase;
              rt = 0;
              elim = kWave;
              valid = ~0ull;
              if (rs >= kWave) { why = kLeft; break; }
            }
            DBG_T(tl1);
            DBG_ADD(12, tl1 - tl0);
            if (why == kNoEvent) {  // the run leaves the window (or its consecutive part) without a match
              K |= valid & (~0ull << rs);
              base = wbase + elim;
              t0 = rt + (elim - rs);
              exit_kind = elim < kWave ? 2 : 0;
            } else {  // the match left the window
              base = anchor;
              t0 = 0;
              exit_kind = anchor >= mfl1 ? 1 : 0;
            }
          }
          DBG_T(t4e);
          DBG_ADD(3, t4e - t4d);
          if (exit_kind == 1) break;
          // ================= commit: the highest kept lane of every hash group writes ==================
          const bool kept = ((K >> lane) & 1ull) != 0ull;
          if (kept) T[h] = (uint16_t)p;
          if (K & Dp) {
            for (;;) {
              const bool redo = kept && (uint32_t)T[h] < (uint32_t)p;
              if (!__ballot(redo)) break;
              if (redo) T[h] = (uint16_t)p;
            }
          }
          if (pend_q >= 0) {  // the put happens at the top of the next iteration (base - 2 == pend_q)
            const int dq = pend_q - wbase;
            vput = dq < 2 * kWave ? __builtin_amdgcn_readlane(pw1.y, dq - kWave)
                                  : (dq < 3 * kWave ? __builtin_amdgcn_readlane(pw2.y, dq - 2 * kWave)
                                                    : in.rd32(pend_q));
            put_pending = true;
          }
          force_general = exit_kind == 2;
          DBG_T(t4f);
          DBG_ADD(5, t4f - t4e);
          continue;
        }
        force_general = false;
      }
      if constexpr (kMode == 1) {
        const int wbase = base & ~63;
        if (!force_general && t0 <= 48 && wbase <= fast_limit) {
          if (put_pending) T[hash13(vput)] = (uint16_t)(base - 2);  // LZ4_putPosition(ip - 2)
          put_pending = false;
          have_pre = false;
          // ================= window preparation (vector work, every lane = one position) ============
          DBG_T(tw0);
          DBG_ADD(8, 1);
          const int p = wbase + lane;
          const uint32_t v = (kn == wbase) ? vn : in.rd32(p);
          const bool vp_ok = kp == wbase - 64;  // vp holds the previous window's dwords (literals may start there)
          vn = in.rd32(p + 64);
          kn = wbase + 64;
          const uint32_t h = hash13(v);
          const int rs0 = base - wbase;
          const bool live = lane >= rs0;
          // table candidates, and duplicate-hash groups among the live lanes (two speculative store
          // passes, rolled back: lanes of a group of >= 2 either lose pass 1 or see pass 2's winner)
          const uint32_t cp = T[h];
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
          DBG_T(tw1);
          const uint32_t w = in.rd32((int)cp);
          const bool em = live && (w == v);
          DBG_T(tw2x);
          DBG_ADD(1, tw2x - tw1 + (__ballot(em) & 0));
          DBG_T(tw2);
          // per-lane event record: [15:0] table candidate, [22:16] forward length 0..64,
          // [27:24] backward equal bytes 0..8 (8 = at least, 9 = unknown), [31] suspect lane (member of
          // a duplicate-hash group)
          uint32_t info = grp ? 0x80000000u : 0u, info2 = 0u;
          if (em) {
            const int pa = p + kMinMatch, pb = (int)cp + kMinMatch;
            const uint4 a0 = in.ld16(pa), b0 = in.ld16(pb);
            const uint4 a1 = in.ld16(pa + 16), b1 = in.ld16(pb + 16);
            const uint4 a2 = in.ld16(pa + 32), b2 = in.ld16(pb + 32);
            const uint4 a3 = in.ld16(pa + 48), b3 = in.ld16(pb + 48);
            uint32_t be = 9;  // 9 = unknown (too close to the start of the chunk), 8 = at least 8
            if (cp >= 8u && p >= 8) {
              const uint2 qa = in.ld8(p - 8), qb = in.ld8((int)cp - 8);
              const uint32_t xh = qa.y ^ qb.y, xl = qa.x ^ qb.x;
              be = xh ? (uint32_t)(__builtin_clz(xh) >> 3) : (xl ? 4u + (uint32_t)(__builtin_clz(xl) >> 3) : 8u);
            }
            int fl = first_diff16(make_uint4(a0.x ^ b0.x, a0.y ^ b0.y, a0.z ^ b0.z, a0.w ^ b0.w));
            if (fl == 16) {
              fl = 16 + first_diff16(make_uint4(a1.x ^ b1.x, a1.y ^ b1.y, a1.z ^ b1.z, a1.w ^ b1.w));
              if (fl == 32) {
                fl = 32 + first_diff16(make_uint4(a2.x ^ b2.x, a2.y ^ b2.y, a2.z ^ b2.z, a2.w ^ b2.w));
                if (fl == 48) fl = 48 + first_diff16(make_uint4(a3.x ^ b3.x, a3.y ^ b3.y, a3.z ^ b3.z, a3.w ^ b3.w));
              }
            }
            info = cp | ((uint32_t)fl << 16) | (be << 24) | (grp ? 0x80000000u : 0u);
            // for the straight-line steps: backward bytes to take (0..8) and the longest literal run for which
            // that count is exact (be < 8: any; be == 8: 8; unknown: 0)
            info2 = (be <= 8u ? be : 0u) | ((be < 8u ? 0x7fffu : (be == 8u ? 8u : 0u)) << 4);
          }
          // vn (issued before every load above) has landed by now: pin it here, before the emit stores,
          // so that no later use has to drain the in-order vmcnt queue behind those stores
          asm volatile("" : "+v"(vn));
          const uint64_t Ecp = __ballot(em);
          const uint64_t Dp = __ballot(grp);
          uint64_t ED = Ecp | Dp;  // lanes the run loop has to look at
          DBG_T(tw3);
          DBG_ADD(0, tw1 - tw0);
          DBG_ADD(2, tw3 - tw2 + (__builtin_amdgcn_readfirstlane(info) & 0));
          // ================= runs (scalar work) ========================================================
          uint64_t K = 0;      // lanes the sequential code inserts: probes and ip-2 positions
          int rs = rs0, rt = t0, pend_q = -1;
          int exit_kind;       // 0: next window / batch, 1: last literals, 2: the general batch takes over
          // the first run may continue an older one: its probes are consecutive only up to lane e0-1
          const int e0 = rs0 + 66 - t0;  // t0 <= 48  =>  e0 >= rs0 + 18
          int elim = e0 < kWave ? e0 : kWave;
          uint64_t runmask = e0 < kWave ? ((1ull << e0) - 1ull) : ~0ull;
          // ---- straight-line steps: up to three "plain" sequences (the first event of the run has its true candidate
          // in the table, exact lengths, short-form encoding, literals in the registers of this or the previous
          // window, not the end of the chunk) without the generic loop's control flow: ONE branch decides,
          // everything else is arithmetic.  Anything else falls through to the generic loop below, which
          // continues from whatever state the steps left.
          constexpr int kFastSteps = S3S_FAST_STEPS;
          bool left_window = false;
          const int lit_floor = vp_ok ? wbase - 64 : wbase;  // literals must start at or after this position
#pragma unroll
          for (int step = 0; step < kFastSteps; step++) {
            const uint64_t live_m = runmask & (~0ull << rs);
            const uint64_t cm = ED & live_m;
            if (cm == 0ull) break;  // no event left for this run: the generic loop's exit code handles it
            const int m = __builtin_ctzll(cm);
            const uint64_t bit = 1ull << m;
            const uint32_t inf = __builtin_amdgcn_readlane(info, m);
            const uint32_t inf2 = __builtin_amdgcn_readlane(info2, m);
            // a duplicate-hash lane is plain iff no earlier kept lane of the window shares its hash
            const uint32_t hv = __builtin_amdgcn_readlane(h, m);
            const uint64_t dk = __ballot(h == hv) & (bit - 1ull) & (K | live_m);
            const int ip0 = wbase + m;
            const int mpos = (int)(inf & 0xffffu), fwd = (int)((inf >> 16) & 0x7fu);
            const int nbmax = ip0 - anchor;
            const int nbv = (int)(inf2 & 0xfu);
            const int nb = nbv < nbmax ? nbv : nbmax;
            const int lit = nbmax - nb, mcode = nb + fwd;
            const int total = 3 + lit + (mcode >= 15 ? 1 : 0);
            const int ipe = ip0 + kMinMatch + fwd;
            // largest of the "how far beyond its limit" terms: <= 0 iff every condition holds
            int over = fwd - 63;                                       // forward length capped
            over = over > nbmax - (int)(inf2 >> 4) ? over : nbmax - (int)(inf2 >> 4);  // backward count not exact
            over = over > lit_floor - anchor ? over : lit_floor - anchor;  // literals older than the registers
            over = over > lit - 14 ? over : lit - 14;                  // literal length needs extra bytes
            over = over > mcode - 269 ? over : mcode - 269;            // more than one match-length byte
            over = over > ipe - mfl1 + 1 ? over : ipe - mfl1 + 1;      // end of the chunk
            over = over > op + total - len ? over : op + total - len;  // would not fit: the frame is stored RAW
            if ((Ecp & bit) == 0ull || dk != 0ull || over > 0) break;
            DBG_ADD(9, 1);
            K |= live_m & ((bit << 1) - 1ull);
            {
              const int offset = ip0 - mpos;
              const int rel = (lane - anchor) & 63;
              uint32_t bv = ((anchor + rel < wbase) ? vp : v) & 0xffu;
              int idx = rel < lit ? 1 + rel : rel;
              if (rel == lit) {
                bv = (uint32_t)(lit << 4) | (uint32_t)(mcode < 15 ? mcode : 15);
                idx = 0;
              }
              if (rel == lit + 1) bv = (uint32_t)offset;
              if (rel == lit + 2) bv = (uint32_t)offset >> 8;
              if (rel == lit + 3) bv = (uint32_t)(mcode - 15);
              if (rel < total) out[op + idx] = (uint8_t)bv;
            }
            op += total;
            anchor = ipe;
            const int q = ipe - 2 - wbase;  // LZ4_putPosition(ip - 2)
            K |= q < kWave ? (1ull << (q & 63)) : 0ull;
            pend_q = q < kWave ? pend_q : q + wbase;
            rs = ipe - wbase;  // (>= 64 when the match leaves the window: not used then)
            rt = 0;
            elim = kWave;
            runmask = ~0ull;
            if (ipe >= wbase + kWave) {
              base = ipe;
              t0 = 0;
              left_window = true;
              break;
            }
          }
          exit_kind = 0;
          if (!left_window) for (;;) {
            const uint64_t cm = ED & runmask & (~0ull << rs);
            if (cm == 0ull) {  // the run leaves the window (or its consecutive part) without a match
              K |= runmask & (~0ull << rs);
              base = wbase + elim;
              t0 = rt + (elim - rs);
              exit_kind = elim < kWave ? 2 : 0;
              break;
            }
            const int m = __builtin_ctzll(cm);
            const uint32_t inf = __builtin_amdgcn_readlane(info, m);
            const int ip0 = wbase + m;
            int mpos = (int)(inf & 0xffffu);
            int fwd = (int)((inf >> 16) & 0x7fu);
            const int be = (int)((inf >> 24) & 0xfu);
            const int nbmax = ip0 - anchor;  // (the table candidate of a lane with known be is >= 8)
            int nb = be < nbmax ? be : nbmax;
            bool need_ext = fwd >= 64 || (be >= 8 && nbmax > (be == 8 ? 8 : 0));  // a length hit its cap
            if (__builtin_expect((int)inf < 0, 0)) {
              // ---- suspect lane: another live lane of the window has the same hash ----------------------
              const uint64_t bit = 1ull << m;
              bool is_match = (Ecp & bit) != 0ull;
              const uint32_t hv = __builtin_amdgcn_readlane(h, m);
              const uint64_t dk = __ballot(h == hv) & (bit - 1ull) & (K | (runmask & (~0ull << rs)));
              if (dk) {  // the sequential code's candidate is an earlier position of this window
                const int d = 63 - __builtin_clzll(dk);
                is_match = __builtin_amdgcn_readlane(v, d) == __builtin_amdgcn_readlane(v, m);
                mpos = wbase + d;
                need_ext = true;
              }
              if (!is_match) {
                ED &= ~bit;  // a plain no-match probe: the run goes on behind it
                continue;
              }
            }
            if (__builtin_expect(need_ext, 0)) {
              DBG_T(ts0);
              fwd = extend_match(in, ip0, mpos, anchor, matchlimit, last4, lane, nb);
              DBG_T(ts1);
              DBG_ADD(4, ts1 - ts0);
              DBG_ADD(10, 1);
            }
            DBG_ADD(9, 1);
            K |= ((2ull << m) - 1ull) & (~0ull << rs);
            const int lit = ip0 - nb - anchor, offset = ip0 - mpos, mcode = nb + fwd;
            if (__builtin_expect(anchor >= wbase && lit < 15 && mcode < 15 + 255, 1)) {
              // every literal is the low byte of a lane's v: lane L stores its own byte, four
              // otherwise idle lanes store token / offset / match-length byte — one store
              const int total = 3 + lit + (mcode >= 15 ? 1 : 0);
              if (op + total > len) return -1;
              const int rel = (lane - (anchor - wbase)) & 63;
              uint32_t bv = v & 0xffu;
              int idx = rel < lit ? 1 + rel : rel;  // literals follow the token; the rest is in place
              if (rel == lit) {
                bv = (uint32_t)(lit << 4) | (uint32_t)(mcode < 15 ? mcode : 15);
                idx = 0;
              }
              if (rel == lit + 1) bv = (uint32_t)offset;
              if (rel == lit + 2) bv = (uint32_t)offset >> 8;
              if (rel == lit + 3) bv = (uint32_t)(mcode - 15);
              if (rel < total) out[op + idx] = (uint8_t)bv;
              op += total;
            } else {
              op = emit_sequence(out, len, op, in, anchor, lit, true, offset, mcode, false, 0u, lane);
              if (op < 0) return -1;
            }
            const int ipe = ip0 + kMinMatch + fwd;
            anchor = ipe;
            if (__builtin_expect(ipe >= mfl1, 0)) {
              exit_kind = 1;
              break;
            }
            const int q = ipe - 2 - wbase;  // LZ4_putPosition(ip - 2)
            if (q < kWave) K |= 1ull << q;
            else pend_q = q + wbase;
            if (ipe >= wbase + kWave) {
              base = ipe;
              t0 = 0;
              exit_kind = 0;
              break;
            }
            rs = ipe - wbase;
            rt = 0;
            elim = kWave;
            runmask = ~0ull;
          }
          DBG_T(tw4);
          DBG_ADD(3, tw4 - tw3);
          if (exit_kind == 1) break;
          // ================= commit: the highest kept lane of every hash group writes ==================
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
            const uint32_t vq = pend_q < wbase + 2 * kWave
                                    ? __builtin_amdgcn_readlane(vn, pend_q - wbase - kWave)
                                    : in.rd32(pend_q);
            T[hash13(vq)] = (uint16_t)pend_q;
          }
          vp = v;
          kp = wbase;
          force_general = exit_kind == 2;
          DBG_T(tw5);
          DBG_ADD(5, tw5 - tw4);
          continue;
        }
        force_general = false;
      }
      DBG_ADD(11, 1);
      // ---- one batch: lane i evaluates probe t0+i of the current no-match run ----------------
      int pos, nvalid;
      if (t0 <= 2) {  // probes 0..65 of a run are consecutive bytes
        pos = base + lane;
        // lane valid iff its successor position <= mflimitPlusOne; the post-match probe (t = 0)
        // is only reached with base < mflimitPlusOne, so the same bound covers it
        nvalid = mfl1 - base;
      } else {
        const int S0 = sched_S(t0);
        pos = base + sched_S(t0 + lane) - S0;
        const int nextpos = base + sched_S(t0 + lane + 1) - S0;
        nvalid = __popcll(__ballot(nextpos <= mfl1));  // valid lanes form a prefix
      }
      nvalid = nvalid < kWave ? nvalid : kWave;
      const bool valid = lane < nvalid;
      if (put_pending) T[hash13(vput)] = (uint16_t)(base - 2);  // LZ4_putPosition(ip - 2)
      put_pending = false;
      const uint32_t v = have_pre ? vpre : in.rd32(pos < last4 ? pos : last4);
      have_pre = false;
      const uint32_t h = hash13(v);
      uint32_t c = 0, r = (uint32_t)pos;
      if (valid) {
        c = T[h];              // candidate as of the start of the batch
        T[h] = (uint16_t)pos;  // speculative insert, all lanes at once
        r = T[h];              // readback: did this lane own its slot?
      }
      const uint32_t w = in.rd32((int)c);
      const uint32_t vprev = __builtin_amdgcn_update_dpp(~v, v, 0x138 /*wave_shr:1*/, 0xf, 0xf, false);
      const uint64_t L = __ballot(r != (uint32_t)pos);              // lost its slot
      const uint64_t M = __ballot(valid && w == v);                 // start-of-batch candidate matches
      const uint64_t A = __ballot(valid && lane > 0 && v == vprev); // repeats the previous probe
      // Clean prefix [0,B): lanes whose start-of-batch candidate is what the sequential code reads.
      // Everything below the smallest loser c0 is clean; c0 itself is clean iff its slot's winner is
      // a LATER lane (an earlier member of its hash group would have lost too).
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
        m = lim;  // first non-clean lane repeats its clean predecessor: a certain match at offset step
        adj = true;
        keep = lim + 1;
      }
      // table fix-up: winners the sequential code never reached restore the old entry, then
      // committed losers re-insert (c0, unless the adjacent match lane overrides the same slot)
      if (valid && lane >= keep && r == (uint32_t)pos) T[h] = (uint16_t)c;
      const bool redo_c0 = clean0 && c0 < keep && !(adj && c0 == m - 1);
      if ((redo_c0 && lane == c0) || (adj && lane == m)) T[h] = (uint16_t)pos;

      if (m < 0) {
        if (lim == nvalid && nvalid < kWave) break;  // ran into mflimit: last literals
        base += sched_S(t0 + lim) - sched_S(t0);     // continue the run at lane `lim`
        t0 += lim;
        continue;
      }

      // ---- match at lane m --------------------------------------------------------------------
      const int ip0 = (int)__builtin_amdgcn_readlane((uint32_t)pos, m);
      const int match0 = adj ? (int)__builtin_amdgcn_readlane((uint32_t)pos, m - 1)
                             : (int)__builtin_amdgcn_readlane(c, m);
      int nb = 0;
      const int fwd = extend_match(in, ip0, match0, anchor, matchlimit, last4, lane, nb);
      const int ip = ip0 - nb, match = match0 - nb;
      const int ipe = ip0 + kMinMatch + fwd;  // first byte after the match
      // prefetch what the next batch needs while the sequence is being written out
      if (ipe < mfl1) {
        vpre = in.rd32(ipe + lane < last4 ? ipe + lane : last4);
        vput = in.rd32(ipe - 2);
        have_pre = true;
        put_pending = true;
      }
      const int mcode = nb + fwd;  // bytes beyond MINMATCH, counted from the moved-back ip
      // literals: when the run started with this batch (post-match probe at `anchor`), output
      // byte k (1 <= k <= lit) is the low byte of probe k-1, i.e. of the lane to the left
      const bool lit_in_regs = (t0 == 0) && (anchor == base);
      op = emit_sequence(out, len, op, in, anchor, ip - anchor, true, ip - match, mcode,
                         lit_in_regs, vprev & 0xffu, lane);
      if (op < 0) return -1;
      anchor = ipe;
      if (ipe >= mfl1) break;  // end of chunk
      base = ipe;  // next batch starts with the "test next position" probe (t = 0)
      t0 = 0;
    }
  }
  return emit_sequence(out, len, op, in, anchor, len - anchor, false, 0, 0, false, 0u, lane);
}

// LZ4Block frame header + item size, written by lanes 0..20 of the parsing wave
__device__ __forceinline__ void finish_frame(uint8_t* slot, int len, int clen, uint32_t check,
                                             int level, uint32_t* item_size_out, int lane) {
  // LZ4BlockOutputStream.flushBufferedData(): compressedLength >= o  => stored raw
  const bool raw = clen < 0 || clen >= len;
  const uint32_t plen = raw ? (uint32_t)len : (uint32_t)clen;
  const uint32_t token = (raw ? 0x10u : 0x20u) | ((uint32_t)level & 0x0Fu);
  uint8_t* hdr = slot + (kSlotHeader - kLz4FrameHeader);
  if (lane < kLz4FrameHeader) {
    const uint64_t magic = 0x6b636f6c42345a4cull;  // "LZ4Block" little-endian
    uint32_t b;
    if (lane < 8) b = (uint32_t)(magic >> (8 * lane));
    else if (lane == 8) b = token;
    else if (lane < 13) b = plen >> (8 * (lane - 9));
    else if (lane < 17) b = (uint32_t)len >> (8 * (lane - 13));
    else b = (check & 0x0FFFFFFFu) >> (8 * (lane - 17));
    hdr[lane] = (uint8_t)b;
  }
  if (lane == 0) *item_size_out = (kLz4FrameHeader + plen) | (raw ? kRawFlag : 0u);
}

// ---- variant A: chunk staged in LDS --------------------------------------------------------
struct __attribute__((aligned(16))) Lz4LdsA {
  uint8_t in[kMaxBlock + kLdsSlack];
  uint16_t table[8192];
};

__global__ __launch_bounds__(kWave) void lz4_compress_lds_kernel(
    const uint8_t* __restrict__ src, const Item* __restrict__ items, int32_t n_items,
    const uint32_t* __restrict__ item_check, uint8_t* __restrict__ slots,
    uint32_t* __restrict__ item_size) {
  __shared__ Lz4LdsA s;
  const int it = blockIdx.x;
  if (it >= n_items) return;
  const Item item = items[it];
  const int kind = item.kind & 0xff;
  const int lane = threadIdx.x;
  if (kind != kItemLz4Chunk) {
    if (lane == 0 && kind == kItemLz4End) item_size[it] = kLz4FrameHeader;
    return;
  }
  const int len = item.len;
  const uint8_t* g = src + item.src_off;
  for (int i = lane * 16; i + 16 <= len; i += kWave * 16) {
    uint4 x;
    __builtin_memcpy(&x, g + i, 16);
    *reinterpret_cast<uint4*>(s.in + i) = x;
  }
  for (int i = (len & ~15) + lane; i < len; i += kWave) s.in[i] = g[i];
  {
    uint4* tz = reinterpret_cast<uint4*>(s.table);
    for (int i = lane; i < (int)(sizeof(s.table) / 16); i += kWave) tz[i] = make_uint4(0, 0, 0, 0);
  }
  __syncthreads();
  uint8_t* slot = slots + (size_t)item.chunk * kSlotBytes;
  const int clen = lz4_compress_wave<SrcLds, 0>(SrcLds{s.in}, (lds_u16*)s.table, len, slot + kSlotHeader, lane);
  finish_frame(slot, len, clen, item_check[it], item.kind >> 8, item_size + it, lane);
}

constexpr uint32_t XXP1 = 2654435761u, XXP2 = 2246822519u, XXP3 = 3266489917u,
                   XXP4 = 668265263u, XXP5 = 374761393u;

// xxHash32 of g[0,len) by one wavefront: the chunk is streamed 256 bytes per coalesced load, the four
// stripe accumulators live in lanes 0..3 and pull their words out of the block by cross-lane reads.
// (Fused into the compress kernel it also brings the chunk into L2 right before the parse.)
__device__ __forceinline__ uint32_t xxh32_wave(const uint8_t* g, int len, uint32_t seed, int lane) {
  uint32_t acc = lane == 0 ? seed + XXP1 + XXP2 : lane == 1 ? seed + XXP2 : lane == 2 ? seed : seed - XXP1;
  const int stripes = len >> 4, nblk = len >> 8;
  auto ld32u = [&](int byte_pos) -> uint32_t {
    uint32_t x;
    __builtin_memcpy(&x, g + byte_pos, 4);
    return x;
  };
  uint32_t cur = nblk > 0 ? ld32u(4 * lane) : 0u;
  for (int bk = 0; bk < nblk; bk++) {
    const uint32_t nxt = bk + 1 < nblk ? ld32u(256 * (bk + 1) + 4 * lane) : 0u;
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const uint32_t wv = (uint32_t)__shfl((int)cur, 4 * j + (lane & 3));
      acc = rotl32(acc + wv * XXP2, 13) * XXP1;
    }
    cur = nxt;
  }
  for (int j = nblk * 16; j < stripes; j++) acc = rotl32(acc + ld32u(16 * j + 4 * (lane & 3)) * XXP2, 13) * XXP1;
  uint32_t h;
  if (len >= 16) {
    const uint32_t v1 = __builtin_amdgcn_readlane(acc, 0), v2 = __builtin_amdgcn_readlane(acc, 1),
                   v3 = __builtin_amdgcn_readlane(acc, 2), v4 = __builtin_amdgcn_readlane(acc, 3);
    h = rotl32(v1, 1) + rotl32(v2, 7) + rotl32(v3, 12) + rotl32(v4, 18);
  } else {
    h = seed + XXP5;
  }
  h += (uint32_t)len;
  int p = stripes << 4;
  for (; p + 4 <= len; p += 4) h = rotl32(h + ld32u(p) * XXP3, 17) * XXP4;
  for (; p < len; p++) h = rotl32(h + (uint32_t)g[p] * XXP5, 11) * XXP1;
  h ^= h >> 15;
  h *= XXP2;
  h ^= h >> 13;
  h *= XXP3;
  h ^= h >> 16;
  return h;
}

// ---- variant B: chunk read in place, only the table in LDS ----------------------------------
template <int kMode, bool kFusedHash = false, int kLdsPad = 0>
__global__ __launch_bounds__(kWave) void lz4_compress_l2_kernel(
    const uint8_t* __restrict__ src, const Item* __restrict__ items, int32_t n_items,
    const uint32_t* __restrict__ item_check, uint8_t* __restrict__ slots,
    uint32_t* __restrict__ item_size) {
  __shared__ __attribute__((aligned(16))) uint16_t table[8192 + kLdsPad / 2];  // kLdsPad: occupancy experiments
  const int it = blockIdx.x;
  if (it >= n_items) return;
  const Item item = items[it];
  const int kind = item.kind & 0xff;
  const int lane = threadIdx.x;
  if (kind != kItemLz4Chunk) {
    if (lane == 0 && kind == kItemLz4End) item_size[it] = kLz4FrameHeader;
    return;
  }
  {
    uint4* tz = reinterpret_cast<uint4*>(table);
    for (int i = lane; i < (int)(sizeof(table) / 16); i += kWave) tz[i] = make_uint4(0, 0, 0, 0);
  }
  __syncthreads();
  uint8_t* slot = slots + (size_t)item.chunk * kSlotBytes;
  uint32_t check;
  if constexpr (kFusedHash) check = xxh32_wave(src + item.src_off, item.len, kLz4BlockSeed, lane);
  else check = item_check[it];
  const int clen = lz4_compress_wave<SrcGlobal, kMode>(SrcGlobal{src + item.src_off}, (lds_u16*)table,
                                                      item.len, slot + kSlotHeader, lane);
  finish_frame(slot, item.len, clen, check, item.kind >> 8, item_size + it, lane);
}

// ---- xxHash32 of every chunk: 4 lanes per chunk (one per stripe accumulator) ----------------
constexpr int kXxhThreads = 256;

__device__ __forceinline__ uint32_t ld32u(const uint8_t* p) {
  uint32_t v;
  __builtin_memcpy(&v, p, 4);
  return v;
}

__global__ __launch_bounds__(kXxhThreads) void xxh32_items_kernel(
    const uint8_t* __restrict__ src, const Item* __restrict__ items, int32_t n_items,
    uint32_t seed, uint32_t* __restrict__ item_check) {
  const int it = blockIdx.x * (kXxhThreads / 4) + (threadIdx.x >> 2);
  const int l = threadIdx.x & 3;
  int len = 0;
  const uint8_t* g = src;
  bool chunk = false;
  if (it < n_items) {
    const Item item = items[it];
    const int kind = item.kind & 0xff;
    chunk = (kind == kItemLz4Chunk);
    if (chunk) {
      len = item.len;
      g = src + item.src_off;
    }
  }
  uint32_t acc = l == 0 ? seed + XXP1 + XXP2 : l == 1 ? seed + XXP2 : l == 2 ? seed : seed - XXP1;
  const int stripes = len >> 4;
  const uint8_t* q = g + 4 * l;
#pragma unroll 8
  for (int j = 0; j < stripes; j++) acc = rotl32(acc + ld32u(q + 16 * j) * XXP2, 13) * XXP1;
  // gather the group's four accumulators (all lanes execute the shuffles)
  const int g0 = (threadIdx.x & 63) & ~3;
  const uint32_t v1 = __shfl(acc, g0), v2 = __shfl(acc, g0 + 1), v3 = __shfl(acc, g0 + 2),
                 v4 = __shfl(acc, g0 + 3);
  if (!chunk || l != 0) return;
  uint32_t h = len >= 16 ? rotl32(v1, 1) + rotl32(v2, 7) + rotl32(v3, 12) + rotl32(v4, 18)
                         : seed + XXP5;
  h += (uint32_t)len;
  int p = stripes << 4;
  for (; p + 4 <= len; p += 4) h = rotl32(h + ld32u(g + p) * XXP3, 17) * XXP4;
  for (; p < len; p++) h = rotl32(h + (uint32_t)g[p] * XXP5, 11) * XXP1;
  h ^= h >> 15;
  h *= XXP2;
  h ^= h >> 13;
  h *= XXP3;
  h ^= h >> 16;
  item_check[it] = h;
}

// one wavefront per chunk, coalesced streaming (xxh32_wave): ~3x the 4-lanes-per-chunk kernel above
__global__ __launch_bounds__(kWave) void xxh32_items_wave_kernel(
    const uint8_t* __restrict__ src, const Item* __restrict__ items, int32_t n_items, uint32_t seed,
    uint32_t* __restrict__ item_check) {
  const int it = blockIdx.x;
  if (it >= n_items) return;
  const Item item = items[it];
  if ((item.kind & 0xff) != kItemLz4Chunk) return;
  const int lane = threadIdx.x;
  const uint32_t h = xxh32_wave(src + item.src_off, item.len, seed, lane);
  if (lane == 0) item_check[it] = h;
}

}  // namespace

void launch_lz4_compress(const uint8_t* d_src, const Item* d_items, int32_t n_items,
                         uint32_t* d_item_check, uint8_t* d_slots, uint32_t* d_item_size,
                         int variant, hipStream_t st, hipEvent_t after_hash) {
  if (n_items <= 0) {
    if (after_hash) hipEventRecord(after_hash, st);
    return;
  }
  if (variant != 5)  // variant 5 computes the frame check inside the compress kernel
    hipLaunchKernelGGL(xxh32_items_wave_kernel, dim3((unsigned)n_items), dim3(kWave), 0, st, d_src, d_items,
                       n_items, kLz4BlockSeed, d_item_check);
  if (after_hash) hipEventRecord(after_hash, st);
  if (variant == 0)
    hipLaunchKernelGGL(lz4_compress_lds_kernel, dim3((unsigned)n_items), dim3(kWave), 0, st, d_src,
                       d_items, n_items, d_item_check, d_slots, d_item_size);
  else if (variant == 1)
    hipLaunchKernelGGL(lz4_compress_l2_kernel<0>, dim3((unsigned)n_items), dim3(kWave), 0, st, d_src,
                       d_items, n_items, d_item_check, d_slots, d_item_size);
  else if (variant == 2)
    hipLaunchKernelGGL(lz4_compress_l2_kernel<1>, dim3((unsigned)n_items), dim3(kWave), 0, st, d_src,
                       d_items, n_items, d_item_check, d_slots, d_item_size);
  else if (variant == 3)
    hipLaunchKernelGGL(lz4_compress_l2_kernel<2>, dim3((unsigned)n_items), dim3(kWave), 0, st, d_src,
                       d_items, n_items, d_item_check, d_slots, d_item_size);
  else if (variant == 4)
    hipLaunchKernelGGL(lz4_compress_l2_kernel<3>, dim3((unsigned)n_items), dim3(kWave), 0, st, d_src,
                       d_items, n_items, d_item_check, d_slots, d_item_size);
  else if (variant == 10)
    hipLaunchKernelGGL(lz4_compress_l2_kernel<4>, dim3((unsigned)n_items), dim3(kWave), 0, st, d_src,
                       d_items, n_items, d_item_check, d_slots, d_item_size);
  else if (variant == 5)
    hipLaunchKernelGGL((lz4_compress_l2_kernel<0, true>), dim3((unsigned)n_items), dim3(kWave), 0, st, d_src,
                       d_items, n_items, d_item_check, d_slots, d_item_size);
  else if (variant == 6)  // occupancy experiment: 5 wavefronts per CU
    hipLaunchKernelGGL((lz4_compress_l2_kernel<0, false, 16384>), dim3((unsigned)n_items), dim3(kWave), 0, st, d_src,
                       d_items, n_items, d_item_check, d_slots, d_item_size);
  else  // occupancy experiment: 7 wavefronts per CU
    hipLaunchKernelGGL((lz4_compress_l2_kernel<0, false, 6144>), dim3((unsigned)n_items), dim3(kWave), 0, st, d_src,
                       d_items, n_items, d_item_check, d_slots, d_item_size);
}

}  // namespace s3s
