// lz4_compress.hip — bit-exact LZ4 block compression of 32 KiB shuffle chunks on CDNA4.
//
// Replaces the [EXT] LZ4BlockOutputStream.flushBufferedData() stage (lz4-java 1.8.0 ->
// liblz4 1.9.3 LZ4_compress_default + xxHash32) that produces the bytes arriving at
// S3ShuffleMapOutputWriter.scala:182-188 in the reference.  Output must equal the JVM path
// byte for byte, so this is NOT a "GPU-friendly LZ4 variant": it reproduces the greedy
// single-pass parse of LZ4_compress_generic(byU16, acceleration 1) exactly, including its
// hash-table update order, skip schedule, backward catch-up and end-of-block rules.
//
// One wavefront per 32 KiB chunk; the chunk is read in place through L1/L2 and only the 8192 x u16 hash table
// lives in LDS (16 KiB -> 10 wavefronts per CU: the parse is issue / latency bound, so resident wavefronts are
// what buy throughput).  Two parses, same bytes:
//   general batch (kWindows = false, S3S_OPT_LZ4_VARIANT 1)  the sequential probe loop 64 probes at a time:
//       lane i takes the i-th position of the deterministic "no match yet" schedule, hashes it, reads the table
//       and ALL lanes insert speculatively with one ds_write.  A readback tells every lane whether it lost a
//       same-slot race; the first such lane bounds the clean prefix, the first matching lane below it wins
//       (ballot + ctz), lanes behind it undo their inserts.  Match extension is cooperative (256 B forward /
//       64 B backward per memory round trip).  tests/model/lz4_wave_model.cpp proves that which lane wins a
//       same-address LDS store is irrelevant.
//   lean exact windows (kWindows = true, variant 10, default)  in front of the general batch: aligned 64-byte
//       windows resolved with scalar mask arithmetic (see below; tests/model/lz4_window_model.cpp).
// Earlier experiments (chunk staged in LDS, pipelined windows, run loop on the vector ALU, fused frame hash,
// occupancy probes: variants 0, 2-7 of round 1) are in the git history and DESIGN.md §6, not in the product.
// The frame's xxHash32 is computed by a separate streaming kernel.  Output: token / literal / offset bytes go
// straight to the chunk's slot in HBM; compressed output never exceeds the chunk length (anything longer is
// stored RAW by the frame rule compressedLength >= originalLength), so a slot is 32 B + 32 KiB.
#include "s3s_internal.h"
#include "lz4_window_engine.inc"

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
      // lanes 1..lit carry the literals; every other lane re-reads the first one (never past the chunk: it
      // may end the caller's allocation)
      const int lp = (lane >= 1 && lane <= lit) ? anchor + lane - 1 : anchor;
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

struct TabLds {
  static constexpr bool kLds = true;
  volatile lds_u16* t;
  struct Cell {
    volatile lds_u16* p;
    __device__ __forceinline__ operator uint32_t() const { return *p; }
    __device__ __forceinline__ void operator=(uint16_t v) const { *p = v; }
  };
  __device__ __forceinline__ Cell operator[](uint32_t i) const { return Cell{t + i}; }
};

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
template <typename Src, bool kWindows, typename Tab = TabLds>
__device__ int lz4_compress_wave(const Src in, const Tab T, int len, uint8_t* out, int lane) {
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
    int kp = -2;         // first position of the window whose v is held in vp
    uint32_t vp = 0;
    int pk = -1000;      // first position of the window whose stream bytes are held in pw0
    uint4 pw0 = make_uint4(0, 0, 0, 0), pw1 = pw0, pw2 = pw0;
    bool force_general = false;
    for (;;) {
      if constexpr (kWindows) {
        // ===== lean exact windows (variant 10) =====================================================================
        // Runs are resolved with scalar mask arithmetic over K / ED (round 1's exact windows), but the
        // window is prepared with ONE memory round trip instead of three:
        //   * every lane keeps the 16 bytes p-4 .. p+11 of its position for this window and the next two
        //     (linear, issued two windows ahead: never waited for in steady state);
        //   * after the table read each lane gathers the 16 bytes cp-4 .. cp+11 around its candidate; that one
        //     load decides "candidate matches", the forward length up to 8 bytes beyond MINMATCH and the
        //     backward count up to 4 — enough for the short matches of serialized rows; anything longer goes
        //     to the cooperative extension (one more round trip per LONG sequence only);
        //   * the duplicate-hash passes over the table run while that gather is in flight.
        int wbase = base & ~63;
        if (!force_general && t0 <= 48 && wbase >= 64 && wbase <= fast_limit) {
          have_pre = false;
          // window state: set by the preparation below, or by the hand-written block when it hands over in mid-window
          uint32_t v, h, cp, info;
          uint64_t Dp, ED, K, valid;
          int rs, pend_q;
          bool resumed = false;
#ifndef S3S_NO_WINDOW_ENGINE
          if constexpr (Tab::kLds)
          if ((uint32_t)reinterpret_cast<uintptr_t>(T.t) == 0u) {  // (the block addresses the table at LDS offset 0)
            // Whole windows in one hand-written gfx950 block (lz4_window_engine.inc): it returns at a window
            // boundary (codes 0, 3, 4) or at the first event it does not handle (code 2: the loop below resumes there).
            int code, putp = __builtin_amdgcn_readfirstlane(put_pending ? 1 : 0);
            const int base_in = base;
            uint32_t vput_s = __builtin_amdgcn_readfirstlane(vput);
            uint64_t p0l = pw0.x | ((uint64_t)pw0.y << 32), p0h = pw0.z | ((uint64_t)pw0.w << 32);
            uint64_t p1l = pw1.x | ((uint64_t)pw1.y << 32), p1h = pw1.z | ((uint64_t)pw1.w << 32);
            uint64_t p2l = pw2.x | ((uint64_t)pw2.y << 32), p2h = pw2.z | ((uint64_t)pw2.w << 32);
            asm volatile(S3S_WINDOW_ENGINE_ASM
                         : [code] "=&s"(code), [h] "=&v"(h), [cp] "=&v"(cp), [info] "=&v"(info), [base] "+s"(base),
                           [t0] "+s"(t0), [anchor] "+s"(anchor), [op] "+s"(op), [pk] "+s"(pk), [kp] "+s"(kp),
                           [vput] "+s"(vput_s), [putp] "+s"(putp), [K] "=&s"(K), [valid] "=&s"(valid), [ED] "=&s"(ED),
                           [Dp] "=&s"(Dp), [rs] "=&s"(rs), [pendq] "=&s"(pend_q), [p0l] "+v"(p0l), [p0h] "+v"(p0h),
                           [p1l] "+v"(p1l), [p1h] "+v"(p1h), [p2l] "+v"(p2l), [p2h] "+v"(p2h), [vp] "+v"(vp)
                         : [len] "s"(len), [inp] "s"(in.base), [outp] "s"(out)
                         : "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86",
                           "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99",
                           "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133",
                           "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145",
                           "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v157", "v158",
#if !defined(S3S_ENGINE_NO_SPEC)  // speculative next-window gather: two more gather register sets, cpS, the next window's hash
                           "v155", "v156", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167",
#if !defined(S3S_ENGINE_NO_DEFER)  // deferred emission: the waiting sequences' records (v118, v119), the flush's scratch, the record count
                           "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110",
                           "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "m0",
#endif
#elif defined(S3S_ABL_DUP_PW) || defined(S3S_ABL_DUP_GATHER) || defined(S3S_ABL_DUP_IPSIDE)
                           "v160", "v161", "v162", "v163",
#endif
                           "vcc", "scc", "memory");
            pw0 = make_uint4((uint32_t)p0l, (uint32_t)(p0l >> 32), (uint32_t)p0h, (uint32_t)(p0h >> 32));
            pw1 = make_uint4((uint32_t)p1l, (uint32_t)(p1l >> 32), (uint32_t)p1h, (uint32_t)(p1h >> 32));
            pw2 = make_uint4((uint32_t)p2l, (uint32_t)(p2l >> 32), (uint32_t)p2h, (uint32_t)(p2h >> 32));
            vput = vput_s;
            put_pending = putp != 0;
            if (code == 3) break;  // last literals
            if (code == 4) {
              force_general = true;
              continue;
            }
            if (code == 0 && base != base_in) continue;  // the conditions fail at the next window: re-dispatch
            resumed = code == 2;
            wbase = base & ~63;
          }
#endif
          const int p = wbase + lane;
          const int rs0 = base - wbase;
          if (!resumed) {
          if (put_pending) T[hash13(vput)] = (uint16_t)(base - 2);  // LZ4_putPosition(ip - 2)
          put_pending = false;
          DBG_T(t4a);
          DBG_ADD(8, 1);
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
          v = pw0.y;
          h = hash13(v);
          const bool live = lane >= rs0;
          cp = T[h];
#ifdef S3S_LZ4_TIMING
          asm volatile("" ::"v"(cp));
#endif
          DBG_T(t4b);
          const bool near0 = cp < 4u;  // (also every empty slot: it reads as position 0)
          const uint4 G = in.ld16(near0 ? 0 : (int)cp - 4);
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
          const bool em = live && (w == v);
          // per-lane event record: [15:0] table candidate, [19:16] forward length 0..8 (8 = at least),
          // [22:20] backward equal bytes 0..4 (4 = at least), [30] candidate matches
          info = 0u;
          if (em) {
            uint32_t fl = 8u, be = 0u;  // near0: lengths by the cooperative extension
            if (!near0) {
              const uint32_t xz = G.z ^ pw0.z, xw = G.w ^ pw0.w, xb = G.x ^ pw0.x;
              fl = xz ? (uint32_t)(__builtin_ctz(xz) >> 3) : (xw ? 4u + (uint32_t)(__builtin_ctz(xw) >> 3) : 8u);
              be = xb ? (uint32_t)(__builtin_clz(xb) >> 3) : 4u;
            }
            info = cp | (fl << 16) | (be << 20) | 0x40000000u;
          }
          const uint64_t Ecp = __ballot(em);
          Dp = __ballot(grp);
          ED = Ecp | Dp;  // lanes the run loop has to look at
          DBG_T(t4d0);
          DBG_ADD(0, t4b - t4a);
          DBG_ADD(1, t4c0 - t4b);
          DBG_ADD(2, t4c - t4c0);
          DBG_ADD(6, t4d0 - t4c);
          K = 0;  // lanes the sequential code inserts: probes and ip-2 positions
          rs = rs0;
          pend_q = -1;
          }
          v = pw0.y;
          const bool vp_ok = kp == wbase - 64;  // vp holds the previous window's dwords (literals may start there)
          DBG_T(t4d);
          // ================= runs (scalar work) ========================================================
          // One loop with a single hot path: next event -> (rare: suspect lane) -> (cooperative extension)
          // -> short-form emit -> advance.  Conditions are folded into sign tests of small integer
          // expressions so that hipcc does not materialise 64-bit lane masks for every boolean.
          const bool had_match = resumed && K != 0ull;  // (a match sets its lane in K)
          int rt = had_match ? 0 : t0;
          int exit_kind = 0;   // 0: next window / batch, 1: last literals, 2: the general batch takes over
          const int e0 = rs0 + 66 - t0;  // the first run may continue an older one (t0 <= 48 => e0 >= rs0 + 18)
          int elim = (had_match || e0 >= kWave) ? kWave : e0;
          if (!resumed) valid = e0 < kWave ? ((1ull << e0) - 1ull) : ~0ull;  // consecutive probes of the current run
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
                {
                const uint32_t rel = (uint32_t)(lane - anchor) & 63u;
                const uint32_t tok = (uint32_t)(lit << 4) | (uint32_t)(mcode < 15 ? mcode : 15);
                const uint32_t d = rel - (uint32_t)lit;  // 0: token, 1/2: offset, 3: match-length byte
                uint32_t bv = (((int)(anchor + rel) < wbase) ? vp : v) & 0xffu;
                bv = d == 0u ? tok : bv;
                bv = d == 1u ? (uint32_t)offset : bv;
                bv = d == 2u ? (uint32_t)offset >> 8 : bv;
                bv = d == 3u ? (uint32_t)(mcode - 15) : bv;
                const uint32_t idx = d == 0u ? 0u : rel + (rel < (uint32_t)lit ? 1u : 0u);
                if (rel < (uint32_t)total) out[(uint32_t)op + idx] = (uint8_t)bv;
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

constexpr uint32_t XXP1 = 2654435761u, XXP2 = 2246822519u, XXP3 = 3266489917u,
                   XXP4 = 668265263u, XXP5 = 374761393u;

// xxHash32 of g[0,len) by one wavefront: the chunk is streamed 256 bytes per coalesced load, the four
// stripe accumulators live in lanes 0..3 and pull their words out of the block by cross-lane reads.
// (Fused into the compress kernel it also brings the chunk into L2 right before the parse.)
__device__ __forceinline__ uint32_t xxh32_wave(const uint8_t* g, int len, uint32_t seed, int lane) {
  uint32_t acc = lane == 0 ? seed + XXP1 + XXP2 : lane == 1 ? seed + XXP2 : lane == 2 ? seed : seed - XXP1;
  const int stripes = len >> 4, nblk = len >> 8;
  auto ld32u = [&](int byte_pos) -> uint32_t {
    // non-temporal: the compress wavefront fetches its block itself right before the parse (see the kernel), what this
    // pass reads must not push the blocks that are being parsed out of L2
    typedef uint32_t u32_unaligned __attribute__((aligned(1)));
    return __builtin_nontemporal_load(reinterpret_cast<const u32_unaligned*>(g + byte_pos));
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
template <bool kWindows>
__global__ __launch_bounds__(kWave) void lz4_compress_l2_kernel(
    const uint8_t* __restrict__ src_, const Item* __restrict__ items_, int32_t n_items_, int32_t slot_stride_,
    const uint32_t* __restrict__ item_check_, uint8_t* __restrict__ slots_,
    uint32_t* __restrict__ item_size_, uint32_t* __restrict__ work_) {  // (read through the kernarg segment, see the loop)
#ifdef S3S_ABL_LDS_PAD  // occupancy experiment: fewer wavefronts per CU (timing only)
  __shared__ __attribute__((aligned(16))) uint16_t table[8192 + S3S_ABL_LDS_PAD / 2];
#else
  __shared__ __attribute__((aligned(16))) uint16_t table[8192];
#endif
  const int lane = threadIdx.x;
  // A persistent grid (10 wavefronts per CU, what the 16 KiB tables allow): every wavefront takes the next block from
  // a counter until none is left.  Against one workgroup per block: no workgroup launches between blocks, the blocks in
  // flight stay neighbours in memory, and a launch that is alone on the chip packs its last round
  // (profiles/r03_experiments.md §8).
  // The arguments are read again from the kernarg segment in every round (the empty asm hides from the compiler that
  // the pointer is the same): kept in registers across the parse they cost 15 SGPRs next to the ones the window block
  // pins, and the compiler spills more scalars into VGPR lanes.
  struct KArgs {
    const uint8_t* src;
    const Item* items;
    int32_t n_items, slot_stride;
    const uint32_t* item_check;
    uint8_t* slots;
    uint32_t* item_size;
    uint32_t* work;
  };
  typedef const KArgs __attribute__((address_space(4))) * KArgsPtr;
  for (;;) {
  KArgsPtr ka = (KArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(ka));
  const uint8_t* const src = ka->src;
  const Item* const items = ka->items;
  const int n_items = ka->n_items;
  const uint32_t* const item_check = ka->item_check;
  uint8_t* const slots = ka->slots;
  uint32_t* const item_size = ka->item_size;
  uint32_t it0 = 0;
  if (lane == 0) it0 = atomicAdd(ka->work, 1u);
  const int it = (int)__builtin_amdgcn_readfirstlane(it0);
  if (it >= n_items) break;
  Item item = items[it];
  {  // (a vector load through a pointer the compiler knows nothing about: the fields are uniform, say so)
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)item.src_off);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)item.src_off >> 32));
    item.src_off = (int64_t)(((uint64_t)hi << 32) | lo);
    item.len = (int32_t)__builtin_amdgcn_readfirstlane((uint32_t)item.len);
    item.kind = (int32_t)__builtin_amdgcn_readfirstlane((uint32_t)item.kind);
    item.chunk = (int32_t)__builtin_amdgcn_readfirstlane((uint32_t)item.chunk);
  }
  const int kind = item.kind & 0xff;
  if (kind != kItemLz4Chunk) {
    if (lane == 0 && kind == kItemLz4End) item_size[it] = kLz4FrameHeader;
    continue;
  }
  __syncthreads();
  {
    uint4* tz = reinterpret_cast<uint4*>(table);
    for (int i = lane; i < 16384 / 16; i += kWave) tz[i] = make_uint4(0, 0, 0, 0);
  }
  __syncthreads();
  // Touch the whole block first: 32 independent 1 KiB rows per wavefront, one wait.  The parse that follows reads the
  // block through ONE wavefront's dependent gathers; with the block brought into this XCD's L2 right before it they hit
  // there (measured with the persistent grid: headline 89.6 - 90.2 -> 93.4 - 95.8 GB/s, a launch alone 3.06 -> 2.84 ms;
  // with one workgroup per block it had been + 1.8 %).  The xxHash32 pre-pass no longer has to warm anything and reads
  // non-temporally (another + 1 %, where it cost 7 % before).
  {
    const uint8_t* g = src + item.src_off;
    uint32_t acc = 0;
    const int touch = item.len;
    for (int i = lane * 16; i + 16 <= touch; i += kWave * 16) {
      uint4 x;
      __builtin_memcpy(&x, g + i, 16);
      acc ^= x.x ^ x.y ^ x.z ^ x.w;
    }
    if (acc == 0x12345678u && item.len < 0) table[0] = 1;  // (never taken: keeps the loads)
  }
  uint8_t* slot = slots + (size_t)item.chunk * (size_t)(uint32_t)ka->slot_stride;
  const uint32_t check = item_check[it];
  const int clen = lz4_compress_wave<SrcGlobal, kWindows>(SrcGlobal{src + item.src_off}, TabLds{(lds_u16*)table},
                                                      item.len, slot + kSlotHeader, lane);
  finish_frame(slot, item.len, clen, check, item.kind >> 8, item_size + it, lane);
  }
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


// Round 5: four lanes per chunk, sixteen chunks per wavefront.  xxHash32 is four serial chains per chunk (the stripe
// accumulators); one wavefront per chunk kept 4 of 64 lanes busy and spent the rest on cross-lane reads (1.36 TB/s on
// a 1 GiB block, 0.34 ms in front of every map-side call).  Here lane 4 b + k is accumulator k of the wavefront's chunk b
// and reads its own dword of every stripe: a quad reads 16 contiguous bytes per load, eight stripes ahead.
template <bool kNonTemporal>
__global__ __launch_bounds__(kWave) void xxh32_items_quad_kernel(
    const uint8_t* __restrict__ src, const Item* __restrict__ items, int32_t n_items, uint32_t seed,
    uint32_t* __restrict__ item_check) {
  const int lane = threadIdx.x, k = lane & 3;
  const int it = (int)blockIdx.x * 16 + (lane >> 2);
  Item item;
  item.src_off = 0;
  item.len = 0;
  item.kind = 0;
  item.chunk = 0;
  if (it < n_items) item = items[it];
  const bool mine = it < n_items && (item.kind & 0xff) == kItemLz4Chunk;
  const int len = mine ? item.len : 0;
  const uint8_t* g = src + item.src_off;
  typedef uint32_t u32_unaligned __attribute__((aligned(1)));
  auto ld = [&](int byte_pos) -> uint32_t {
    const u32_unaligned* q = reinterpret_cast<const u32_unaligned*>(g + byte_pos);
    return kNonTemporal ? __builtin_nontemporal_load(q) : *q;
  };
  uint32_t acc = k == 0 ? seed + XXP1 + XXP2 : k == 1 ? seed + XXP2 : k == 2 ? seed : seed - XXP1;
  const int stripes = len >> 4;
  int i = 0;
  // The chain of a chunk is 2 048 dependent rounds (~14 cycles each) and a map task has only 4 096 chunks = 256 wavefronts
  // of this kernel: nothing else hides the memory latency, so every lane keeps kDepth stripes (kDepth dwords) in flight — a
  // ring of registers, refilled sixteen at a time while the sixteen before them are consumed (first version: 8 in flight,
  // 1.08 ms per two map tasks; the round-2 kernel with one wavefront per chunk: 0.34 ms).
  // Chunks of a wavefront differ in length (every partition ends with a short one, and the plan has items that are not
  // chunks): NO branch may depend on that, or the compiler can no longer count the loads in flight and waits for all of
  // them at every use (measured: 409 cycles per round).  So every lane streams — a lane without 16 stripes of its own
  // reads the chunk of the wavefront's first lane that has them and discards the result — and a lane whose chunk ends
  // early clamps its stripe index and stops accumulating (one select per 16 rounds).
  const int s16 = stripes & ~15;
  const unsigned long long good = __ballot(s16 > 0);
  int wmax = s16;
#pragma unroll
  for (int d = 32; d >= 4; d >>= 1) {
    const int o = __shfl_xor(wmax, d);
    wmax = o > wmax ? o : wmax;
  }
  wmax = __builtin_amdgcn_readfirstlane(wmax);
  constexpr int kDepth = 48;  // (vmcnt counts to 63: more loads in flight cannot be waited for selectively)
  if (good != 0ull) {
    const int donor = (int)__builtin_ctzll(good);
    // (offsets, not pointers, cross the lanes: a pointer rebuilt from integers loses its address space and the loads become
    // flat_load, whose second counter forces a wait for ALL of them)
    const uint64_t od = (uint64_t)(uint32_t)__shfl((int)(uint32_t)(uint64_t)item.src_off, donor) |
                        ((uint64_t)(uint32_t)__shfl((int)(uint32_t)((uint64_t)item.src_off >> 32), donor) << 32);
    const int sd = __shfl(s16, donor);
    const uint8_t* gr = src + (s16 > 0 ? item.src_off : (int64_t)od);  // what this lane streams
    const int last = (s16 > 0 ? s16 : sd) - 1;                                            // its last whole stripe
    auto ldr = [&](int t) -> uint32_t {
      const u32_unaligned* q = reinterpret_cast<const u32_unaligned*>(gr + 16 * (t < last ? t : last) + 4 * k);
      return kNonTemporal ? __builtin_nontemporal_load(q) : *q;
    };
    uint32_t w[kDepth];
#pragma unroll
    for (int j = 0; j < kDepth; j++) w[j] = ldr(j);
    // invariant: the ring holds stripes i .. i + kDepth - 1 (clamped)
    for (; i + kDepth <= wmax; i += kDepth) {  // one turn: every group is consumed, then loaded kDepth stripes ahead
#pragma unroll
      for (int gq = 0; gq < kDepth / 16; gq++) {
        uint32_t a = acc;
#pragma unroll
        for (int j = 0; j < 16; j++) a = rotl32(a + w[16 * gq + j] * XXP2, 13) * XXP1;
        acc = i + 16 * gq < s16 ? a : acc;
#pragma unroll
        for (int j = 0; j < 16; j++) w[16 * gq + j] = ldr(i + kDepth + 16 * gq + j);
      }
    }
#pragma unroll
    for (int gq = 0; gq < kDepth / 16; gq++) {
      uint32_t a = acc;
#pragma unroll
      for (int j = 0; j < 16; j++) a = rotl32(a + w[16 * gq + j] * XXP2, 13) * XXP1;
      acc = i + 16 * gq < s16 ? a : acc;
    }
    i = s16;
  }
  for (; i < stripes; i++) acc = rotl32(acc + ld(16 * i + 4 * k) * XXP2, 13) * XXP1;
  const int q0 = lane & ~3;
  const uint32_t v1 = (uint32_t)__shfl((int)acc, q0), v2 = (uint32_t)__shfl((int)acc, q0 + 1),
                 v3 = (uint32_t)__shfl((int)acc, q0 + 2), v4 = (uint32_t)__shfl((int)acc, q0 + 3);
  uint32_t h = len >= 16 ? rotl32(v1, 1) + rotl32(v2, 7) + rotl32(v3, 12) + rotl32(v4, 18) : seed + XXP5;
  h += (uint32_t)len;
  if (mine && k == 0) {
    int p = stripes << 4;
    for (; p + 4 <= len; p += 4) h = rotl32(h + ld(p) * XXP3, 17) * XXP4;
    for (; p < len; p++) h = rotl32(h + (uint32_t)g[p] * XXP5, 11) * XXP1;
    h ^= h >> 15;
    h *= XXP2;
    h ^= h >> 13;
    h *= XXP3;
    h ^= h >> 16;
    item_check[it] = h;
  }
}

}  // namespace

void launch_lz4_compress(const uint8_t* d_src, const Item* d_items, int32_t n_items,
                         uint32_t* d_item_check, uint8_t* d_slots, int32_t slot_stride, uint32_t* d_item_size, uint32_t* d_work, int resident_waves,
                         int variant, hipStream_t st, hipEvent_t after_hash) {
  if (n_items <= 0) {
    if (after_hash) hipEventRecord(after_hash, st);
    return;
  }
  // Default (round 5): four lanes per chunk, sixteen chunks per wavefront, 48 loads in flight per lane, ORDINARY loads —
  // measured against the per-chunk wavefronts of round 2 and against non-temporal loads (profiles/r05_experiments.md §4):
  // hash stage of a two-task call 0.20 ms / 0.38 / 0.35 - 0.42, headline 99.2 - 99.4 / 98.1 - 98.3 / 96.6 - 98.1 GB/s, a 1 GiB
  // block 0.22 ms (4.8 TB/s) / 0.75 / 0.53.  (Round 3 had made the pass non-temporal so that it would not displace the blocks
  // being parsed; with the block touch inside the persistent wavefront that no longer shows, and the pass leaves the chunks
  // in the Infinity Cache for that touch.)  S3S_XXH forces a variant: 0 = per-chunk wavefronts, 1 = quads, non-temporal.
  static const int xxh_env = getenv("S3S_XXH") ? atoi(getenv("S3S_XXH")) : -1;
  const int xxh_mode = xxh_env >= 0 ? xxh_env : 2;
  if (xxh_mode == 0)
    hipLaunchKernelGGL(xxh32_items_wave_kernel, dim3((unsigned)n_items), dim3(kWave), 0, st, d_src, d_items,
                       n_items, kLz4BlockSeed, d_item_check);
  else if (xxh_mode == 2)
    hipLaunchKernelGGL(xxh32_items_quad_kernel<false>, dim3((unsigned)((n_items + 15) / 16)), dim3(kWave), 0, st, d_src, d_items,
                       n_items, kLz4BlockSeed, d_item_check);
  else
    hipLaunchKernelGGL(xxh32_items_quad_kernel<true>, dim3((unsigned)((n_items + 15) / 16)), dim3(kWave), 0, st, d_src, d_items,
                       n_items, kLz4BlockSeed, d_item_check);
  if (after_hash) hipEventRecord(after_hash, st);
  // d_work: the launch's block counter (zeroed in stream order); resident_waves: 10 per CU
  (void)hipMemsetAsync(d_work, 0, sizeof(uint32_t), st);
  static const int grid_env = getenv("S3S_LZ4_GRID") ? atoi(getenv("S3S_LZ4_GRID")) : 0;  // (experiments)
  int grid = grid_env > 0 ? grid_env : resident_waves;
  if (grid > n_items) grid = n_items;
  if (variant == 1)
    hipLaunchKernelGGL(lz4_compress_l2_kernel<false>, dim3((unsigned)grid), dim3(kWave), 0, st, d_src,
                       d_items, n_items, slot_stride, d_item_check, d_slots, d_item_size, d_work);
  else
    hipLaunchKernelGGL(lz4_compress_l2_kernel<true>, dim3((unsigned)grid), dim3(kWave), 0, st, d_src,
                       d_items, n_items, slot_stride, d_item_check, d_slots, d_item_size, d_work);
}

}  // namespace s3s
