// lz4_decode_batch.hip — LZ4 block decoder that works on BATCHES of sequences (decode variant 4).
//
// Replaces the per-block LZ4_decompress call inside [EXT] LZ4BlockInputStream.refill()
// (S3ShuffleReader.scala:108 wraps the range stream with it).  One wavefront per LZ4Block frame.
//
// The ring decoder (lz4_decompress.hip, variant 3) spends ~100+ vector instructions per LZ4 sequence
// and a CU issues one vector (and one scalar) instruction per cycle for all of its wavefronts, so it is
// instruction-issue bound at ~1 sequence per 2000 CU cycles.  A decoder is free to reorganise the work (no bit-exact parse to follow), so
// this one makes the lanes work on DIFFERENT sequences:
//
//   parse    lane i assumes a token starts at stream byte ip+i and decodes it speculatively (literal
//            length, offset, match length, position of the next token) from two dword loads; a short
//            scalar walk over `next` marks the real tokens of the 64-byte window (5 instructions per
//            sequence, hand-written); they are appended to the batch (one record per lane, via LDS)
//   batch    (up to 64 sequences) a prefix sum gives every sequence its output position; ALL literal
//            runs of the batch are copied at once, one lane per sequence; matches are copied as 16-byte
//            PIECES, one lane per piece, in dependency rounds — after every match has been pointed at
//            where it REALLY reads from: a source inside one earlier match's output is that match's
//            source shifted (chains resolved by pointer doubling), a source inside the batch's literals
//            is final from the start, periods 1 / 2 / 4 are splats; a match above 64 bytes, an odd
//            period or a source that straddles the window base is copied by the whole wave
//   output   is staged in a sliding LDS window (4.4 KiB, 2 KiB of history survive a slide): match
//            sources are LDS reads, the block leaves in 16-byte coalesced stores; a source older than
//            the window is read back from L2 (after the flush that wrote it has drained)
//   tokens the fast parse does not take (lengths with a 255 chain, the block's last sequence) go
//            through a byte-wise scalar path that performs every LZ4_decompress_safe bounds check
//
// tests/model/lz4_batch_decode_model.cpp is the lock-step CPU model of this kernel (fuzzed against
// liblz4 incl. malformed blocks); malformed input ends in S3S_E_BAD_FRAME, never out of bounds.
#include "s3s_internal.h"

#ifdef S3S_LZ4_TIMING
__device__ unsigned long long g_bdec_dbg[16];  // (instrumented build: phase ticks of batch_decode_kernel)
#endif

namespace s3s {
namespace {

#ifndef S3S_BWIN
#define S3S_BWIN 4544
#endif
#ifndef S3S_BHIST
#define S3S_BHIST 2048
#endif
constexpr int kBWin = S3S_BWIN;    // staged output bytes (multiple of 16); with pad + records = 5 KiB -> 32 wavefronts / CU
                                   // (round 2, compiled kernel: 7616 / 4096 = 20 per CU 278 GB/s, 5568 / 4096 298, 4544 / 3072 284, 3520 / 2048 292;
                                   //  round 6, after the instruction diet (profiles/r06p_*, TeraSort / wide rows LZ4 / Snappy / 1 GiB blocks):
                                   //  7616 / 4096 480 / 223 / 211 / 466, 5568 / 4096 492 - 503 / 247 / 226 / 478, 4544 / 3072 494 / 257 / 226 / 497,
                                   //  4544 / 2048 501 / 261 / 226 / 505, 3520 / 2048 493 / 257 / 224 / 489 — with fewer instructions per frame the
                                   //  kernel follows its residency again)
constexpr int kBHist = S3S_BHIST;  // history a slide keeps
static_assert(kBWin % 16 == 0 && kBHist % 16 == 0 && kBWin >= kBHist + 1024, "window geometry");
constexpr int kBPad = 64;
#ifdef S3S_DEC_RING
// (experiment, -DS3S_DEC_RING) The parse windows read the compressed stream through a ring in LDS (two halves of 256 bytes, one dword per lane per
// refill, the next half prefetched into a register while the current one is parsed): a window's two dependent reads
// cost two LDS round trips instead of two trips to L2.  + 4: the ring's first dword again, for reads across the end.
constexpr int kSRing = 512, kSHalf = 256;
#endif
constexpr int kSmallLit = 16; // literal runs up to this length are copied one lane per sequence

constexpr uint32_t XP1 = 2654435761u, XP2 = 2246822519u, XP3 = 3266489917u, XP4 = 668265263u, XP5 = 374761393u;

__device__ __forceinline__ uint32_t g_ld32(const uint8_t* p) {
  uint32_t v;
  __builtin_memcpy(&v, p, 4);
  return v;
}
__device__ __forceinline__ uint32_t l2_ld8(const uint8_t* p) {
  return (uint32_t)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// four bytes at an arbitrarily aligned address, served by L2 (the bytes were written by this wave's own flush):
// gfx950 takes unaligned dword addresses, and the cache policy bit is all the atomic is for
__device__ __forceinline__ uint32_t l2_ld32(const uint8_t* p) {
  return __hip_atomic_load(reinterpret_cast<const uint32_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Unaligned LDS dwords: gfx950 serves a ds_read / ds_write whose address is not a multiple of its width one lane at a
// time (tools/probe/lds_align_probe.hip: 65 LDS cycles per wavefront instruction against 3.4 - 5 aligned).  A version
// of the copies below that reads and writes aligned dwords only (v_alignbyte_b32 for the shifts, byte stores at the
// ends of a span) removed every such stall and halved the LDS pipe's busy time — and was 7 % SLOWER: the CU's scalar
// and vector issue (one of each per cycle for all its waves) is what this kernel runs out of, and the aligned form
// costs 10 % more instructions (profiles/r03_experiments.md §7).
__device__ __forceinline__ uint32_t lds_ld32(const uint8_t* p) {
  uint32_t v;
  __builtin_memcpy(&v, p, 4);  // gfx950: unaligned ds_read_b32
  return v;
}
__device__ __forceinline__ void lds_st32(uint8_t* p, uint32_t v) { __builtin_memcpy(p, &v, 4); }
__device__ __forceinline__ void lds_st16(uint8_t* p, uint32_t v) {
  const uint16_t h = (uint16_t)v;
  __builtin_memcpy(p, &h, 2);
}
// 16 bytes at any address with ONE instruction each way (the compiler splits what it cannot prove aligned; the
// hardware takes any address, at the unaligned cost of one dword access).  The read waits for its data; the stores
// are ordinary in-order LDS operations (the compiler's lgkmcnt bookkeeping only gets more conservative by them).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)reinterpret_cast<uintptr_t>(p); }
__device__ __forceinline__ u32x4 lds_ld128u(const uint8_t* p) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(lds_addr(p)) : "memory");
  return v;
}
__device__ __forceinline__ void lds_st128u(uint8_t* p, u32x4 v) {
  asm volatile("ds_write_b128 %0, %1" : : "v"(lds_addr(p)), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_st64u(uint8_t* p, uint32_t lo, uint32_t hi) {
  const uint64_t v = (uint64_t)lo | ((uint64_t)hi << 32);
  asm volatile("ds_write_b64 %0, %1" : : "v"(lds_addr(p)), "v"(v) : "memory");
}
// the first n (1..16) bytes of x to d.  Per-lane lengths without a branch (round 6): the compiler's form of the obvious
// cascade (b64 if n & 8, b32 if n & 4, ...) is four s_and_saveexec / s_or exec pairs with their skips — 49 instructions,
// two thirds of them scalar, per round of the match copies, in a kernel that is bound by the CU's issue ports.  Here
// the four dwords go out under v_cmpx (exec = "my run has this dword"), then the 1 - 3 tail bytes as a halfword and a
// byte: 30 instructions, 5 of them scalar, no branch.  n = 0 stores nothing.
__device__ __forceinline__ void lds_store16(uint8_t* d, u32x4 x, int n) {
  const uint32_t a = lds_addr(d);
  uint64_t sv, q;
  uint32_t t, t2, ta, b;
  // wait states inside the block (tests/isa/hazards.py checks them): a v_cmp's SGPR / VCC result is read by a v_cndmask
  // no sooner than the third instruction behind it
  asm volatile(
      "s_mov_b64 %[sv], exec\n\t"
      "v_cmpx_le_u32_e32 vcc, 4, %[n]\n\t"
      "ds_write_b32 %[a], %[x0]\n\t"
      "v_cmpx_le_u32_e32 vcc, 8, %[n]\n\t"
      "ds_write_b32 %[a], %[x1] offset:4\n\t"
      "v_cmpx_le_u32_e32 vcc, 12, %[n]\n\t"
      "ds_write_b32 %[a], %[x2] offset:8\n\t"
      "v_cmpx_le_u32_e32 vcc, 16, %[n]\n\t"
      "ds_write_b32 %[a], %[x3] offset:12\n\t"
      "s_mov_b64 exec, %[sv]\n\t"
      // the dword that holds the tail: x[n >> 2] (two-level select on bits 2 and 3 of n)
      "v_and_b32_e32 %[b], 4, %[n]\n\t"
      "v_and_b32_e32 %[t], 8, %[n]\n\t"
      "v_cmp_ne_u32_e64 %[q], 0, %[b]\n\t"
      "v_cmp_ne_u32_e32 vcc, 0, %[t]\n\t"
      "v_and_b32_e32 %[ta], 12, %[n]\n\t"
      "v_and_b32_e32 %[b], 2, %[n]\n\t"
      "v_cndmask_b32_e64 %[t], %[x0], %[x1], %[q]\n\t"
      "v_add_u32_e32 %[ta], %[a], %[ta]\n\t"
      "v_cndmask_b32_e64 %[t2], %[x2], %[x3], %[q]\n\t"
      "v_cndmask_b32_e32 %[t], %[t], %[t2], vcc\n\t"
      "v_cmpx_ne_u32_e32 vcc, 0, %[b]\n\t"
      "ds_write_b16 %[ta], %[t]\n\t"
      "s_mov_b64 exec, %[sv]\n\t"
      "v_add_u32_e32 %[ta], %[ta], %[b]\n\t"
      "v_lshlrev_b32_e32 %[b], 3, %[b]\n\t"
      "v_lshrrev_b32_e32 %[t], %[b], %[t]\n\t"
      "v_and_b32_e32 %[b], 1, %[n]\n\t"
      "v_cmpx_ne_u32_e32 vcc, 0, %[b]\n\t"
      "ds_write_b8 %[ta], %[t]\n\t"
      "s_mov_b64 exec, %[sv]"
      : [sv] "=&s"(sv), [q] "=&s"(q), [t] "=&v"(t), [t2] "=&v"(t2), [ta] "=&v"(ta), [b] "=&v"(b)
      : [a] "v"(a), [n] "v"(n), [x0] "v"(x.x), [x1] "v"(x.y), [x2] "v"(x.z), [x3] "v"(x.w)
      : "vcc", "memory");
}

// (HIP's __ballot takes an int: the compiler first materialises the condition as 0 / 1 and compares it again)
__device__ __forceinline__ uint64_t ballot64(bool b) { return __builtin_amdgcn_ballot_w64(b); }

// inclusive prefix sum over the 64 lanes in the DPP network (no LDS round trips): Hillis-Steele inside each row of
// 16 lanes, then lane 15 of rows 0 / 2 into rows 1 / 3, then lane 31 into rows 2 and 3
__device__ __forceinline__ int wave_scan_add(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
  return v;
}

enum { kFmtLz4 = 0, kFmtSnappy = 1, kFmtLzf = 2 };  // kFmtLzf (round 4): liblzf blocks of compress-lzf chunks, elements like Snappy's

#ifdef S3S_LZ4_TIMING
// phase accounting of the batch decoder (instrumented build only; tools/dec_timing.py): s_memtime ticks per phase
#define BT_DECL unsigned long long bt_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long bt_t0_ = __builtin_readcyclecounter(), bt_start_ = bt_t0_
#define BT_MARK(k) do { const unsigned long long t_ = __builtin_readcyclecounter(); bt_[k] += t_ - bt_t0_; bt_t0_ = t_; } while (0)
#define BT_COUNT(k) bt_[k]++
#define BT_DONE() do { if (lane == 0) { bt_[8] = __builtin_readcyclecounter() - bt_start_; for (int k_ = 0; k_ < 12; k_++) atomicAdd(&g_bdec_dbg[k_], bt_[k_]); atomicAdd(&g_bdec_dbg[12], 1ull); } } while (0)
#else
#define BT_DECL
#define BT_MARK(k)
#define BT_COUNT(k)
#define BT_DONE()
#endif
// The tokens of a parse window are marked by a scalar walk over the chain (five instructions per token, see the loop)
// or, when the previous window held at least this many tokens, by pointer doubling through LDS (≈ 110 instructions
// whatever the count).  Round 2 measured the compiler's 13-instruction walk: always scalar 295 / 148 GB/s on TeraSort /
// wide rows, always doubling 291 / 164, hence a threshold of 12 for LZ4 and doubling always for Snappy.  With the
// 5-instruction walk the break-even is above what a window can hold for LZ4 (21 tokens) and ≈ 24 for Snappy.
#ifndef S3S_PWALK_TOKENS
#define S3S_PWALK_TOKENS 24
#endif
#ifndef S3S_SWALK_TOKENS
#define S3S_SWALK_TOKENS 24
#endif
constexpr int kParallelWalkTokens = S3S_PWALK_TOKENS;
constexpr int kSnappyWalkTokens = S3S_SWALK_TOKENS;

// ---- element front ends of the Snappy / LZF parse block (operands of that asm statement) ------------------------------------
// in: d0 = the dword at the lane's stream position cpos.  out: isl (lanes whose element is a literal run), cx (byte-wise path),
// r0 / r1 (the record halves of the generic front end), nxt (stream position of the element behind), islv (isl as 0 / 1).
// Snappy: literal with its length in the tag (n6 + 1) or in one / two bytes behind it (n6 = 60 / 61; longer forms, a literal
// past the block or above 32 KiB: byte-wise), copy with a 1-byte offset (length 4 + (n6 & 7), offset (tag >> 5) << 8 | next byte)
// or a 2-byte offset (length n6 + 1); a 4-byte offset: byte-wise.
// LZF: ctrl < 32: ctrl + 1 literals; else a back reference of (ctrl >> 5 [7: + next byte]) + 2 bytes, ((ctrl & 31) << 8 | last
// byte) + 1 back.
#define S3S_EL_FRONT_SNAPPY \
  "v_and_b32_e32 %[ty], 3, %[d0]\n\t" \
  "v_bfe_u32 %[n6], %[d0], 2, 6\n\t" \
  "v_bfe_u32 %[b1], %[d0], 8, 8\n\t" \
  "v_bfe_u32 %[w16], %[d0], 8, 16\n\t" \
  "v_cmp_eq_u32_e32 vcc, 60, %[n6]\n\t" \
  "v_add_u32_e32 %[len], 1, %[n6]\n\t" \
  "v_add_u32_e32 %[vt], 1, %[b1]\n\t" \
  "v_cmp_eq_u32_e64 %[t1], 61, %[n6]\n\t" \
  "v_cndmask_b32_e32 %[len], %[len], %[vt], vcc\n\t" \
  "v_cndmask_b32_e64 %[hdr], 1, 2, vcc\n\t" \
  "v_add_u32_e32 %[vt], 1, %[w16]\n\t" \
  "v_cmp_lt_u32_e64 %[cx], 61, %[n6]\n\t" \
  "v_cndmask_b32_e64 %[len], %[len], %[vt], %[t1]\n\t" \
  "v_cndmask_b32_e64 %[hdr], %[hdr], 3, %[t1]\n\t" \
  "v_cmp_eq_u32_e64 %[isl], 0, %[ty]\n\t" \
  "v_add3_u32 %[nxt], %[cpos], %[hdr], %[len]\n\t" \
  "v_cmp_lt_u32_e32 vcc, 0x8000, %[len]\n\t" \
  "v_add_u32_e32 %[vt2], %[lane], %[hdr]\n\t" \
  "s_or_b64 %[cx], %[cx], vcc\n\t" \
  "v_cmp_lt_i32_e32 vcc, %[clen], %[nxt]\n\t" \
  "v_lshlrev_b32_e32 %[vt2], 16, %[vt2]\n\t" \
  "s_or_b64 %[cx], %[cx], vcc\n\t" \
  "s_and_b64 %[cx], %[cx], %[isl]\n\t" \
  "v_cmp_eq_u32_e32 vcc, 3, %[ty]\n\t" \
  "v_and_b32_e32 %[vt], 7, %[n6]\n\t" \
  "s_or_b64 %[cx], %[cx], vcc\n\t" \
  "v_cmp_eq_u32_e32 vcc, 1, %[ty]\n\t" \
  "v_add_u32_e32 %[vt], 4, %[vt]\n\t" \
  "v_add_u32_e32 %[m], 1, %[n6]\n\t" \
  "v_bfe_u32 %[jmp], %[d0], 5, 3\n\t" \
  "v_cndmask_b32_e32 %[vt], %[m], %[vt], vcc\n\t" \
  "v_lshl_or_b32 %[jmp], %[jmp], 8, %[b1]\n\t" \
  "v_lshlrev_b32_e32 %[vt], 16, %[vt]\n\t" \
  "v_cndmask_b32_e32 %[jmp], %[w16], %[jmp], vcc\n\t" \
  "v_cndmask_b32_e64 %[m], 3, 2, vcc\n\t" \
  "v_cndmask_b32_e64 %[r0], %[vt], %[len], %[isl]\n\t" \
  "v_add_u32_e32 %[m], %[cpos], %[m]\n\t" \
  "v_cndmask_b32_e64 %[r1], %[jmp], %[vt2], %[isl]\n\t" \
  "v_cndmask_b32_e64 %[nxt], %[m], %[nxt], %[isl]\n\t" \
  "v_cndmask_b32_e64 %[islv], 0, 1, %[isl]\n\t"

#define S3S_EL_FRONT_LZF \
  "v_and_b32_e32 %[ty], 0xff, %[d0]\n\t" \
  "v_bfe_u32 %[b1], %[d0], 8, 8\n\t" \
  "v_bfe_u32 %[w16], %[d0], 16, 8\n\t" \
  "v_cmp_gt_u32_e64 %[isl], 32, %[ty]\n\t" \
  "v_add_u32_e32 %[len], 1, %[ty]\n\t" \
  "v_lshrrev_b32_e32 %[n6], 5, %[ty]\n\t" \
  "v_add3_u32 %[nxt], %[cpos], %[len], 1\n\t" \
  "v_cmp_lt_i32_e32 vcc, %[clen], %[nxt]\n\t" \
  "v_cmp_eq_u32_e64 %[t1], 7, %[n6]\n\t" \
  "s_and_b64 %[cx], %[isl], vcc\n\t" \
  "v_add_u32_e32 %[vt], 7, %[b1]\n\t" \
  "v_add_u32_e32 %[vt2], 1, %[lane]\n\t" \
  "v_cndmask_b32_e64 %[vt], %[n6], %[vt], %[t1]\n\t" \
  "v_lshlrev_b32_e32 %[vt2], 16, %[vt2]\n\t" \
  "v_add_u32_e32 %[vt], 2, %[vt]\n\t" \
  "v_and_b32_e32 %[m], 0x1f, %[ty]\n\t" \
  "v_cndmask_b32_e64 %[jmp], %[b1], %[w16], %[t1]\n\t" \
  "v_lshlrev_b32_e32 %[vt], 16, %[vt]\n\t" \
  "v_lshl_or_b32 %[jmp], %[m], 8, %[jmp]\n\t" \
  "v_cndmask_b32_e64 %[m], 2, 3, %[t1]\n\t" \
  "v_add_u32_e32 %[jmp], 1, %[jmp]\n\t" \
  "v_add_u32_e32 %[m], %[cpos], %[m]\n\t" \
  "v_cndmask_b32_e64 %[r0], %[vt], %[len], %[isl]\n\t" \
  "v_cndmask_b32_e64 %[r1], %[jmp], %[vt2], %[isl]\n\t" \
  "v_cndmask_b32_e64 %[nxt], %[m], %[nxt], %[isl]\n\t" \
  "v_cndmask_b32_e64 %[islv], 0, 1, %[isl]\n\t"

// one round of the pointer-doubling walk inside the Snappy parse block (operands of that asm statement; written out six times:
// the interpreter of tests/isa reads the compiler's text, where an assembler loop would still be a directive)
#define S3S_SNAPPY_DBL_ROUND \
  "v_cmp_gt_i32_e32 vcc, 64, %[jmp]\n\t" \
  "v_lshlrev_b32_e32 %[vt2], 2, %[jmp]\n\t" \
  "s_and_b64 %[t1], %[rch], vcc\n\t" \
  "s_and_saveexec_b64 %[sv], %[t1]\n\t" \
  "v_add_u32_e32 %[vt], %[mark], %[jmp]\n\t" \
  "ds_write_b8 %[vt], %[one]\n\t" \
  "s_mov_b64 exec, %[sv]\n\t" \
  "ds_read_u8 %[m], %[lma]\n\t" \
  "ds_bpermute_b32 %[vt], %[vt2], %[jmp]\n\t" \
  "s_waitcnt lgkmcnt(0)\n\t" \
  "v_cmp_ne_u32_e64 %[t1], 0, %[m]\n\t" \
  "v_cndmask_b32_e32 %[jmp], %[c64], %[vt], vcc\n\t" \
  "s_or_b64 %[rch], %[rch], %[t1]\n\t"

// ---- the Snappy / LZF parse block: ONE asm statement (front end FRONT + walk + join + stores), see its use in the kernel ----
#define S3S_EL_PARSE_BLOCK(FRONT) \
          asm volatile( \
              "v_mov_b32_e32 %[one], 1\n\t" \
              "v_mov_b32_e32 %[c64], 64\n\t" \
              "v_add_u32_e32 %[lma], %[mark], %[lane]\n\t" \
              ".Ls_loop%=:\n\t" \
              "v_add_u32_e32 %[cpos], %[ip], %[lane]\n\t" \
              "global_load_dword %[d0], %[cpos], %[c]\n\t" \
              "s_waitcnt vmcnt(0)\n\t" \
              FRONT \
              "v_subrev_u32_e32 %[nxt], %[ip], %[nxt]\n\t"  /* next token, window-relative */ \
              "s_mov_b64 %[m64], 0\n\t" \
              "s_cmp_lt_u32 %[lt], 12\n\t" \
              "v_cndmask_b32_e64 %[nrel], %[nxt], -1, %[cx]\n\t" \
              "s_cbranch_scc1 .Ls_swalk%=\n\t" \
              /* ---- pointer doubling: six rounds of "lanes on the chain mark the lane 2^k tokens behind them" ---- */ \
              "v_min_i32_e32 %[jmp], 64, %[nxt]\n\t" \
              "v_mov_b32_e32 %[vt], 0\n\t" \
              "v_cndmask_b32_e64 %[jmp], %[jmp], %[c64], %[cx]\n\t" \
              "ds_write_b8 %[lma], %[vt]\n\t" \
              "s_mov_b64 %[rch], 1\n\t" \
              S3S_SNAPPY_DBL_ROUND S3S_SNAPPY_DBL_ROUND S3S_SNAPPY_DBL_ROUND S3S_SNAPPY_DBL_ROUND S3S_SNAPPY_DBL_ROUND S3S_SNAPPY_DBL_ROUND \
              "s_and_b64 %[t1], %[rch], %[cx]\n\t"  /* the (at most one) byte-wise token the chain runs into */ \
              "s_andn2_b64 %[m64], %[rch], %[t1]\n\t" \
              "s_cmp_lg_u64 %[t1], 0\n\t" \
              "s_cbranch_scc0 .Ls_dnocx%=\n\t" \
              "s_ff1_i32_b64 %[rel], %[t1]\n\t" \
              "s_branch .Ls_wdone%=\n\t" \
              ".Ls_dnocx%=:\n\t" \
              "s_flbit_i32_b64 %[n], %[m64]\n\t"  /* (lane 0 is on the chain: mask != 0) */ \
              "s_sub_i32 %[n], 63, %[n]\n\t" \
              "v_readlane_b32 %[rel], %[nrel], %[n]\n\t" \
              "s_branch .Ls_wdone%=\n\t" \
              /* ---- the scalar walk (few tokens) ---- */ \
              ".Ls_swalk%=:\n\t" \
              "s_mov_b32 %[rel], 0\n\t" \
              "v_readlane_b32 %[n], %[nrel], %[rel]\n\t" \
              "s_cmp_lt_u32 %[n], 64\n\t" \
              "s_cbranch_scc0 .Ls_wout%=\n\t" \
              ".Ls_wnext%=:\n\t" \
              "s_bitset1_b64 %[m64], %[rel]\n\t" \
              "s_mov_b32 %[rel], %[n]\n\t" \
              "v_readlane_b32 %[n], %[nrel], %[rel]\n\t" \
              "s_cmp_lt_u32 %[n], 64\n\t" \
              "s_cbranch_scc1 .Ls_wnext%=\n\t" \
              ".Ls_wout%=:\n\t" \
              "s_cmp_gt_i32 %[n], -1\n\t" \
              "s_cbranch_scc0 .Ls_wdone%=\n\t" \
              "s_bitset1_b64 %[m64], %[rel]\n\t" \
              "s_mov_b32 %[rel], %[n]\n\t" \
              ".Ls_wdone%=:\n\t" \
              "s_bcnt1_i32_b64 %[lt], %[m64]\n\t"  /* tokens of this window (scc = any); picks the next walk */ \
              "s_mov_b32 %[code], 1\n\t" \
              "s_cbranch_scc0 .Ls_exit%=\n\t" \
              /* ---- which copies join the literal in front of them: literal tokens mark their successor ---- */ \
              "s_and_b64 %[lm], %[isl], %[m64]\n\t" \
              "v_mov_b32_e32 %[vt], 0\n\t" \
              "v_cmp_gt_u32_e32 vcc, 64, %[nxt]\n\t"  /* successor inside the window */ \
              "ds_write_b8 %[lma], %[vt]\n\t" \
              "s_and_b64 %[t1], %[lm], vcc\n\t" \
              "s_and_saveexec_b64 %[sv], %[t1]\n\t" \
              "v_add_u32_e32 %[vt], %[mark], %[nxt]\n\t" \
              "ds_write_b8 %[vt], %[one]\n\t" \
              "s_mov_b64 exec, %[sv]\n\t" \
              "ds_read_u8 %[m], %[lma]\n\t" \
              "s_andn2_b64 %[J], %[m64], %[isl]\n\t"  /* copy tokens ... */ \
              "s_waitcnt lgkmcnt(0)\n\t" \
              "v_cmp_ne_u32_e64 %[t1], 0, %[m]\n\t" \
              "s_nop 0\n\t" \
              "s_or_b32 %[n], %[open], 0\n\t"  /* (the first token's predecessor is the batch's last record) */ \
              "s_mov_b32 %[cnt], 0\n\t" \
              "s_mov_b64 %[S], 0\n\t" \
              "s_cmp_lg_u32 %[open], 0\n\t" \
              "s_cbranch_scc0 .Ls_noopen%=\n\t" \
              "s_bitset1_b64 %[t1], %[cnt]\n\t"  /* ... lane 0 (always a token) counts as marked */ \
              ".Ls_noopen%=:\n\t" \
              "s_and_b64 %[J], %[J], %[t1]\n\t"  /* ... behind a literal: joined */ \
              "s_andn2_b64 %[S], %[m64], %[J]\n\t"  /* tokens that start a record */ \
              "s_bcnt1_i32_b64 %[cnt], %[S]\n\t" \
              "s_add_i32 %[t], %[nseq], %[cnt]\n\t" \
              "s_mov_b32 %[code], 2\n\t" \
              "s_cmp_gt_i32 %[t], 64\n\t" \
              "s_cbranch_scc1 .Ls_exit%=\n\t" \
              /* ---- stores: a record per S lane; a joined copy writes its length and offset into the record in front ---- */ \
              "s_cmp_eq_u32 %[nseq], 0\n\t" \
              "s_cselect_b32 %[sbase], %[ip], %[sbase]\n\t" \
              "s_sub_i32 %[cnt], %[ip], %[sbase]\n\t" \
              "s_lshl_b32 %[cnt], %[cnt], 16\n\t" \
              "s_mov_b64 vcc, %[S]\n\t" \
              "v_mbcnt_lo_u32_b32 %[vt], vcc_lo, 0\n\t" \
              "v_mbcnt_hi_u32_b32 %[vt], vcc_hi, %[vt]\n\t" \
              "v_add_u32_e32 %[vt2], %[cnt], %[r1]\n\t" \
              "v_add_lshl_u32 %[vt], %[vt], %[nseq], 3\n\t" \
              "v_add_u32_e32 %[vt], %[rec], %[vt]\n\t" \
              "v_lshrrev_b32_e32 %[m], 16, %[r0]\n\t" \
              "s_mov_b64 exec, %[S]\n\t" \
              "ds_write_b32 %[vt], %[r0]\n\t" \
              "ds_write_b32 %[vt], %[vt2] offset:4\n\t" \
              "s_mov_b64 exec, %[J]\n\t" \
              "v_add_u32_e32 %[vt], -8, %[vt]\n\t" \
              "ds_write_b16 %[vt], %[m] offset:2\n\t" \
              "ds_write_b16 %[vt], %[r1] offset:4\n\t" \
              "s_mov_b64 exec, -1\n\t" \
              /* the window's last token is a literal: the next window's first copy may join it */ \
              "s_flbit_i32_b64 %[n], %[m64]\n\t" \
              "s_sub_i32 %[n], 63, %[n]\n\t" \
              "s_bitcmp1_b64 %[lm], %[n]\n\t" \
              "s_cselect_b32 %[open], 1, 0\n\t" \
              "s_mov_b32 %[nseq], %[t]\n\t" \
              "s_add_i32 %[ip], %[ip], %[rel]\n\t" \
              "s_add_i32 %[t], %[ip], 67\n\t" \
              "s_mov_b32 %[code], 0\n\t" \
              "s_cmp_le_i32 %[t], %[clen]\n\t" \
              "s_cbranch_scc1 .Ls_loop%=\n\t" \
              ".Ls_exit%=:" \
              : [code] "=&s"(code), [m64] "=&s"(mask), [rel] "=&s"(rel), [r0] "=&v"(r0), [r1] "=&v"(r1), [islv] "=&v"(v_isl), \
                [ip] "+s"(ip), [nseq] "+s"(nseq), [sbase] "+s"(sbase), [lt] "+s"(lt_), [open] "+s"(open_), \
                [n] "=&s"(n_), [cnt] "=&s"(cnt_), [t] "=&s"(t_), [cx] "=&s"(cx_), [sv] "=&s"(sv_), [isl] "=&s"(isl_), \
                [t1] "=&s"(t1_), [lm] "=&s"(lm_), [S] "=&s"(S_), [J] "=&s"(J_), [rch] "=&s"(rch_), \
                [cpos] "=&v"(v_cpos), [d0] "=&v"(v_d0), [ty] "=&v"(v_ty), [n6] "=&v"(v_n6), [b1] "=&v"(v_b1), [w16] "=&v"(v_w16), \
                [len] "=&v"(v_len), [hdr] "=&v"(v_hdr), [vt] "=&v"(v_t), [vt2] "=&v"(v_t2), [nxt] "=&v"(v_nxt), [nrel] "=&v"(v_nrel), \
                [jmp] "=&v"(v_jmp), [m] "=&v"(v_m), [one] "=&v"(n_one), [c64] "=&v"(n_c64), [lma] "=&v"(n_lma) \
              : [c] "s"(c), [clen] "s"(clen), [lane] "v"(lane), [rec] "s"(lds_addr(rec)), [mark] "s"(markb) \
              : "vcc", "scc", "memory");

// kFmt selects the front end (token parse + byte-wise path); batches, rounds and the output window are the same:
// a Snappy element is a sequence with either literals only (ml = 0) or a copy only (lit = 0).
template <int kFmt>
__global__ __launch_bounds__(kWave, 8) void batch_decode_kernel(
    const uint8_t* __restrict__ comp, const Frame* __restrict__ frames, int32_t n_frames,
    const int64_t* __restrict__ frame_out, uint8_t* dst, int32_t* __restrict__ status
) {
  __shared__ __attribute__((aligned(16))) uint8_t win[kBWin + kBPad];
  __shared__ uint2 rec[kWave];
#ifdef S3S_DEC_RING
  __shared__ __attribute__((aligned(4))) uint8_t sring[kSRing + 8];
#endif
  const int f = blockIdx.x;
  if (f >= n_frames) return;
  const Frame fr = frames[f];
  const int olen = fr.orig_len, clen = fr.comp_len;
  const int lane = threadIdx.x;
  // LZ4 blocks above 32 KiB (a writer configured with a larger spark.io.compression.lz4.blockSize, S3ShuffleReader.scala:57-59)
  // are decoded here as well since round 4: the records of a batch keep stream offsets and output positions RELATIVE to
  // the batch's first token / first output byte (a batch of 64 fast-path sequences spans < 18 KiB of stream and < 35 KiB of
  // output), so nothing but the frame header's own 32-bit lengths depends on the block size.  kBatchMaxBlock = lz4-java's
  // largest block (1 << 25).  Snappy chunks stay at 32 KiB (snappy-java's block size cannot exceed the fragment size here).
  if (olen > (kFmt == kFmtSnappy ? kMaxBlock : kFmt == kFmtLzf ? 0xFFFF : kBatchMaxBlock) && (kFmt == kFmtSnappy || fr.method != 0x10)) {
    if (lane == 0) atomicExch(status, S3S_E_UNSUPPORTED);
    return;
  }
  if (olen == 0 && (kFmt == kFmtLz4 || clen == 0)) return;  // (end-of-stream frame; a frame the batched call skips)
  const uint8_t* c = comp + fr.comp_off;
  uint8_t* out = dst + frame_out[f];
  bool bad = false;
  int ip_start = 0;
#ifndef S3S_DEC_NO_PREFETCH
  // Touch the frame's compressed bytes once (a few independent 1 KiB rows, one wait): the parse reads the stream through
  // dependent loads, one new cache line every other window, and a line that comes from HBM instead of L2 is on the chain.
  {
    uint32_t acc = 0;
    const int lim = clen < kMaxBlock + kMaxBlock / 6 + 64 ? clen : kMaxBlock + kMaxBlock / 6 + 64;
    for (int i = lane * 16; i + 16 <= lim; i += kWave * 16) {
      uint4 x;
      __builtin_memcpy(&x, c + i, 16);
      acc ^= x.x ^ x.y ^ x.z ^ x.w;
    }
    if (acc == 0x12345678u && clen < 0) win[0] = 1;  // (never taken: keeps the loads)
  }
#endif
  if (kFmt == kFmtSnappy) {
    // preamble: varint32 uncompressed length (scalar, once per block)
    uint32_t ulen = 0;
    int vs = 0;
    for (;;) {
      if (ip_start >= clen || vs > 28) { bad = true; break; }
      const uint32_t b = __builtin_amdgcn_readfirstlane((uint32_t)c[ip_start]);
      ip_start++;
      ulen |= (b & 0x7fu) << vs;
      if (!(b & 0x80u)) break;
      vs += 7;
    }
    if (!bad && (int)ulen != olen) bad = true;
  }
  if (bad) {
  } else if ((kFmt == kFmtSnappy ? clen > kMaxBlock + kMaxBlock / 6 + 64
                                 : (int64_t)clen > (int64_t)olen + olen / (kFmt == kFmtLzf ? 16 : 255) + 16) &&
             !(kFmt != kFmtSnappy && fr.method == 0x10)) {
    bad = true;  // no block of that decoded size is that long (LZ4: LZ4_compressBound; a compressed frame is shorter than its block anyway)
  } else if (kFmt != kFmtSnappy && fr.method == 0x10) {  // stored frame (LZ4Block RAW / LZF non-compressed chunk)
    for (int j = lane * 4; j < olen; j += kWave * 4) {
      if (j + 4 <= olen) {
        const uint32_t x = g_ld32(c + j);
        __builtin_memcpy(out + j, &x, 4);
      } else {
        for (int k = j; k < olen; k++) out[k] = c[k];
      }
    }
  } else {
    const int sh = (int)(reinterpret_cast<uintptr_t>(out) & 15u);  // window index of output byte o: o + sh - wb
    int wb = 0, flushed = 0, op = 0, ip = ip_start, nseq = 0;      // wave-uniform
    bool need_drain = false;  // stores of a flush may still be in flight (matters to far matches only)
    int last_tokens = 0;      // tokens found in the previous parse window (picks the walk for this one)
    bool open_lit = false;    // Snappy: the batch's last record is a literal element that a following copy may join
    int sbase = 0;            // stream position the batch's records count from (set with the batch's first record)
    BT_DECL;

    // ---- window management ---------------------------------------------------------------------------
    auto flush_to = [&](int upto) __attribute__((always_inline)) {  // window -> out for output bytes [flushed, upto)
      int o = flushed;
      int head = (-(o + sh)) & 15;
      head = head < upto - o ? head : upto - o;
      if (lane < head) out[o + lane] = win[o + sh - wb + lane];
      o += head;
      const int body = (upto - o) & ~15;
      for (int j = lane * 16; j < body; j += kWave * 16) {
        const uint4 x = *reinterpret_cast<const uint4*>(win + (o + sh - wb) + j);
        *reinterpret_cast<uint4*>(out + o + j) = x;
      }
      o += body;
      if (lane < upto - o) out[o + lane] = win[o + sh - wb + lane];
      flushed = upto;
      need_drain = true;
    };
    auto slide = [&]() __attribute__((always_inline)) {
      flush_to(op);
      const int nwb = (op + sh - kBHist) & ~15;
      if (nwb > wb) {
        const int shift = nwb - wb, keep = op + sh - nwb;
        for (int j = lane * 16; j < keep; j += kWave * 16) {
          const uint4 x = *reinterpret_cast<const uint4*>(win + shift + j);
          *reinterpret_cast<uint4*>(win + j) = x;
        }
        wb = nwb;
      }
    };
    auto room = [&]() __attribute__((always_inline)) -> int { return wb + kBWin - (op + sh); };

    // l0 stream bytes from position r0 to window index didx, by the whole wave (arguments uniform)
    auto stream_to_window = [&](int didx, int r0, int l0) __attribute__((always_inline)) {
      uint8_t* d = win + didx;
      const int body = l0 >> 2;
      for (int j = lane; j < body; j += kWave) lds_st32(d + 4 * j, g_ld32(c + r0 + 4 * j));
      if (lane < (l0 & 3)) d[4 * body + lane] = c[r0 + 4 * body + lane];
    };

    // ---- generic emitters: any length, chunked by the room of the window (all arguments uniform) ------
    auto emit_literals = [&](int src, int n) __attribute__((always_inline)) -> bool {
      if (n < 0 || n > clen - src || n > olen - op) return false;
      while (n > 0) {
        if (room() == 0) slide();
        int k = room();
        k = k < n ? k : n;
        stream_to_window(op + sh - wb, src, k);
        op += k;
        src += k;
        n -= k;
      }
      return true;
    };
    auto emit_match = [&](int off, int ml) __attribute__((always_inline)) -> bool {
      if (off <= 0 || off > op || ml > olen - op) return false;
      while (ml > 0) {
        if (room() == 0) slide();
        int k = room();
        k = k < ml ? k : ml;
        const int srco = op - off;
        uint8_t* d = win + (op + sh - wb);
        if (srco + sh < wb) {
          // far: the source left the window; it was flushed, read it back from L2
          const int avail = wb - sh - srco;
          k = k < avail ? k : avail;
          if (need_drain) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            need_drain = false;
          }
          for (int j = lane; j < k; j += kWave) d[j] = (uint8_t)l2_ld8(out + srco + j);
        } else {
          const uint8_t* s = win + (srco + sh - wb);
          if (off >= kWave) {
            for (int j0 = 0; j0 < k; j0 += kWave) {  // each step reads what earlier steps (or older data) wrote
              const int j = j0 + lane;
              if (j < k) d[j] = s[j];
            }
          } else {
            // overlapping: a periodic pattern of the `off` bytes in front of the destination
            int idx = lane % off;
            const int r = kWave % off;
            for (int j0 = 0; j0 < k; j0 += kWave) {
              const int j = j0 + lane;
              if (j < k) d[j] = s[idx];
              idx += r;
              idx = idx >= off ? idx - off : idx;
            }
          }
        }
        op += k;
        ml -= k;
      }
      return true;
    };

    // A match whose output range is known to fit into the window (no slide, `op` untouched): far part from L2,
    // then the part inside the window, plain or as a periodic pattern.  Arguments uniform.
    auto copy_match_fit = [&](int ms, int off, int ml) __attribute__((always_inline)) {
      while (ml > 0) {
        int k = ml;
        const int srco = ms - off;
        uint8_t* d = win + (ms + sh - wb);
        if (srco + sh < wb) {
          const int avail = wb - sh - srco;
          k = k < avail ? k : avail;
          if (need_drain) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            need_drain = false;
          }
          for (int j = lane; j < k; j += kWave) d[j] = (uint8_t)l2_ld8(out + srco + j);
        } else {
          const uint8_t* s = win + (srco + sh - wb);
          if (off >= kWave) {
            for (int j0 = 0; j0 < k; j0 += kWave) {
              const int j = j0 + lane;
              if (j < k) d[j] = s[j];
            }
          } else {
            int idx = lane % off;
            const int r = kWave % off;
            for (int j0 = 0; j0 < k; j0 += kWave) {
              const int j = j0 + lane;
              if (j < k) d[j] = s[idx];
              idx += r;
              idx = idx >= off ? idx - off : idx;
            }
          }
        }
        ms += k;
        ml -= k;
      }
    };

    // ---- the batch: one sequence per lane -------------------------------------------------------------
    auto flush_batch = [&]() __attribute__((always_inline)) -> bool {
      if (nseq == 0) return true;
      BT_MARK(0);
      const uint2 rc = rec[lane];
      const bool act = lane < nseq;
      const int lit = act ? (int)(rc.x & 0xffffu) : 0;
      const int ml = act ? (int)(rc.x >> 16) : 0;
      const int off = (int)(rc.y & 0xffffu);
      const int src = (int)(rc.y >> 16) + sbase;
      const int opb = op;  // the packed words below carry output positions relative to the batch's first byte
      // Pieces: a match of up to 64 bytes that is not an odd periodic pattern is copied as 1..4 PIECES of 16 bytes, one
      // lane per piece (`npc`); one inclusive prefix sum gives the output positions (low 22 bits; a lane's share is
      // clamped so that a malformed batch cannot carry into the piece count: its `end` is above olen either way) and
      // the piece positions (high bits).
      const bool patok = off >= ml || off == 1 || off == 2 || off == 4;
      const int npc = (ml > 0 && ml <= kWave && patok) ? (ml + 15) >> 4 : 0;
      const int len0 = lit + ml;
      int pk = (len0 < kMaxBlock + 1 ? len0 : kMaxBlock + 1) | (npc << 22);
      pk = wave_scan_add(pk);
      const int end = (pk & 0x3fffff) + op, lpe = pk >> 22, lpx = lpe - npc;  // (pieces up to and including / in front of this lane)
      const int start = end - lit - ml, mstart = end - ml;
      const bool wrong = act && ((ml > 0 && (off == 0 || off > mstart)) || end > olen || src + lit > clen);
      if (ballot64(wrong)) return false;
      // piece list (in the record array, which is free until the next parse window): piece p -> lane | index << 8
      {
        uint16_t* pl = reinterpret_cast<uint16_t*>(rec) + lpx;
        if (npc > 0) pl[0] = (uint16_t)lane;
        if (npc > 1) pl[1] = (uint16_t)(lane | 0x100);
        if (npc > 2) pl[2] = (uint16_t)(lane | 0x200);
        if (npc > 3) pl[3] = (uint16_t)(lane | 0x300);
      }
      // ---- where a match really reads from ----
      // The bytes a match needs are [a, a + need): its source, or the `off` bytes in front of it if it overlaps its
      // own output.  count(e) = number of sequences t with mstart[t] < e (binary search over the sorted mstart): the
      // match may run in a round that starts at sequence cur iff count(a + need) <= cur.  Two refinements cut the
      // dependency chains (TeraSort: every record copies from the record before it):
      //   * a range that lies in the literals of the batch (or in front of its first match) is final from the start,
      //     because all literals are copied before the first round;
      //   * a range that lies inside the output of ONE earlier match t reads what t copied: out[q] = out[q - off[t]]
      //     for every q there, so it can read from t's source instead — and so on along the chain (pointer doubling
      //     over the lanes; a match that overlaps its own output ends a chain, its bytes are not a plain shift).
      const uint32_t me = (uint32_t)(mstart - opb) | ((uint32_t)(end - opb) << 16);
      auto count_below = [&](int e) __attribute__((always_inline)) -> int {
        int n = 0;
#pragma unroll
        for (int step = 32; step >= 1; step >>= 1) {
          const int probe = n + step - 1;  // candidate index
          const int mv = (int)(__shfl(me, probe & 63) & 0xffffu);  // (relative to opb, as is e)
          const bool take = probe < nseq && mv < e;
          n = take ? n + step : n;
        }
        return n;
      };
      const int need = off < ml ? off : ml;
      int a2 = mstart - off;
      int dep = count_below(a2 + need - opb);  // (<= lane: mstart[t] < a + need <= mstart[lane])
      const int t1 = dep > 0 ? dep - 1 : 0;
      const uint32_t m1 = __shfl(me, t1);
      const int ms1 = (int)(m1 & 0xffffu) + opb, en1 = (int)(m1 >> 16) + opb;
      bool ground = ml == 0 || dep == 0 || a2 >= en1;  // in the literals of sequence `dep` / in front of the batch's matches
      const bool inside = ml > 0 && dep > 0 && a2 >= ms1 && a2 + need <= en1;  // inside the output of match t1
      if (ballot64(inside)) {
        int shift = off;  // how far the output of this match is from the bytes it is a copy of
        int nx = (inside && off >= ml) ? t1 : -1;
        while (ballot64(nx >= 0)) {
          const int s2 = __shfl(shift, nx & 63), n2 = __shfl(nx, nx & 63);
          shift += nx >= 0 ? s2 : 0;
          nx = nx >= 0 ? n2 : nx;
        }
        const int st1 = __shfl(shift, t1);
        a2 -= inside ? st1 : 0;
        dep = count_below(a2 + need - opb);
        const int t2 = dep > 0 ? dep - 1 : 0;
        const int en2 = (int)(__shfl(me, t2) >> 16) + opb;
        ground = ml == 0 || dep == 0 || a2 >= en2;
      }
      if (ground) dep = 0;
      // what a piece's lane fetches from its sequence
      // (a2 may lie up to a whole block in front of the batch: 28 signed bits; the period is 0, 1, 2 or 4)
      const uint32_t W1 = (uint32_t)(mstart - opb) | ((uint32_t)ml << 16);
      const uint32_t W2 = ((uint32_t)(a2 - opb) & 0x0fffffffu) | ((uint32_t)(off < ml ? off : 0) << 28);
      BT_MARK(1);
      int s0 = 0;
      while (s0 < nseq) {
        // sequences [s0, s1) fit into the window
        const uint64_t fits = ballot64(act && lane >= s0 && end + sh <= wb + kBWin);
        const uint64_t fr0 = fits >> s0;
        const int nfit = (~fr0 == 0ull) ? kWave - s0 : __builtin_ctzll(~fr0);
        const int s1 = s0 + nfit;
        if (nfit == 0) {
          if (op + sh - wb > kBHist + 16) {
            slide();
            BT_MARK(6);
            continue;
          }
          // one sequence larger than the free part of a freshly slid window
          const int l0 = __builtin_amdgcn_readlane(lit, s0), m0 = __builtin_amdgcn_readlane(ml, s0);
          const int o0 = __builtin_amdgcn_readlane(off, s0), r0 = __builtin_amdgcn_readlane(src, s0);
          if (!emit_literals(r0, l0)) return false;
          if (m0 > 0 && !emit_match(o0, m0)) return false;
          s0++;
          BT_MARK(7);
          continue;
        }
        const bool in = lane >= s0 && lane < s1;
        // ---- literals of [s0, s1) ----
        {
          uint8_t* d = win + (start + sh - wb);
          const bool smalll = in && lit <= kSmallLit;
          if (ballot64(smalll && lit > 0)) {
            if (smalll && lit > 0) {
              if (src + 16 <= clen) {
                u32x4 x;
                __builtin_memcpy(&x, c + src, 16);
                lds_store16(d, x, lit);
              } else {
                for (int j = 0; j < lit; j++) d[j] = c[src + j];  // a literal run in the last bytes of the block
              }
            }
          }
          uint64_t big = ballot64(in && lit > kSmallLit);
          while (big) {
            const int s = __builtin_ctzll(big);
            big &= big - 1;
            const int l0 = __builtin_amdgcn_readlane(lit, s), r0 = __builtin_amdgcn_readlane(src, s);
            const int st0 = __builtin_amdgcn_readlane(start, s);
            stream_to_window(st0 + sh - wb, r0, l0);
          }
        }
        BT_MARK(2);
        // ---- matches of [s0, s1) in dependency rounds ----
        // A round takes the longest run of sequences from `cur` whose matches have pieces (see above), read bytes
        // that are final (dep <= cur) wholly inside or wholly in front of the window, and need at most 64 pieces
        // together: every piece's lane reads its 16 bytes, then writes them.  Anything else (longer than 64 bytes,
        // an odd periodic pattern, a source that straddles the window base) goes alone, by the whole wave.
        // (a source in front of the window was flushed long ago: those lanes read it back from L2)
        const bool near = a2 + sh >= wb, far = a2 + need + sh <= wb;
        const bool pieces = in && npc > 0 && (near || far);
        // wave-uniform facts about [s0, s1), taken once instead of in every round (round 6): is any source in front of the
        // window (those lanes read L2), is any match a period-1 / 2 / 4 splat
        const int any_far = ballot64(pieces & !near) != 0ull ? 1 : 0;
        const int any_pat = (ballot64(pieces) & ballot64(off < ml)) != 0ull ? 1 : 0;
        if (need_drain && any_far) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          need_drain = false;
        }
        const uint64_t PM = ballot64(pieces) | (ballot64(in) & ballot64(ml == 0));
        const uint16_t* plist = reinterpret_cast<const uint16_t*>(rec);
        const int lpe64 = lpe - kWave;  // (lpe - lp0 <= 64  <=>  lpe - 64 <= lp0)
        int cur = s0;
        while (cur < s1) {
          const int lp0 = __builtin_amdgcn_readlane(lpx, cur);  // pieces in front of sequence cur
          const uint64_t am = (ballot64(dep <= cur) & ballot64(lpe64 <= lp0) & PM) >> cur;
          int run;  // the first zero bit of am, or all that is left (s_ff0 answers -1 when there is no zero bit)
          asm("s_ff0_i32_b64 %0, %1\n\ts_min_u32 %0, %0, %2" : "=&s"(run) : "s"(am), "s"(kWave - cur) : "scc");
          if (run == 0) {
            const int m0 = __builtin_amdgcn_readlane(ml, cur), o0 = __builtin_amdgcn_readlane(off, cur);
            const int ms0 = __builtin_amdgcn_readlane(mstart, cur);
            copy_match_fit(ms0, o0, m0);
            cur++;
            BT_MARK(5);
            BT_COUNT(11);
            continue;
          }
          const int np = __builtin_amdgcn_readlane(lpe, cur + run - 1) - lp0;  // <= 64
          const int pi = lp0 + lane < 255 ? lp0 + lane : 255;
          const uint32_t pe = plist[pi];
          const int sq = (int)(pe & 63u), k16 = (int)(pe >> 8) << 4;
          const uint32_t w1 = __shfl(W1, sq), w2 = __shfl(W2, sq);
          const bool actp = lane < np;
          const int msq = (int)(w1 & 0xffffu) + opb, mlq = (int)(w1 >> 16);
          const int aq = ((int)(w2 << 4) >> 4) + opb, pat = (int)(w2 >> 28);
          int n = mlq - k16;
          n = n < 16 ? n : 16;
          n = actp ? n : 0;  // (an idle lane stores nothing)
          const int so = aq + (pat ? 0 : k16);
          const bool nearp = aq + sh >= wb;
          // every lane reads 16 bytes of the window (an idle lane, or one whose source is in L2: the first 16)
          u32x4 x = lds_ld128u(win + ((actp && nearp) ? so + sh - wb : 0));
          if (any_far) {
            if (actp && !nearp) {
              const uint8_t* gsrc = out + so;
              x.x = l2_ld32(gsrc);
              if (n > 4 && !pat) x.y = l2_ld32(gsrc + 4);
              if (n > 8 && !pat) x.z = l2_ld32(gsrc + 8);
              if (n > 12 && !pat) x.w = l2_ld32(gsrc + 12);
            }
          }
          if (any_pat) {  // period 1, 2 or 4: every dword of the output is the same
            const uint32_t sp = pat == 1 ? (x.x & 0xffu) * 0x01010101u : (pat == 2 ? (x.x & 0xffffu) * 0x00010001u : x.x);
            if (pat) x.x = x.y = x.z = x.w = sp;
          }
          lds_store16(win + (msq + k16 + sh - wb), x, n);
          cur += run;
          BT_MARK(3);
          BT_COUNT(9);
        }
        op = __builtin_amdgcn_readlane(end, s1 - 1);
        s0 = s1;
      }
      nseq = 0;
      return true;
    };

    // ---- one sequence, byte by byte (255-chains, the last sequence of the block) -------------------------
    // returns 1: block finished, 0: go on, -1: malformed
    auto slow_sequence = [&]() __attribute__((always_inline)) -> int {
      int ips = ip;
      if (ips >= clen) return -1;
      if constexpr (kFmt == kFmtLzf) {
        // one element: ctrl < 32: ctrl + 1 literals; else a back reference of (ctrl >> 5 [7: + next byte]) + 2 bytes,
        // ((ctrl & 31) << 8 | next byte) + 1 back (liblzf's lzf_decompress; every bound checked here or by the emitters)
        const uint32_t ctrl = __builtin_amdgcn_readfirstlane((uint32_t)c[ips]);
        ips++;
        if (ctrl < 32u) {
          const int n = (int)ctrl + 1;
          if (n > clen - ips || n > olen - op) return -1;
          if (!emit_literals(ips, n)) return -1;
          ip = ips + n;
          return 0;
        }
        int len = (int)(ctrl >> 5);
        if (len == 7) {
          if (ips >= clen) return -1;
          len += (int)__builtin_amdgcn_readfirstlane((uint32_t)c[ips]);
          ips++;
        }
        if (ips >= clen) return -1;
        const int off = (int)(((ctrl & 0x1fu) << 8) | __builtin_amdgcn_readfirstlane((uint32_t)c[ips])) + 1;
        ips++;
        ip = ips;
        if (!emit_match(off, len + 2)) return -1;
        return 0;
      }
      if constexpr (kFmt == kFmtSnappy) {
        // one element: literal (length in the tag or in 1-4 bytes behind it) or copy with 1 / 2 / 4 offset bytes
        const uint32_t tag = __builtin_amdgcn_readfirstlane((uint32_t)c[ips]);
        ips++;
        const uint32_t ty = tag & 3u;
        if (ty == 0u) {
          uint32_t len = tag >> 2;
          if (len >= 60u) {
            const int nb = (int)len - 59;
            if (clen - ips < nb) return -1;
            uint32_t v = 0;
            for (int k = 0; k < nb; k++) v |= __builtin_amdgcn_readfirstlane((uint32_t)c[ips + k]) << (8 * k);
            ips += nb;
            len = v;
            if (len >= 0x7fffffffu) return -1;
          }
          const int n = (int)len + 1;
          if (n > clen - ips || n > olen - op) return -1;
          if (!emit_literals(ips, n)) return -1;
          ip = ips + n;
          return 0;
        }
        int len, off;
        if (ty == 1u) {
          if (clen - ips < 1) return -1;
          len = 4 + (int)((tag >> 2) & 7u);
          off = (int)(((tag >> 5) << 8) | __builtin_amdgcn_readfirstlane((uint32_t)c[ips]));
          ips += 1;
        } else if (ty == 2u) {
          if (clen - ips < 2) return -1;
          len = (int)(tag >> 2) + 1;
          off = (int)__builtin_amdgcn_readfirstlane((uint32_t)c[ips] | ((uint32_t)c[ips + 1] << 8));
          ips += 2;
        } else {
          if (clen - ips < 4) return -1;
          len = (int)(tag >> 2) + 1;
          const uint32_t o4 = __builtin_amdgcn_readfirstlane((uint32_t)c[ips] | ((uint32_t)c[ips + 1] << 8) |
                                                              ((uint32_t)c[ips + 2] << 16) | ((uint32_t)c[ips + 3] << 24));
          if (o4 > 0x7fffffffu) return -1;
          off = (int)o4;
          ips += 4;
        }
        ip = ips;
        if (!emit_match(off, len)) return -1;
        return 0;
      }
      const uint32_t token = __builtin_amdgcn_readfirstlane((uint32_t)c[ips]);
      ips++;
      int lit = (int)(token >> 4);
      if (lit == 15) {
        uint32_t b;
        do {
          if (ips >= clen) return -1;
          b = __builtin_amdgcn_readfirstlane((uint32_t)c[ips]);
          ips++;
          lit += (int)b;
        } while (b == 255u);
      }
      if (lit > clen - ips || lit > olen - op) return -1;
      if (!emit_literals(ips, lit)) return -1;
      ips += lit;
      ip = ips;
      if (ips == clen) return 1;
      if (clen - ips < 2) return -1;
      const int off = (int)__builtin_amdgcn_readfirstlane((uint32_t)c[ips] | ((uint32_t)c[ips + 1] << 8));
      ips += 2;
      int ml = (int)(token & 15u);
      if (ml == 15) {
        uint32_t b;
        do {
          if (ips >= clen) return -1;
          b = __builtin_amdgcn_readfirstlane((uint32_t)c[ips]);
          ips++;
          ml += (int)b;
        } while (b == 255u);
      }
      ml += 4;
      ip = ips;
      if (!emit_match(off, ml)) return -1;
      return 0;
    };

#ifdef S3S_DEC_RING
    // ---- the stream ring: bytes [s_hi - 512, s_hi) of the stream (those below clen), `s_pre` = the dwords behind ----
    int s_hi = 0;
    uint32_t s_pre = 0;
    auto stream_dword = [&](int p) __attribute__((always_inline)) -> uint32_t {  // lane's dword at p + 4 * lane, clamped
      const int q = p + 4 * lane;
      uint32_t v = 0;
      if (q + 4 <= clen) {
        v = g_ld32(c + q);
      } else {
        for (int k = 0; k < 4; k++)
          if (q + k < clen) v |= (uint32_t)c[q + k] << (8 * k);
      }
      return v;
    };
    auto ring_ld32 = [&](int p) __attribute__((always_inline)) -> uint32_t {  // the stream's dword at p (aligned reads)
      const int r = p & (kSRing - 1);
      const uint8_t* q = sring + (r & ~3);
      return __builtin_amdgcn_alignbyte(*reinterpret_cast<const uint32_t*>(q + 4), *reinterpret_cast<const uint32_t*>(q), (uint32_t)r & 3u);
    };
    auto refill = [&]() __attribute__((always_inline)) {  // afterwards: s_hi >= ip + 256 or s_hi >= clen
      while (s_hi - ip < kSHalf && s_hi < clen) {
        if (ip >= s_hi) {  // first window / the byte-wise path ran ahead: start over where the parse is
          s_hi = ip & ~(kSHalf - 1);
          s_pre = stream_dword(s_hi);
        }
        const int ri = s_hi & (kSRing - 1);
        *reinterpret_cast<uint32_t*>(sring + ri + 4 * lane) = s_pre;
        if (ri == 0 && lane == 0) *reinterpret_cast<uint32_t*>(sring + kSRing) = s_pre;
        s_hi += kSHalf;
        if (s_hi < clen) s_pre = stream_dword(s_hi);
      }
    };

#endif
    // ---- main loop: parse windows of 64 stream bytes ---------------------------------------------------
    for (;;) {
#ifdef S3S_DEC_RING
      refill();
#endif
      bool eob = false;
      bool cx = true, is_lit = false;
      int nxt = 0;
      uint32_t r0 = 0, r1 = 0;
      uint64_t mask = 0;
      int rel = 0;
      bool parsed = false;
#if !defined(S3S_DEC_RING) && !defined(S3S_DEC_NO_PARSE_BLOCK)
      if constexpr (kFmt == kFmtLz4) {
        // ---- LZ4, interior windows: the hand-written parse block (round 6) ----------------------------------------------
        // Same arithmetic as the generic front end + walk + record store below, for windows whose 64 first dwords lie inside
        // the block (ip + 67 <= clen), in one asm statement that LOOPS over windows for as long as the plain case holds:
        // the chain starts with a real token (mask != 0) and the window's sequences still fit into the batch.  It leaves
        //   code 0: every window it took is stored (ip, nseq, sbase advanced) and the next one is not interior,
        //   code 1: the window at ip starts with a token of the byte-wise path (mask == 0),
        //   code 2: the window at ip is parsed (mask, rel, r0, r1) but the batch is full,
        // and the code below carries on from there (flush, byte-wise path, store) exactly as after its own front end.
        // Why by hand: the compiler's form of this loop is ~90 scalar instructions per window (flag registers for every
        // break / continue, re-materialised bools) and the CU's ONE scalar port is what its 32 decoder wavefronts queue for
        // (profiles/r06_experiments.md §2); this block has ~62, and 33 vector instructions instead of ~48.
        if (ip + 67 <= clen) {
          int code, n_, cnt_, t_;
          uint64_t cx_, sv_, t1_;
          uint32_t v_cpos, v_d0, v_lit, v_b1, v_ml, v_t, v_hdr, v_p2, v_d1, v_e, v_adv, v_t2, v_nrel;
          asm volatile(
              ".Lp_loop%=:\n\t"
              "v_add_u32_e32 %[cpos], %[ip], %[lane]\n\t"
              "global_load_dword %[d0], %[cpos], %[c]\n\t"
              "s_waitcnt vmcnt(0)\n\t"
              "v_bfe_u32 %[lit], %[d0], 4, 4\n\t"
              "v_bfe_u32 %[b1], %[d0], 8, 8\n\t"
              "v_and_b32_e32 %[ml], 15, %[d0]\n\t"
              "v_cmp_eq_u32_e32 vcc, 15, %[lit]\n\t"              // literal length continues in the next byte
              "v_add_u32_e32 %[vt], 15, %[b1]\n\t"
              "v_cmp_eq_u32_e64 %[cx], %[ff], %[b1]\n\t"          // ... and that byte is 255: byte-wise path
              "v_cndmask_b32_e32 %[lit], %[lit], %[vt], vcc\n\t"
              "v_cndmask_b32_e64 %[hdr], 1, 2, vcc\n\t"
              "s_and_b64 %[cx], %[cx], vcc\n\t"
              "v_add3_u32 %[p2], %[cpos], %[hdr], %[lit]\n\t"      // where offset + match-length byte sit
              "v_add_u32_e32 %[vt], 4, %[p2]\n\t"
              "v_cmp_ge_i32_e32 vcc, %[clen], %[vt]\n\t"           // p2 + 4 <= clen
              "s_orn2_b64 %[cx], %[cx], vcc\n\t"
              "s_and_saveexec_b64 %[sv], vcc\n\t"
              "global_load_dword %[d1], %[p2], %[c]\n\t"
              "s_mov_b64 exec, %[sv]\n\t"
              "v_cmp_eq_u32_e32 vcc, 15, %[ml]\n\t"               // match length continues in the byte behind the offset
              "v_add_u32_e32 %[t2], %[lane], %[hdr]\n\t"
              "s_waitcnt vmcnt(0)\n\t"
              "v_bfe_u32 %[e], %[d1], 16, 8\n\t"
              "v_and_b32_e32 %[d1], 0xffff, %[d1]\n\t"
              "v_cndmask_b32_e64 %[adv], 2, 3, vcc\n\t"
              "v_add_u32_e32 %[vt], %[ml], %[e]\n\t"
              "v_cmp_eq_u32_e64 %[t1], %[ff], %[e]\n\t"
              "v_cndmask_b32_e32 %[ml], %[ml], %[vt], vcc\n\t"
              "s_and_b64 %[t1], %[t1], vcc\n\t"
              "s_or_b64 %[cx], %[cx], %[t1]\n\t"
              "v_add_u32_e32 %[nrel], %[p2], %[adv]\n\t"
              "v_add_u32_e32 %[ml], 4, %[ml]\n\t"
              "v_subrev_u32_e32 %[nrel], %[ip], %[nrel]\n\t"        // position of the next token, window-relative
              "v_lshl_or_b32 %[r0], %[ml], 16, %[lit]\n\t"
              "v_lshl_or_b32 %[r1], %[t2], 16, %[d1]\n\t"
              "v_cndmask_b32_e64 %[nrel], %[nrel], -1, %[cx]\n\t"
              // the walk (five instructions per token, see the generic one below)
              "s_mov_b32 %[rel], 0\n\t"
              "s_mov_b64 %[m], 0\n\t"
              "v_readlane_b32 %[n], %[nrel], %[rel]\n\t"
              "s_cmp_lt_u32 %[n], 64\n\t"
              "s_cbranch_scc0 .Lp_wout%=\n\t"
              ".Lp_wnext%=:\n\t"
              "s_bitset1_b64 %[m], %[rel]\n\t"
              "s_mov_b32 %[rel], %[n]\n\t"
              "v_readlane_b32 %[n], %[nrel], %[rel]\n\t"
              "s_cmp_lt_u32 %[n], 64\n\t"
              "s_cbranch_scc1 .Lp_wnext%=\n\t"
              ".Lp_wout%=:\n\t"
              "s_cmp_gt_i32 %[n], -1\n\t"                           // the chain left the window: its last token is a real one
              "s_cbranch_scc0 .Lp_wdone%=\n\t"
              "s_bitset1_b64 %[m], %[rel]\n\t"
              "s_mov_b32 %[rel], %[n]\n\t"
              ".Lp_wdone%=:\n\t"
              "s_bcnt1_i32_b64 %[cnt], %[m]\n\t"                    // scc = (mask != 0)
              "s_mov_b32 %[code], 1\n\t"
              "s_cbranch_scc0 .Lp_exit%=\n\t"
              "s_add_i32 %[t], %[nseq], %[cnt]\n\t"
              "s_mov_b32 %[code], 2\n\t"
              "s_cmp_gt_i32 %[t], 64\n\t"
              "s_cbranch_scc1 .Lp_exit%=\n\t"
              // store the window's records: token lanes only, record index = nseq + tokens below the lane
              "s_cmp_eq_u32 %[nseq], 0\n\t"
              "s_cselect_b32 %[sbase], %[ip], %[sbase]\n\t"
              "s_sub_i32 %[cnt], %[ip], %[sbase]\n\t"
              "s_lshl_b32 %[cnt], %[cnt], 16\n\t"
              "s_mov_b64 exec, %[m]\n\t"
              "v_mbcnt_lo_u32_b32 %[vt], exec_lo, 0\n\t"
              "v_mbcnt_hi_u32_b32 %[vt], exec_hi, %[vt]\n\t"
              "v_add_u32_e32 %[r1], %[cnt], %[r1]\n\t"
              "v_add_lshl_u32 %[vt], %[vt], %[nseq], 3\n\t"
              "v_add_u32_e32 %[vt], %[rec], %[vt]\n\t"
              "ds_write_b32 %[vt], %[r0]\n\t"
              "ds_write_b32 %[vt], %[r1] offset:4\n\t"
              "s_mov_b64 exec, -1\n\t"
              "s_mov_b32 %[nseq], %[t]\n\t"
              "s_add_i32 %[ip], %[ip], %[rel]\n\t"
              "s_add_i32 %[t], %[ip], 67\n\t"
              "s_mov_b32 %[code], 0\n\t"
              "s_cmp_le_i32 %[t], %[clen]\n\t"
              "s_cbranch_scc1 .Lp_loop%=\n\t"
              ".Lp_exit%=:"
              : [code] "=&s"(code), [m] "=&s"(mask), [rel] "=&s"(rel), [r0] "=&v"(r0), [r1] "=&v"(r1), [ip] "+s"(ip),
                [nseq] "+s"(nseq), [sbase] "+s"(sbase), [n] "=&s"(n_), [cnt] "=&s"(cnt_), [t] "=&s"(t_), [cx] "=&s"(cx_),
                [sv] "=&s"(sv_), [t1] "=&s"(t1_), [cpos] "=&v"(v_cpos), [d0] "=&v"(v_d0), [lit] "=&v"(v_lit), [b1] "=&v"(v_b1),
                [ml] "=&v"(v_ml), [vt] "=&v"(v_t), [hdr] "=&v"(v_hdr), [p2] "=&v"(v_p2), [d1] "=&v"(v_d1), [e] "=&v"(v_e),
                [adv] "=&v"(v_adv), [t2] "=&v"(v_t2), [nrel] "=&v"(v_nrel)
              : [c] "s"(c), [clen] "s"(clen), [lane] "v"(lane), [rec] "s"(lds_addr(rec)), [ff] "s"(255)
              : "vcc", "scc", "memory");
          if (code == 0) continue;  // (the windows it took are stored; the next one is near the block's end: generic path)
          parsed = true;
        }
      }
      if constexpr (kFmt != kFmtLz4) {
        // ---- Snappy / LZF, interior windows: the hand-written parse block (round 6, second half) -------------------------------
        // The LZ4 block's recipe for Snappy's elements: tag decode without a branch (one dword per lane: literal with its
        // length in the tag or in 1 - 2 bytes behind it, copy with a 1- or 2-byte offset; anything else = byte-wise path),
        // the walk — pointer doubling when the previous window held 12 tokens or more, the five-instruction scalar loop
        // otherwise —, which copy joins the literal in front of it (a literal token marks its successor through the window's
        // pad in LDS; the window's first token looks at open_lit), and the record stores.  Leaves like the LZ4 block:
        //   code 0  windows stored, the next one is not interior;  code 1  mask == 0;  code 2  batch full
        // (for 1 / 2 the code below redoes the join from mask / is_lit / open_lit, as after its own front end).
        // Compiled form of the same: ~306 instructions per window on wide rows (104 tag decode with a branch per element type,
        // 100 walk, 102 join + store + flags); this block: ~165.
        if (ip + 67 <= clen) {
          int code, n_, cnt_, t_, lt_ = __builtin_amdgcn_readfirstlane(last_tokens), open_ = __builtin_amdgcn_readfirstlane(open_lit ? 1 : 0);
          uint64_t cx_, sv_, isl_, t1_, lm_, S_, J_, rch_;
          uint32_t v_cpos, v_d0, v_ty, v_n6, v_b1, v_w16, v_len, v_hdr, v_t, v_t2, v_nxt, v_nrel, v_jmp, v_m, v_isl, n_one, n_c64, n_lma;
          const uint32_t markb = lds_addr(win + kBWin);
          if constexpr (kFmt == kFmtSnappy) {
            S3S_EL_PARSE_BLOCK(S3S_EL_FRONT_SNAPPY)
          } else {
            S3S_EL_PARSE_BLOCK(S3S_EL_FRONT_LZF)
          }
          last_tokens = lt_;
          open_lit = open_ != 0;
          if (code == 0) continue;
          is_lit = v_isl != 0u;
          cx = false;
          parsed = true;
        }
      }
#endif
      if (!parsed) {
      if (ip >= clen) {
        // LZ4 blocks end inside the byte-wise path (last sequence: literals only); Snappy blocks end here
        if (kFmt == kFmtLz4 || ip > clen) { bad = true; break; }
        eob = true;
      } else {
        const int cpos = ip + lane;
        if (cpos + 4 <= clen) {
#ifdef S3S_DEC_RING
          const uint32_t d0 = ring_ld32(cpos);
#else
          const uint32_t d0 = g_ld32(c + cpos);
#endif
          if constexpr (kFmt == kFmtLzf) {
            const uint32_t ctrl = d0 & 0xffu;
            if (ctrl < 32u) {  // literal run (1 .. 32 bytes)
              const int len = (int)ctrl + 1;
              nxt = cpos + 1 + len;
              cx = nxt > clen;
              is_lit = true;
              r0 = (uint32_t)len;
              r1 = (uint32_t)(lane + 1) << 16;
            } else {           // back reference: 2 or 3 bytes
              const uint32_t l3 = ctrl >> 5, b1 = (d0 >> 8) & 0xffu, b2 = (d0 >> 16) & 0xffu;
              const bool ext = l3 == 7u;
              nxt = cpos + (ext ? 3 : 2);
              cx = false;
              r0 = ((ext ? 7u + b1 : l3) + 2u) << 16;
              r1 = (((ctrl & 0x1fu) << 8) | (ext ? b2 : b1)) + 1u;
            }
          } else if constexpr (kFmt == kFmtSnappy) {
            const uint32_t tag = d0 & 0xffu, ty = tag & 3u, n6 = tag >> 2;
            if (ty == 0u) {
              int len = (int)n6 + 1, hdr = 1;
              if (n6 == 60u) {
                len = (int)((d0 >> 8) & 0xffu) + 1;
                hdr = 2;
              } else if (n6 == 61u) {
                len = (int)((d0 >> 8) & 0xffffu) + 1;
                hdr = 3;
              }
              nxt = cpos + hdr + len;
              cx = n6 > 61u || nxt > clen || len > kMaxBlock;
              is_lit = true;
              r0 = (uint32_t)len;
              r1 = (uint32_t)(lane + hdr) << 16;  // (window-relative; the batch's base is added when the record is stored)
            } else if (ty == 1u) {
              nxt = cpos + 2;
              cx = false;
              r0 = (4u + (n6 & 7u)) << 16;
              r1 = ((tag >> 5) << 8) | ((d0 >> 8) & 0xffu);
            } else if (ty == 2u) {
              nxt = cpos + 3;
              cx = false;
              r0 = (n6 + 1u) << 16;
              r1 = (d0 >> 8) & 0xffffu;
            }
          } else {
            const uint32_t tok = d0 & 0xffu, b1 = (d0 >> 8) & 0xffu;
            int lit = (int)(tok >> 4), hdr = 1;
            bool complex_ = false;
            if (lit == 15) {
              lit += (int)b1;
              hdr = 2;
              complex_ = b1 == 255u;
            }
            const int p2 = cpos + hdr + lit;
            if (p2 + 4 <= clen) {
#ifdef S3S_DEC_RING
              const uint32_t d1 = p2 + 4 <= s_hi ? ring_ld32(p2) : g_ld32(c + p2);
#else
              const uint32_t d1 = g_ld32(c + p2);
#endif
              int ml = (int)(tok & 15u), adv = 2;
              if (ml == 15) {
                const uint32_t e = (d1 >> 16) & 0xffu;
                ml += (int)e;
                adv = 3;
                complex_ = complex_ || e == 255u;
              }
              cx = complex_;
              nxt = p2 + adv;
              r0 = (uint32_t)lit | ((uint32_t)(ml + 4) << 16);
              r1 = (d1 & 0xffffu) | ((uint32_t)(lane + hdr) << 16);
            }
          }
        }
      }
      // scalar walk over the chain of real tokens: lane `rel` holds the window-relative position of the
      // token behind it, or -1 if the byte-wise path has to take it
      const int nrel = cx ? -1 : nxt - ip;
      if (last_tokens >= (kFmt != kFmtLz4 ? kSnappyWalkTokens : kParallelWalkTokens)) {
        // many short tokens (Snappy elements are 2-3 bytes long on match-dense data, 25 and more per window): the chain
        // is followed by pointer doubling — six rounds of "lanes on the chain mark the lane 2^k tokens behind them"
        // through 64 bytes of LDS (the window's pad), whatever the number of tokens.
        // The marks are how LANES talk to each other (lane i marks lane jmp[i]): to the compiler a plain store / load pair of one
        // thread — it forwards "my own mark is still 0" into lanes that did not store (seen in round 6 when the read stopped being
        // conditional; the interpreter's wait-count check caught it).  So the LDS accesses are asm: opaque, in program order.
        const uint32_t mark = lds_addr(win + kBWin);  // (the window's pad: only ever over-read otherwise)
        int jmp = cx ? kWave : (nrel < kWave ? nrel : kWave);
        uint64_t RCH = 1ull;  // lanes known to be on the chain
        asm volatile("ds_write_b8 %0, %1" : : "v"(mark + (uint32_t)lane), "v"(0) : "memory");
#pragma unroll
        for (int k = 0; k < 6; k++) {
          const bool put = ((RCH >> lane) & 1ull) != 0ull && jmp < kWave;
          if (put) asm volatile("ds_write_b8 %0, %1" : : "v"(mark + (uint32_t)jmp), "v"(1) : "memory");
          uint32_t m;
          asm volatile("ds_read_u8 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(m) : "v"(mark + (uint32_t)lane) : "memory");
          RCH |= ballot64(m != 0u);
          const int j2 = __shfl(jmp, jmp & 63);
          jmp = jmp < kWave ? j2 : kWave;
        }
        const bool reach = ((RCH >> lane) & 1ull) != 0ull;
        const uint64_t RM = ballot64(reach);
        const uint64_t CXR = ballot64(reach && cx);  // the (at most one) complex token the chain runs into
        mask = RM & ~CXR;
        if (CXR) rel = __builtin_ctzll(CXR);
        else rel = __builtin_amdgcn_readlane(nrel, 63 - __builtin_clzll(mask));  // (lane 0 is on the chain: mask != 0)
      } else {
        // for (;;) { n = nrel[rel]; if (n < 0) break; mask |= 1 << rel; rel = n; if (rel >= 64) break; } — five
        // instructions per token, both ends of the chain in one unsigned compare (the compiler's loop has thirteen,
        // and the CU's one scalar issue per cycle is what its 32 decoder waves wait for)
        int n;
        asm volatile(
            "v_readlane_b32 %[n], %[v], %[rel]\n"
            "s_cmp_lt_u32 %[n], 64\n"
            "s_cbranch_scc0 .Lwalk_out%=\n"
            ".Lwalk_next%=:\n"
            "s_bitset1_b64 %[m], %[rel]\n"
            "s_mov_b32 %[rel], %[n]\n"
            "v_readlane_b32 %[n], %[v], %[rel]\n"
            "s_cmp_lt_u32 %[n], 64\n"
            "s_cbranch_scc1 .Lwalk_next%=\n"
            ".Lwalk_out%=:\n"
            : [m] "+s"(mask), [rel] "+s"(rel), [n] "=&s"(n)
            : [v] "v"(nrel)
            : "scc");
        if (n >= 0) {  // the chain left the window: its last token is a real one
          mask |= 1ull << rel;
          rel = n;
        }
      }
      }  // (!parsed)
      last_tokens = __builtin_popcountll(mask);
      const int cur = ip + rel;
      // Snappy: a copy element that directly follows a literal element shares the literal's record (literal run +
      // match = one sequence, as in LZ4): half as many records, batches and round lanes.  S = tokens that start a
      // record, J = copy tokens that join the record in front of them (`open`: the batch's last record is a
      // literal whose successor is this window's first token).
      uint64_t S = mask, J = 0ull;
      bool joined = false;
      if constexpr (kFmt != kFmtLz4) {
        const uint64_t LM = ballot64(is_lit) & mask;
        const uint64_t below = mask & ((1ull << lane) - 1ull);
        const bool prev_lit = below ? ((LM >> (63 - __builtin_clzll(below))) & 1ull) != 0ull : open_lit;
        joined = ((mask >> lane) & 1ull) != 0ull && !is_lit && prev_lit;
        J = ballot64(joined);
        S = mask & ~J;
      }
      int cnt = __builtin_popcountll(S);
      // the batch is flushed when this window's sequences do not fit any more, and in front of every
      // sequence of the byte-wise path (so in particular in front of the block's last sequence)
      if (mask == 0ull || nseq + cnt > kWave) {
        if (!flush_batch()) { bad = true; break; }
        if constexpr (kFmt != kFmtLz4) {
          if (open_lit && (J & mask & (0ull - mask)) != 0ull) {  // the first token cannot join a flushed record any more
            const uint64_t first = mask & (0ull - mask);
            J &= ~first;
            S |= first;
            joined = joined && ((first >> lane) & 1ull) == 0ull;
            cnt = __builtin_popcountll(S);
          }
          open_lit = false;
        }
      }
      if (eob) break;
      if (mask == 0ull) {
        BT_MARK(0);
        const int r = slow_sequence();
        BT_MARK(7);
        if (r < 0) { bad = true; break; }
        if (r == 1) break;
        continue;
      }
      if (nseq == 0) sbase = ip;  // (the batch in front of this window's records was flushed, or there was none)
      if ((mask >> lane) & 1ull) {
        const int t = nseq + __builtin_amdgcn_mbcnt_hi((uint32_t)(S >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)S, 0u));
        if (!joined) rec[t] = make_uint2(r0, r1 + ((uint32_t)(ip - sbase) << 16));  // stream offsets count from the batch's first token
      }
      if constexpr (kFmt != kFmtLz4) {
        if (joined) {  // (after the record's first half has been stored by the literal's lane)
          const int t = nseq + __builtin_amdgcn_mbcnt_hi((uint32_t)(S >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)S, 0u)) - 1;
          uint16_t* h16 = reinterpret_cast<uint16_t*>(rec + t);
          h16[1] = (uint16_t)(r0 >> 16);  // match length
          h16[2] = (uint16_t)r1;          // offset
        }
        const uint64_t LM = ballot64(is_lit) & mask;
        open_lit = ((LM >> (63 - __builtin_clzll(mask))) & 1ull) != 0ull;  // the window's last token is a literal
      }
      nseq += cnt;
      ip = cur;
    }
    if (!bad && nseq != 0) bad = true;  // (a block always ends in the byte-wise path, behind a flush)
    if (!bad && op != olen) bad = true;
    BT_MARK(0);
    if (!bad) flush_to(op);
    BT_MARK(6);
    BT_DONE();
  }
  if (bad) {
    if (lane == 0) atomicExch(status, S3S_E_BAD_FRAME);
    return;
  }
}


// ---- frame checks: xxHash32 of every decoded block, four lanes per frame ------------------------------
// xxHash32 has four accumulators, each a serial multiply chain over every fourth dword — a wavefront per
// frame keeps four lanes busy and pays two quarter-rate 32-bit multiplies per 16 bytes on all 64.  Here a
// wavefront checks 16 frames at once (lane = 4 * frame-in-wave + accumulator); the blocks were written
// moments ago by the decode kernel on the same stream, so they come out of L2 / MALL.
__global__ __launch_bounds__(kWave) void lz4_verify_frames_kernel(
    const Frame* __restrict__ frames, int32_t n_frames, const int64_t* __restrict__ frame_out,
    const uint8_t* __restrict__ dst, int32_t* __restrict__ status) {
  const int f = blockIdx.x * (kWave / 4) + (threadIdx.x >> 2);
  const int l = threadIdx.x & 3;
  int len = 0;
  uint32_t want = 0;
  const uint8_t* g = dst;
  if (f < n_frames) {
    const Frame fr = frames[f];
    len = fr.orig_len;
    want = fr.check;
    g = dst + frame_out[f];
    // a compressed frame above kBatchMaxBlock was NOT decoded by batch_decode_kernel (status = S3S_E_UNSUPPORTED, the
    // caller retries with the ring decoder, which checks the hash itself): hashing the stale destination here
    // would replace that status with S3S_E_BAD_FRAME and suppress the retry
    if (len > kBatchMaxBlock && fr.method != 0x10) len = 0;
  }
  const uint32_t seed = kLz4BlockSeed;
  uint32_t acc = l == 0 ? seed + XP1 + XP2 : l == 1 ? seed + XP2 : l == 2 ? seed : seed - XP1;
  const int stripes = len >> 4;
  const uint8_t* q = g + 4 * l;
  // 24 stripes per iteration in three register sets, each requested two multiply chains (16 stripes) before
  // it is consumed; loads past the frame's last stripe are clamped and their values skipped
  if (stripes > 0) {
    const int last = stripes - 1;
    uint32_t s0[8], s1[8], s2[8];
    auto fetch = [&](uint32_t (&w)[8], int j0) __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int jj = j0 + u;
        w[u] = g_ld32(q + 16 * (jj < last ? jj : last));
      }
    };
    auto fold = [&](const uint32_t (&w)[8], int j0) __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const uint32_t nacc = rotl32(acc + w[u] * XP2, 13) * XP1;
        acc = j0 + u < stripes ? nacc : acc;
      }
    };
    fetch(s0, 0);
    fetch(s1, 8);
    for (int j = 0; j < stripes; j += 24) {
      fetch(s2, j + 16);
      fold(s0, j);
      fetch(s0, j + 24);
      fold(s1, j + 8);
      fetch(s1, j + 32);
      fold(s2, j + 16);
    }
  }
  const int g0 = (threadIdx.x & 63) & ~3;
  const uint32_t v1 = __shfl(acc, g0), v2 = __shfl(acc, g0 + 1), v3 = __shfl(acc, g0 + 2), v4 = __shfl(acc, g0 + 3);
  if (len == 0 || l != 0) return;
  uint32_t h = len >= 16 ? rotl32(v1, 1) + rotl32(v2, 7) + rotl32(v3, 12) + rotl32(v4, 18) : seed + XP5;
  h += (uint32_t)len;
  int p = stripes << 4;
  for (; p + 4 <= len; p += 4) h = rotl32(h + g_ld32(g + p) * XP3, 17) * XP4;
  for (; p < len; p++) h = rotl32(h + (uint32_t)g[p] * XP5, 11) * XP1;
  h ^= h >> 15;
  h *= XP2;
  h ^= h >> 13;
  h *= XP3;
  h ^= h >> 16;
  if ((h & 0x0FFFFFFFu) != want) atomicExch(status, S3S_E_BAD_FRAME);
}

}  // namespace

void launch_lz4_decompress_batch(const uint8_t* d_comp, const Frame* d_frames, int32_t n_frames,
                                 const int64_t* d_frame_out, uint8_t* d_dst, int32_t* d_status,
                                 hipStream_t st, hipEvent_t after_decode) {
  if (n_frames <= 0) {
    if (after_decode) (void)hipEventRecord(after_decode, st);
    return;
  }
  hipLaunchKernelGGL(batch_decode_kernel<kFmtLz4>, dim3((unsigned)n_frames), dim3(kWave), 0, st, d_comp, d_frames,
                     n_frames, d_frame_out, d_dst, d_status);
  if (after_decode) (void)hipEventRecord(after_decode, st);
  hipLaunchKernelGGL(lz4_verify_frames_kernel, dim3((unsigned)((n_frames + kWave / 4 - 1) / (kWave / 4))), dim3(kWave),
                     0, st, d_frames, n_frames, d_frame_out, d_dst, d_status);
}

void launch_lzf_decompress_batch(const uint8_t* d_comp, const Frame* d_frames, int32_t n_frames, const int64_t* d_frame_out,
                                 uint8_t* d_dst, int32_t* d_status, hipStream_t st) {
  if (n_frames <= 0) return;
  hipLaunchKernelGGL(batch_decode_kernel<kFmtLzf>, dim3((unsigned)n_frames), dim3(kWave), 0, st, d_comp, d_frames, n_frames,
                     d_frame_out, d_dst, d_status);
}

void launch_snappy_decompress_batch(const uint8_t* d_comp, const Frame* d_frames, int32_t n_frames,
                                    const int64_t* d_frame_out, uint8_t* d_dst, int32_t* d_status,
                                    hipStream_t st) {
  if (n_frames <= 0) return;
  hipLaunchKernelGGL(batch_decode_kernel<kFmtSnappy>, dim3((unsigned)n_frames), dim3(kWave), 0, st, d_comp, d_frames,
                     n_frames, d_frame_out, d_dst, d_status);
}

}  // namespace s3s
