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
  __device__ __forceinline__ uint32_t rd32(int pos) const {
    uint32_t v;
    __builtin_memcpy(&v, base + pos, 4);  // unaligned global_load_dword
    return v;
  }
  __device__ __forceinline__ uint32_t rd8(int pos) const { return base[pos]; }
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

typedef __attribute__((address_space(3))) uint16_t lds_u16;

// The parse.  Returns the compressed size, or -1 if it would exceed len.
template <typename Src>
__device__ int lz4_compress_wave(const Src in, lds_u16* table, int len, uint8_t* out, int lane) {
  volatile lds_u16* T = table;  // every access is a real ds_read_u16 / ds_write_b16, in order
  const int mfl1 = len - kMfLimit + 1;  // mflimitPlusOne
  const int matchlimit = len - kLastLiterals;
  const int last4 = len - 4;
  int anchor = 0, op = 0;

  if (len >= kMfLimit + 1) {
    T[hash13(in.rd32(0))] = 0;  // LZ4_putPosition(ip = source); all lanes store the same value
    int base = 1, t0 = 1;
    uint32_t vpre = in.rd32(1 + lane < last4 ? 1 + lane : last4);  // prefetched v of the next batch
    bool have_pre = true;
    uint32_t vput = 0;  // the 4 bytes at base-2, to insert before the batch (after a match)
    bool put_pending = false;
    for (;;) {
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
      int ip = ip0, match = match0;
      // both extensions read independent bytes: issue them together
      // backward: catch-up over pending literals, 64 bytes per round
      int maxback = ip - anchor < match ? ip - anchor : match;
      uint32_t ba = 0, bb = 1;
      if (lane < maxback) {
        ba = in.rd8(ip - 1 - lane);
        bb = in.rd8(match - 1 - lane);
      }
      // forward count from the 4 matched bytes, 256 bytes per round (LZ4_count to matchlimit);
      // dwords starting at or beyond matchlimit never count, so clamping their address is free
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
      const int ipe = ip0 + kMinMatch + fwd;  // first byte after the match
      // prefetch what the next batch needs while the sequence is being written out
      if (ipe < mfl1) {
        vpre = in.rd32(ipe + lane < last4 ? ipe + lane : last4);
        vput = in.rd32(ipe - 2);
        have_pre = true;
        put_pending = true;
      }
      const int mcode = (ip0 - ip) + fwd;  // bytes beyond MINMATCH, counted from the moved-back ip
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
  const int clen = lz4_compress_wave(SrcLds{s.in}, (lds_u16*)s.table, len, slot + kSlotHeader, lane);
  finish_frame(slot, len, clen, item_check[it], item.kind >> 8, item_size + it, lane);
}

// ---- variant B: chunk read in place, only the table in LDS ----------------------------------
__global__ __launch_bounds__(kWave) void lz4_compress_l2_kernel(
    const uint8_t* __restrict__ src, const Item* __restrict__ items, int32_t n_items,
    const uint32_t* __restrict__ item_check, uint8_t* __restrict__ slots,
    uint32_t* __restrict__ item_size) {
  __shared__ __attribute__((aligned(16))) uint16_t table[8192];
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
  const int clen = lz4_compress_wave(SrcGlobal{src + item.src_off}, (lds_u16*)table, item.len,
                                     slot + kSlotHeader, lane);
  finish_frame(slot, item.len, clen, item_check[it], item.kind >> 8, item_size + it, lane);
}

// ---- xxHash32 of every chunk: 4 lanes per chunk (one per stripe accumulator) ----------------
constexpr uint32_t XXP1 = 2654435761u, XXP2 = 2246822519u, XXP3 = 3266489917u,
                   XXP4 = 668265263u, XXP5 = 374761393u;
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

}  // namespace

void launch_lz4_compress(const uint8_t* d_src, const Item* d_items, int32_t n_items,
                         uint32_t* d_item_check, uint8_t* d_slots, uint32_t* d_item_size,
                         int variant, hipStream_t st, hipEvent_t after_hash) {
  if (n_items <= 0) {
    if (after_hash) hipEventRecord(after_hash, st);
    return;
  }
  hipLaunchKernelGGL(xxh32_items_kernel, dim3((unsigned)((n_items + kXxhThreads / 4 - 1) / (kXxhThreads / 4))),
                     dim3(kXxhThreads), 0, st, d_src, d_items, n_items, kLz4BlockSeed, d_item_check);
  if (after_hash) hipEventRecord(after_hash, st);
  if (variant == 0)
    hipLaunchKernelGGL(lz4_compress_lds_kernel, dim3((unsigned)n_items), dim3(kWave), 0, st, d_src,
                       d_items, n_items, d_item_check, d_slots, d_item_size);
  else
    hipLaunchKernelGGL(lz4_compress_l2_kernel, dim3((unsigned)n_items), dim3(kWave), 0, st, d_src,
                       d_items, n_items, d_item_check, d_slots, d_item_size);
}

}  // namespace s3s
