// lz4_compress.hip — bit-exact LZ4 block compression of 32 KiB shuffle chunks on CDNA4.
//
// Replaces the [EXT] LZ4BlockOutputStream.flushBufferedData() stage (lz4-java 1.8.0 ->
// liblz4 1.9.3 LZ4_compress_default + xxHash32) that produces the bytes arriving at
// S3ShuffleMapOutputWriter.scala:182-188 in the reference.  Output must equal the JVM path
// byte for byte, so this is NOT a "GPU-friendly LZ4 variant": it reproduces the greedy
// single-pass parse of LZ4_compress_generic(byU16, acceleration 1) exactly, including its
// hash-table update order, skip schedule, backward catch-up and end-of-block rules.
//
// Mapping onto the machine (one workgroup = 2 wavefronts per chunk, 3 workgroups per CU):
//   LDS   chunk bytes (32 KiB + slack) + the 8192 x u16 hash table (16 KiB)  = 48.3 KiB
//   wave0 runs the parse.  The sequential probe loop of the CPU code is evaluated 64
//         positions at a time: lane i takes the i-th position of the deterministic
//         "no match yet" schedule, hashes it, reads the table, and ALL lanes insert
//         speculatively with one ds_write.  A readback tells every lane whether it lost a
//         same-slot race (=> some lane in the batch shares its hash); the first such lane
//         bounds the prefix in which the pre-batch table entries are the true candidates
//         ("cut").  The first lane below the cut whose candidate matches wins
//         (ballot + ctz); lanes after it undo their inserts, so the table state is exactly
//         the sequential one.  Match extension is cooperative: 256 B forward / 64 B backward
//         per LDS round trip.  Which lane wins a same-address LDS store is irrelevant to the
//         result (tests/model/lz4_wave_model.cpp proves it under adversarial orders).
//   wave1 computes the frame's xxHash32 from the same LDS copy (4 lanes, one per lane
//         accumulator) in the shadow of the parse, so the chunk is read from HBM once.
//   out   token/literal/offset bytes go straight to the chunk's slot in HBM; compressed
//         output never exceeds the chunk length (anything longer is stored RAW by the frame
//         rule compressedLength >= originalLength), so a slot is 32 B + 32 KiB.
#include "s3s_internal.h"

namespace s3s {
namespace {

constexpr int kThreads = 128;
constexpr int kLdsSlack = 320;  // cooperative compares over-read at most 4*63+3+7 bytes
constexpr int kMfLimit = 12, kLastLiterals = 5, kMinMatch = 4;

struct __attribute__((aligned(16))) Lz4Lds {
  uint8_t in[kMaxBlock + kLdsSlack];
  uint16_t table[8192];
  uint32_t xxh;
};

__device__ __forceinline__ uint32_t hash13(uint32_t v) { return (v * 2654435761u) >> 19; }

// Cumulative LZ4 skip schedule.  Probe index t (t = 0 is the "test next position" probe right
// after a match, t >= 1 the search loop with searchMatchNb starting at 64) sits S(t) bytes
// after the run's first probe:  step_0 = step_1 = 1, step_t = (62 + t) >> 6 for t >= 2.
__device__ __forceinline__ int sched_S(int t) {
  const int X = 62 + t;
  const int q = X >> 6, r = X & 63;
  return t < 2 ? t : 2 + 32 * q * (q - 1) + q * r;
}

__device__ __forceinline__ int ext_len(int v) { return v >= 15 ? (v - 15) / 255 + 1 : 0; }

// n bytes LDS -> global, dword-vectorised on the destination alignment.
__device__ __forceinline__ void copy_lds_to_global(uint8_t* dst, const uint8_t* in, int src_pos,
                                                   int n, int lane) {
  int head = (int)((4u - (uint32_t)(uintptr_t)dst) & 3u);
  head = head < n ? head : n;
  if (lane < head) dst[lane] = in[src_pos + lane];
  const int body = (n - head) >> 2;
  uint32_t* d32 = reinterpret_cast<uint32_t*>(dst + head);
  for (int j = lane; j < body; j += kWave) d32[j] = lds_rd32(in, src_pos + head + 4 * j);
  const int done = head + 4 * body;
  if (lane < n - done) dst[done + lane] = in[src_pos + done + lane];
}

// Writes one LZ4 sequence (token, literal-length bytes, literals and — if has_match — offset
// and match-length bytes) at out+op.  All scalar arguments are wave-uniform.  Returns the new
// op, or -1 when the sequence would not fit in cap bytes (=> the frame is stored RAW).
__device__ __forceinline__ int emit_sequence(uint8_t* out, int cap, int op, const uint8_t* in,
                                             int anchor, int lit, bool has_match, int offset,
                                             int mcode, int lane) {
  const int le = ext_len(lit);
  const int me = has_match ? ext_len(mcode) : 0;
  const int total = 1 + le + lit + (has_match ? 2 + me : 0);
  if (op + total > cap) return -1;
  const uint32_t token =
      (uint32_t)((lit < 15 ? lit : 15) << 4) | (uint32_t)(has_match ? (mcode < 15 ? mcode : 15) : 0);
  const int lit0 = 1 + le;        // first literal byte
  const int off0 = lit0 + lit;    // offset low byte
  const uint32_t lrem = (uint32_t)((lit - 15) % 255), mrem = (uint32_t)((mcode - 15) % 255);
  uint8_t* o = out + op;
  if (total <= kWave) {  // common case: one byte per lane, one store instruction
    const int k = lane;
    uint32_t b;
    if (k == 0) {
      b = token;
    } else if (k < lit0) {
      b = (k < le) ? 255u : lrem;
    } else if (k < off0) {
      b = in[anchor + (k - lit0)];
    } else if (k == off0) {
      b = (uint32_t)offset & 0xffu;
    } else if (k == off0 + 1) {
      b = (uint32_t)offset >> 8;
    } else {
      b = (k - (off0 + 2) < me - 1) ? 255u : mrem;
    }
    if (k < total) o[k] = (uint8_t)b;
    return op + total;
  }
  if (lane == 0) o[0] = (uint8_t)token;
  for (int j = lane; j < le; j += kWave) o[1 + j] = (uint8_t)(j < le - 1 ? 255u : lrem);
  copy_lds_to_global(o + lit0, in, anchor, lit, lane);
  if (has_match) {
    if (lane == 0) o[off0] = (uint8_t)offset;
    if (lane == 1) o[off0 + 1] = (uint8_t)((uint32_t)offset >> 8);
    for (int j = lane; j < me; j += kWave) o[off0 + 2 + j] = (uint8_t)(j < me - 1 ? 255u : mrem);
  }
  return op + total;
}

// The parse.  Returns the compressed size, or -1 if it would exceed len.
__device__ int lz4_compress_wave(const uint8_t* in, volatile uint16_t* T, int len, uint8_t* out,
                                 int lane) {
  const int mfl1 = len - kMfLimit + 1;         // mflimitPlusOne
  const int matchlimit = len - kLastLiterals;
  int anchor = 0, op = 0;

  if (len >= kMfLimit + 1) {
    if (lane == 0) T[hash13(lds_rd32(in, 0))] = 0;  // LZ4_putPosition(ip = source)
    int base = 1, t0 = 1;
    for (;;) {
      // ---- one batch: lane i evaluates probe t0+i of the current no-match run ----------------
      const int S0 = sched_S(t0);
      const int t = t0 + lane;
      const int pos = base + sched_S(t) - S0;
      const int nextpos = base + sched_S(t + 1) - S0;
      // the CPU loop leaves for _last_literals BEFORE probing pos when nextpos > mflimitPlusOne;
      // the post-match probe (t == 0) has no such test
      const bool valid = (t == 0) || (nextpos <= mfl1);
      const int nvalid = __popcll(__ballot(valid));  // valid lanes form a prefix
      uint32_t v = 0, h = 0, c = 0, r = 0;
      if (valid) {
        v = lds_rd32(in, pos);
        h = hash13(v);
        c = T[h];              // candidate as of the start of the batch
        T[h] = (uint16_t)pos;  // speculative insert, all lanes at once
        r = T[h];              // readback: did this lane own its slot?
      }
      const uint32_t w = lds_rd32(in, (int)c);
      const uint64_t C = __ballot(valid && r != (uint32_t)pos);
      const uint64_t M = __ballot(valid && w == v);
      // lanes below `cut` have pairwise distinct hashes => their start-of-batch candidates are
      // exactly what the sequential code would have read
      int cut = kWave;
      if (C) cut = (C & 1) ? 1 : __builtin_ctzll(C);
      const int lim = cut < nvalid ? cut : nvalid;
      const uint64_t Mv = lim >= kWave ? M : (M & ((1ull << lim) - 1ull));
      int m = -1, keep = lim;
      if (Mv) {
        m = __builtin_ctzll(Mv);
        keep = m + 1;
      }
      // undo inserts of lanes the sequential code never reached
      if (valid && lane >= keep && r == (uint32_t)pos) T[h] = (uint16_t)c;
      if ((C & 1) && lane == 0) T[h] = (uint16_t)pos;  // lane 0 lost its race but is committed

      if (m < 0) {
        if (lim == nvalid && nvalid < kWave) break;  // ran into mflimit: last literals
        base += sched_S(t0 + lim) - S0;              // continue the run at lane `lim`
        t0 += lim;
        continue;
      }

      // ---- match at lane m --------------------------------------------------------------------
      const int ip0 = base + sched_S(t0 + m) - S0;
      const int match0 = (int)__builtin_amdgcn_readlane(c, m);
      int ip = ip0, match = match0;
      // catch-up over pending literals, 64 bytes per round
      {
        int maxback = ip - anchor < match ? ip - anchor : match;
        while (maxback > 0) {
          const bool act = lane < maxback;
          uint32_t a = 0, b = 1;
          if (act) {
            a = in[ip - 1 - lane];
            b = in[match - 1 - lane];
          }
          const uint64_t E = __ballot(act && a == b);
          const int nbk = (~E == 0ull) ? kWave : __builtin_ctzll(~E);
          ip -= nbk;
          match -= nbk;
          if (nbk < kWave) break;
          maxback -= kWave;
        }
      }
      // forward count from the 4 matched bytes, 256 bytes per round (LZ4_count to matchlimit)
      int fwd = 0;
      for (;;) {
        const int avail = matchlimit - (ip0 + kMinMatch + fwd);
        if (avail <= 0) break;
        const uint32_t x = lds_rd32(in, ip0 + kMinMatch + fwd + 4 * lane) ^
                           lds_rd32(in, match0 + kMinMatch + fwd + 4 * lane);
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
      }
      const int mcode = (ip0 - ip) + fwd;  // bytes beyond MINMATCH, counted from the moved-back ip
      op = emit_sequence(out, len, op, in, anchor, ip - anchor, true, ip - match, mcode, lane);
      if (op < 0) return -1;
      ip = ip0 + kMinMatch + fwd;
      anchor = ip;
      if (ip >= mfl1) break;  // end of chunk
      if (lane == 0) T[hash13(lds_rd32(in, ip - 2))] = (uint16_t)(ip - 2);  // fill table
      base = ip;  // next batch starts with the "test next position" probe (t = 0)
      t0 = 0;
    }
  }
  return emit_sequence(out, len, op, in, anchor, len - anchor, false, 0, 0, lane);
}

constexpr uint32_t XXP1 = 2654435761u, XXP2 = 2246822519u, XXP3 = 3266489917u,
                   XXP4 = 668265263u, XXP5 = 374761393u;

// xxHash32 of in[0,len) (16-byte aligned LDS), lanes 0..3 carry the four stripe accumulators.
__device__ uint32_t xxh32_wave(const uint8_t* in, int len, uint32_t seed, int lane) {
  uint32_t h;
  int p = 0;
  if (len >= 16) {
    uint32_t acc = lane == 0 ? seed + XXP1 + XXP2 : lane == 1 ? seed + XXP2 : lane == 2 ? seed : seed - XXP1;
    const int stripes = len >> 4;
    if (lane < 4) {
      const uint32_t* q = reinterpret_cast<const uint32_t*>(in) + lane;
#pragma unroll 8
      for (int j = 0; j < stripes; j++) acc = rotl32(acc + q[4 * j] * XXP2, 13) * XXP1;
    }
    const uint32_t v1 = __builtin_amdgcn_readlane(acc, 0), v2 = __builtin_amdgcn_readlane(acc, 1),
                   v3 = __builtin_amdgcn_readlane(acc, 2), v4 = __builtin_amdgcn_readlane(acc, 3);
    h = rotl32(v1, 1) + rotl32(v2, 7) + rotl32(v3, 12) + rotl32(v4, 18);
    p = stripes << 4;
  } else {
    h = seed + XXP5;
  }
  h += (uint32_t)len;
  for (; p + 4 <= len; p += 4)
    h = rotl32(h + *reinterpret_cast<const uint32_t*>(in + p) * XXP3, 17) * XXP4;
  for (; p < len; p++) h = rotl32(h + (uint32_t)in[p] * XXP5, 11) * XXP1;
  h ^= h >> 15;
  h *= XXP2;
  h ^= h >> 13;
  h *= XXP3;
  h ^= h >> 16;
  return h;
}

__global__ __launch_bounds__(kThreads) void lz4_compress_kernel(const uint8_t* __restrict__ src,
                                                               const Item* __restrict__ items,
                                                               int32_t n_items,
                                                               uint8_t* __restrict__ slots,
                                                               uint32_t* __restrict__ item_size) {
  __shared__ Lz4Lds s;
  const int it = blockIdx.x;
  if (it >= n_items) return;
  const Item item = items[it];
  const int kind = item.kind & 0xff;
  if (kind != kItemLz4Chunk) {
    if (threadIdx.x == 0 && kind == kItemLz4End) item_size[it] = kLz4FrameHeader;
    return;
  }
  const int len = item.len;
  const uint8_t* g = src + item.src_off;
  // stage the chunk (one HBM read of the input) and clear the hash table
  for (int i = threadIdx.x * 16; i + 16 <= len; i += kThreads * 16) {
    uint4 x;
    __builtin_memcpy(&x, g + i, 16);
    *reinterpret_cast<uint4*>(s.in + i) = x;
  }
  for (int i = (len & ~15) + threadIdx.x; i < len; i += kThreads) s.in[i] = g[i];
  {
    uint4* tz = reinterpret_cast<uint4*>(s.table);
    for (int i = threadIdx.x; i < (int)(sizeof(s.table) / 16); i += kThreads) tz[i] = make_uint4(0, 0, 0, 0);
  }
  __syncthreads();

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  uint8_t* slot = slots + (size_t)item.chunk * kSlotBytes;
  int clen = 0;
  if (wave == 0) {
    clen = lz4_compress_wave(s.in, s.table, len, slot + kSlotHeader, lane);
  } else {
    const uint32_t x = xxh32_wave(s.in, len, kLz4BlockSeed, lane);
    if (lane == 0) s.xxh = x;
  }
  __syncthreads();
  if (wave == 0) {
    // LZ4BlockOutputStream.flushBufferedData(): compressedLength >= o  => stored raw
    const bool raw = clen < 0 || clen >= len;
    const uint32_t plen = raw ? (uint32_t)len : (uint32_t)clen;
    const uint32_t check = s.xxh & 0x0FFFFFFFu;
    const uint32_t token = (raw ? 0x10u : 0x20u) | ((uint32_t)(item.kind >> 8) & 0x0Fu);
    uint8_t* hdr = slot + (kSlotHeader - kLz4FrameHeader);
    if (lane < kLz4FrameHeader) {
      const uint64_t magic = 0x6b636f6c42345a4cull;  // "LZ4Block" little-endian
      uint32_t b;
      if (lane < 8) b = (uint32_t)(magic >> (8 * lane));
      else if (lane == 8) b = token;
      else if (lane < 13) b = plen >> (8 * (lane - 9));
      else if (lane < 17) b = (uint32_t)len >> (8 * (lane - 13));
      else b = check >> (8 * (lane - 17));
      hdr[lane] = (uint8_t)b;
    }
    if (lane == 0) item_size[it] = (kLz4FrameHeader + plen) | (raw ? kRawFlag : 0u);
  }
}

}  // namespace

void launch_lz4_compress(const uint8_t* d_src, const Item* d_items, int32_t n_items,
                         uint8_t* d_slots, uint32_t* d_item_size, hipStream_t st) {
  if (n_items <= 0) return;
  hipLaunchKernelGGL(lz4_compress_kernel, dim3((unsigned)n_items), dim3(kThreads), 0, st, d_src,
                     d_items, n_items, d_slots, d_item_size);
}

}  // namespace s3s
