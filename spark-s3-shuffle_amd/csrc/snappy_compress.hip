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
#include "s3s_internal.h"

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

typedef __attribute__((address_space(3))) uint16_t lds_u16;

// The parse of one fragment (chunk <= 32 KiB).  Returns the number of bytes written.
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
    for (;;) {
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
      // FindMatchLength(candidate + 4, ip + 4, ip_end): 256 bytes per round
      int extra = 0;
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
  uint8_t* slot = slots + (size_t)item.chunk * (size_t)slot_stride;
  const int clen = snappy_compress_wave(src + item.src_off, (lds_u16*)table, item.len, slot + kSlotHeader, lane);
  // SnappyOutputStream.dumpOutput(): i32 BE compressed length in front of the raw block
  if (lane < 4) slot[kSlotHeader - 4 + lane] = (uint8_t)((uint32_t)clen >> (8 * (3 - lane)));
  if (lane == 0) item_size[it] = 4u + (uint32_t)clen;
}

}  // namespace

bool snappy_compress_available() { return true; }

void launch_snappy_compress(const uint8_t* d_src, const Item* d_items, int32_t n_items,
                            uint8_t* d_slots, int64_t slot_stride, uint32_t* d_item_size,
                            hipStream_t st) {
  if (n_items <= 0) return;
  hipLaunchKernelGGL(snappy_compress_kernel, dim3((unsigned)n_items), dim3(kWave), 0, st, d_src, d_items,
                     n_items, d_slots, slot_stride, d_item_size);
}

}  // namespace s3s
