// checksum.hip — per-partition Adler32 / CRC32 over byte ranges of the .data image.
//
// Replaces java.util.zip.{Adler32,CRC32} as the reference uses them:
//   write side  [EXT] MutableCheckedOutputStream -> checksums(p) delivered to
//               S3ShuffleMapOutputWriter.scala:91 / S3SingleSpillShuffleMapOutputWriter.scala:27
//   read side   S3ShuffleHelper.scala:94-103 (createChecksumAlgorithm) driven by
//               S3ChecksumValidationStream.scala:54-86 (update per read, compare at the
//               partition boundary).
// Both are defined over a partition's COMPRESSED bytes data[index[p], index[p+1]).
//
// A sequential checksum becomes a two-level reduction because both functions combine:
//   Adler32  a = 1 + sum d_j,  b = L + sum d_j * (L - j)      (all mod 65521)
//   CRC32    crc(X||Y) = crc(X) * x^(8|Y|) mod P  xor  crc(Y)  (zlib crc32_combine identity)
// Level 1: one workgroup per 16 KiB segment of a range; every thread owns a 64-byte piece,
//          right-aligned in the segment so each piece's distance to the segment end is a
//          multiple of 64 (=> the CRC shift operator comes from a 256-entry table).
// Level 2: one workgroup per range folds its segment partials.
// Roofline: HBM read of the range bytes, 1 B/B.
#include "s3s_internal.h"

namespace s3s {
namespace {

constexpr int kThreads = 256;
constexpr int kPiece = 64;
static_assert(kChecksumSegBytes == kThreads * kPiece, "segment = one piece per thread");
constexpr uint32_t kAdlerMod = 65521u;

struct Tables {             // built on the host once per context (codec_api.hip): one set per CRC polynomial
  uint32_t slice[4][256];   // slice-by-4 tables of the reflected polynomial
  uint32_t pow_piece[256];  // x^(8*64*k) mod P
  uint32_t x2n[32];         // x^(2^k) mod P
  uint32_t poly, pad[3];    // 0xEDB88320 (CRC-32, IEEE 802.3: java.util.zip.CRC32) / 0x82F63B78 (CRC-32C, Castagnoli: java.util.zip.CRC32C)
};
constexpr uint32_t kPolyIeee = 0xEDB88320u, kPolyCastagnoli = 0x82F63B78u;

// a(x) * b(x) mod P in the reflected representation (zlib multmodp), branch-free
__device__ __forceinline__ uint32_t multmodp(uint32_t a, uint32_t b, uint32_t poly) {
  uint32_t p = 0;
#pragma unroll
  for (int i = 0; i < 32; i++) {
    p ^= (a & (0x80000000u >> i)) ? b : 0u;
    b = (b >> 1) ^ (poly & (0u - (b & 1u)));
  }
  return p;
}

// x^(8n) mod P
__device__ __forceinline__ uint32_t x8n(const uint32_t* x2n, uint64_t n, uint32_t poly) {
  uint32_t p = 0x80000000u;
  for (int k = 3; n; n >>= 1, k++)
    if (n & 1) p = multmodp(x2n[k & 31], p, poly);
  return p;
}

// CRC-32 without LDS (round 3).  The slice-by-4 tables cost 4 random LDS reads per 4 input bytes: ~4-way bank conflicts, and
// next to a codec kernel every byte of LDS is booked, so the checksum workgroups waited for a parse to finish.  A table
// lookup is linear over GF(2): T[b] = T[b & 0x0f] ^ T[b & 0xf0], so a 256-entry table is two 16-entry ones — and 16 entries
// fit the lanes of ONE register.  Eight registers hold the nibble tables of the four slice positions; a lookup is a
// ds_bpermute_b32 with the nibble as lane index (the LDS crossbar, no LDS memory; lanes that read the same source lane
// are a broadcast, different ones hit different banks: conflict-free by construction).
// Round 5: SIX-bit tables for the word step.  ds_bpermute_b32 picks one of 64 lanes, so a table of 64 entries costs the
// same lookup as one of 16: the 32-bit word is cut into the bit groups [0:5] [6:11] [12:17] [18:23] [24:29] [30:31] — five
// lookups per word (the two top bits: two masked xors on the vector unit) instead of eight (the kernel is bound by the LDS
// crossbar: 2.08 TB/s with eight, 2.5 - 2.6 with six).  A group that
// straddles a byte boundary is the xor of two byte-table entries, folded in when the register is built.  The byte step
// (short first pieces only) keeps its two nibble tables.
struct NibbleTabs {
  uint32_t g[5];  // g[k]: lane v holds f(v << 6 k), f = the four-byte step slice[3][b0] ^ slice[2][b1] ^ slice[1][b2] ^ slice[0][b3]
  uint32_t f30, f31;  // f(1 << 30), f(1 << 31): the sixth group has two bits — two masked xors on the vector unit instead of a lookup
  uint32_t b[2];  // b[h]: lane v holds slice[0][h ? (v & 15) << 4 : v & 15]   (one-byte step)
};
__device__ __forceinline__ NibbleTabs load_nibble_tabs(const Tables* tabs, int lane) {
  NibbleTabs n;
  const uint32_t v = (uint32_t)lane & 63u;
#pragma unroll
  for (int k = 0; k < 5; k++) {
    const uint32_t x = v << (6 * k);
    n.g[k] = tabs->slice[3][x & 0xff] ^ tabs->slice[2][(x >> 8) & 0xff] ^ tabs->slice[1][(x >> 16) & 0xff] ^ tabs->slice[0][x >> 24];
  }
  n.f30 = tabs->slice[0][0x40];
  n.f31 = tabs->slice[0][0x80];
  // (f(0) = 0 for every slice table, so the bytes a group does not touch add nothing — but slice[j][0] is read four times
  // per register; the tables are 4 KiB and stay in cache)
  const uint32_t w = (uint32_t)lane & 15u;
  n.b[0] = tabs->slice[0][w];
  n.b[1] = tabs->slice[0][w << 4];
  return n;
}
__device__ __forceinline__ uint32_t nib_lookup(uint32_t table_reg, uint32_t idx_times_4) {
  return (uint32_t)__builtin_amdgcn_ds_bpermute((int)idx_times_4, (int)table_reg);
}
// one slice-by-4 step: c (already xor-ed with the next word) -> crc after these four bytes
__device__ __forceinline__ uint32_t crc_word(const NibbleTabs& n, uint32_t c) {
  const uint32_t a0 = nib_lookup(n.g[0], (c << 2) & 0xfcu), a1 = nib_lookup(n.g[1], (c >> 4) & 0xfcu);
  const uint32_t a2 = nib_lookup(n.g[2], (c >> 10) & 0xfcu), a3 = nib_lookup(n.g[3], (c >> 16) & 0xfcu);
  const uint32_t a4 = nib_lookup(n.g[4], (c >> 22) & 0xfcu);
  const uint32_t a5 = ((uint32_t)((int32_t)(c << 1) >> 31) & n.f30) ^ ((uint32_t)((int32_t)c >> 31) & n.f31);
  return (a0 ^ a1 ^ a2) ^ (a3 ^ a4 ^ a5);
}
__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const int o = __shfl_xor(v, d);
    v = o > v ? o : v;
  }
  return v;
}
// one byte: crc = slice[0][(c ^ b) & 0xff] ^ (c >> 8); slice[0] is position j = 3: registers 6, 7
__device__ __forceinline__ uint32_t crc_byte(const NibbleTabs& n, uint32_t c, uint32_t b) {
  const uint32_t x = c ^ b;
  return nib_lookup(n.b[0], (x << 2) & 0x3cu) ^ nib_lookup(n.b[1], (x >> 2) & 0x3cu) ^ (c >> 8);
}

// partial[seg] = { A|crc, B, seg_len, 0 }
template <int ALGO>
__device__ __forceinline__ void checksum_segment_body(
    const uint8_t* __restrict__ data, const int64_t* __restrict__ offsets, int32_t n,
    const int32_t* __restrict__ seg_start, const Tables* __restrict__ tabs,
    uint32_t* __restrict__ partial, int64_t data_len, const int b) {
  const int tid = threadIdx.x;
  // which range owns worst-case segment slot b
  int lo = 0, hi = n;  // seg_start[lo] <= b < seg_start[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (seg_start[mid] <= b) lo = mid; else hi = mid;
  }
  const int p = lo;
  const int s = b - seg_start[p];
  const int64_t pstart = offsets[p], plen = offsets[p + 1] - pstart;
  const int64_t soff = (int64_t)s * kChecksumSegBytes;
  if (soff >= plen) return;  // slot beyond the actual (compressed) length
  const int seg_len = (int)((plen - soff) < kChecksumSegBytes ? (plen - soff) : kChecksumSegBytes);
  // never read past the caller's buffer: after a capacity overflow the device-computed offsets exceed it
  // (the call reports S3S_E_CAPACITY; the sums of such ranges are not used)
  if (pstart + soff + seg_len > data_len) return;
  const uint8_t* g = data + pstart + soff;

  NibbleTabs nt;
  if (ALGO == S3S_CHECKSUM_CRC32) nt = load_nibble_tabs(tabs, tid);
  // pieces are right-aligned: thread t of T owns [seg_len - 64*(T-t), seg_len - 64*(T-1-t))
  const int T = (seg_len + kPiece - 1) / kPiece;
  uint32_t v0 = 0, v1 = 0;
  if (ALGO == S3S_CHECKSUM_CRC32) {
    // every lane of the wavefront takes part in the cross-lane lookups: no early-out per thread; idle threads chew zeros
    const bool mine = tid < T;
    const int end = mine ? seg_len - kPiece * (T - 1 - tid) : 0;
    const int beg = end - kPiece > 0 ? end - kPiece : 0;
    const int pl = end - beg;
    uint32_t c = 0xFFFFFFFFu;
    const bool full = pl == kPiece;
    uint4 q[4] = {};
    if (full) __builtin_memcpy(q, g + beg, 64);
    const uint32_t* wds = reinterpret_cast<const uint32_t*>(q);
    if (__ballot(full)) {  // (wave-uniform: the lookups below are executed by all 64 lanes)
      uint32_t cf = 0xFFFFFFFFu;
#pragma unroll
      for (int j = 0; j < 16; j++) cf = crc_word(nt, cf ^ wds[j]);
      c = full ? cf : c;
    }
    // the segment's first piece may be short (right-aligned pieces): byte steps, by whoever has one
    const int maxpl = __builtin_amdgcn_readfirstlane(wave_max_i32(full ? 0 : pl));
    for (int k = 0; k < maxpl; k++) {
      const bool on = !full && k < pl;
      const uint32_t bv = on ? (uint32_t)g[beg + k] : 0u;
      const uint32_t cn = crc_byte(nt, c, bv);
      c = on ? cn : c;
    }
    c = ~c;
    v0 = mine ? multmodp(tabs->pow_piece[T - 1 - tid], c, tabs->poly) : 0u;  // shift by the bytes after this piece
  } else {
    // Adler32 (round 5): position weights instead of pieces.  A byte at index i of a segment of L bytes adds d to A and
    // d * (L - i) to B, whoever reads it — so the loads are laid out for the memory system: thread t takes the 16 bytes
    // [4096 r + 16 t, + 16) of the segment for r = 0..3 (a wavefront reads 1 KiB in one instruction; the pieces of round 1
    // were 64 bytes per THREAD, 64 bytes apart: four instructions to use a cache line, 3.9 TB/s on a 1 GiB range), and
    // the sums come from v_dot4_u32_u8: S = sum d_j, W = sum j d_j per 16 bytes, B += (L - c) S - W.
    uint32_t s1 = 0, s2 = 0;
    auto add16 = [&](const uint4& q, int c) {
      uint32_t S = __builtin_amdgcn_udot4(q.x, 0x01010101u, 0u, false);
      S = __builtin_amdgcn_udot4(q.y, 0x01010101u, S, false);
      S = __builtin_amdgcn_udot4(q.z, 0x01010101u, S, false);
      S = __builtin_amdgcn_udot4(q.w, 0x01010101u, S, false);
      uint32_t W = __builtin_amdgcn_udot4(q.x, 0x03020100u, 0u, false);
      W = __builtin_amdgcn_udot4(q.y, 0x07060504u, W, false);
      W = __builtin_amdgcn_udot4(q.z, 0x0b0a0908u, W, false);
      W = __builtin_amdgcn_udot4(q.w, 0x0f0e0d0cu, W, false);
      s1 += S;                                // <= 4 * 4080
      s2 += (uint32_t)(seg_len - c) * S - W;  // <= 4 * 16384 * 4080 < 2^32; W <= (L - c) S (zero padding has no weight)
    };
    constexpr int kRows = kChecksumSegBytes / (16 * kThreads);
    if (seg_len == kChecksumSegBytes) {  // a whole segment (all but a range's last): every load goes out before the first sum
      uint4 q[kRows];
#pragma unroll
      for (int r = 0; r < kRows; r++) __builtin_memcpy(&q[r], g + 16 * kThreads * r + 16 * tid, 16);
#pragma unroll
      for (int r = 0; r < kRows; r++) add16(q[r], 16 * kThreads * r + 16 * tid);
    } else {
#pragma unroll
      for (int r = 0; r < kRows; r++) {
        const int c = 16 * kThreads * r + 16 * tid;
        if (c < seg_len) {
          uint4 q = make_uint4(0, 0, 0, 0);
          if (c + 16 <= seg_len) {
            __builtin_memcpy(&q, g + c, 16);
          } else {  // the segment's last, short chunk: byte loads, the rest stays zero
            uint32_t w[4] = {0, 0, 0, 0};
            for (int k = 0; k < seg_len - c; k++) w[k >> 2] |= (uint32_t)g[c + k] << (8 * (k & 3));
            q = make_uint4(w[0], w[1], w[2], w[3]);
          }
          add16(q, c);
        }
      }
    }
    v0 = s1;
    v1 = s2 % kAdlerMod;
  }
  uint32_t* out = partial + 4 * (size_t)b;
  if (ALGO == S3S_CHECKSUM_ADLER32) {
    // NO LDS in the Adler32 instantiation (the default algorithm): next to another task thread's codec
    // kernel every byte of LDS is booked, and a workgroup that asks for 16 bytes of it waits for a
    // ~1 ms chunk to finish.  Wavefront sums go to the zero-initialised partial with atomics; the
    // combine kernel applies the modulus.  A <= 255*16384, B <= 256*65520: no overflow.
    uint32_t A = v0, B = v1;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      A += __shfl_xor(A, d);
      B += __shfl_xor(B, d);
    }
    if ((tid & 63) == 0) {
      atomicAdd(&out[0], A);
      atomicAdd(&out[1], B);
    }
    if (tid == 0) out[2] = (uint32_t)seg_len;
  } else {
    // NO LDS either: wavefront xor, then one atomic per wavefront into the zero-initialised partial
    uint32_t c = v0;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c ^= __shfl_xor(c, d);
    if ((tid & 63) == 0) atomicXor(&out[0], c);
    if (tid == 0) out[2] = (uint32_t)seg_len;
  }
}

// unit = bytes one partial stands for (all but a range's last): kChecksumSegBytes, or — behind checksum_fold_kernel —
// fold_groups x kChecksumSegBytes with the partials of range p at partial + 4 * p * max_groups
template <int ALGO>
__global__ __launch_bounds__(kThreads) void checksum_segments_kernel(
    const uint8_t* __restrict__ data, const int64_t* __restrict__ offsets, int32_t n,
    const int32_t* __restrict__ seg_start, const Tables* __restrict__ tabs,
    uint32_t* __restrict__ partial, int64_t data_len) {
  checksum_segment_body<ALGO>(data, offsets, n, seg_start, tabs, partial, data_len, (int)blockIdx.x);
}

// the last t with first(t) <= i (wave-uniform: scalar loads; see assemble.hip)
template <typename F>
__device__ __forceinline__ int owner_task(int32_t n_tasks, int32_t i, F first) {
  int lo = 0, hi = n_tasks;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (first(mid) <= i) lo = mid; else hi = mid;
  }
  return lo;
}

// the segments of EVERY task / fetched range of a batched call in one launch (TaskTail, s3s_internal.h)
template <int ALGO>
__global__ __launch_bounds__(kThreads) void checksum_segments_batch_kernel(
    const TaskTail* __restrict__ tails, int32_t n_tasks, const int64_t* __restrict__ offsets,
    const int32_t* __restrict__ seg_start, const Tables* __restrict__ tabs, uint32_t* __restrict__ partial) {
  const int g = blockIdx.x;
  const int t = owner_task(n_tasks, g, [&](int m) { return tails[m].first_seg; });
  const TaskTail k = tails[t];
  if (g - k.first_seg >= k.n_segs || k.n_parts <= 0) return;
  checksum_segment_body<ALGO>(k.data, offsets + k.first_pp, k.n_parts, seg_start + k.first_pp, tabs, partial + 4 * (size_t)k.first_seg,
                              k.data_len, g - k.first_seg);
}

template <int ALGO>
__device__ __forceinline__ void checksum_combine_body(
    const int64_t* __restrict__ offsets, int32_t n, const int32_t* __restrict__ seg_start,
    const Tables* __restrict__ tabs, const uint32_t* __restrict__ partial,
    int64_t* __restrict__ out, int64_t unit, int32_t max_groups, const int p) {
  const int tid = threadIdx.x;
  if (p >= n) return;
  const int64_t plen = offsets[p + 1] - offsets[p];
  const int64_t nseg = (plen + unit - 1) / unit;
  const uint32_t* part = partial + 4 * (max_groups > 0 ? (size_t)p * (size_t)max_groups : (size_t)seg_start[p]);
  if (ALGO == S3S_CHECKSUM_ADLER32) {
    // one wavefront per partition, no LDS (see checksum_segments_kernel)
    uint32_t sa = 0, sb = 0;
    for (int64_t s = tid; s < nseg; s += kWave) {
      const uint32_t A = part[4 * s] % kAdlerMod, B = part[4 * s + 1] % kAdlerMod, len = part[4 * s + 2];
      const int64_t after = plen - (s * unit + len);
      sa = (sa + A) % kAdlerMod;
      sb = (uint32_t)((sb + B + (uint64_t)A * (uint64_t)(after % kAdlerMod)) % kAdlerMod);
    }
    uint32_t a = sa, b = sb;  // 64 values < 65521 each: no overflow
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      a += __shfl_xor(a, d);
      b += __shfl_xor(b, d);
    }
    if (tid == 0) {
      const uint32_t fa = (1u + a) % kAdlerMod;
      const uint32_t fb = (uint32_t)(((uint64_t)(plen % kAdlerMod) + b) % kAdlerMod);
      out[p] = (int64_t)(((uint64_t)fb << 16) | fa);
    }
  } else {
    // one wavefront per partition, no LDS (see checksum_segments_kernel): lane j folds a contiguous run of segments
    // Horner-style, shifts the run to the end of the partition, the runs are xor-ed across the wave
    const uint32_t* x2n = tabs->x2n;
    const uint32_t poly = tabs->poly;
    const int64_t run = (nseg + kWave - 1) / kWave;
    const int64_t s0 = (int64_t)tid * run, s1 = (s0 + run) < nseg ? (s0 + run) : nseg;
    uint32_t c = 0;
    int64_t end = 0;
    if (s0 < s1) {
      const uint32_t xseg = x8n(x2n, (uint64_t)unit, poly);
      for (int64_t s = s0; s < s1; s++) {
        const uint32_t len = part[4 * s + 2];
        c = multmodp(c, (int64_t)len == unit ? xseg : x8n(x2n, len, poly), poly) ^ part[4 * s];
        end = s * unit + len;
      }
      c = multmodp(c, x8n(x2n, (uint64_t)(plen - end), poly), poly);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c ^= __shfl_xor(c, d);
    if (tid == 0) out[p] = (int64_t)(uint64_t)c;
  }
}


template <int ALGO>
__global__ __launch_bounds__(kThreads) void checksum_combine_kernel(
    const int64_t* __restrict__ offsets, int32_t n, const int32_t* __restrict__ seg_start,
    const Tables* __restrict__ tabs, const uint32_t* __restrict__ partial,
    int64_t* __restrict__ out, int64_t unit, int32_t max_groups) {
  checksum_combine_body<ALGO>(offsets, n, seg_start, tabs, partial, out, unit, max_groups, (int)blockIdx.x);
}

// one wavefront per partition of EVERY task of a batched call
template <int ALGO>
__global__ __launch_bounds__(kWave) void checksum_combine_batch_kernel(
    const TaskTail* __restrict__ tails, int32_t n_tasks, int32_t total_parts, const int64_t* __restrict__ offsets,
    const int32_t* __restrict__ seg_start, const Tables* __restrict__ tabs, const uint32_t* __restrict__ partial,
    int64_t* __restrict__ out) {
  const int g = blockIdx.x;
  if (g >= total_parts) return;
  const int t = owner_task(n_tasks, g, [&](int m) { return tails[m].first_part; });
  const TaskTail k = tails[t];
  if (g - k.first_part >= k.n_parts) return;
  checksum_combine_body<ALGO>(offsets + k.first_pp, k.n_parts, seg_start + k.first_pp, tabs, partial + 4 * (size_t)k.first_seg,
                              out + k.first_part, (int64_t)kChecksumSegBytes, 0, g - k.first_part);
}


// A range of very many segments (a 1 GiB single-partition block: 65 536) would have the combine kernel's ONE wavefront fold
// them all — 0.37 ms of the 0.65 ms a 1 GiB Adler32 took in round 4's layout.  The fold kernel is the same arithmetic one
// level down: one wavefront per (range, group of `G` consecutive segments) writes the group's partial
// { A | crc relative to the group's end, B, bytes } to partial2[p * max_groups + g]; the combine kernel then runs over
// groups (unit = G segments).  Launched only when a range is that large (launch_checksum_with_tables).
template <int ALGO>
__global__ __launch_bounds__(kWave) void checksum_fold_kernel(
    const int64_t* __restrict__ offsets, int32_t n, const int32_t* __restrict__ seg_start,
    const Tables* __restrict__ tabs, const uint32_t* __restrict__ partial, uint32_t* __restrict__ partial2,
    int32_t G, int32_t max_groups) {
  const int p = (int)(blockIdx.x / (uint32_t)max_groups), gi = (int)(blockIdx.x % (uint32_t)max_groups), tid = threadIdx.x;
  if (p >= n) return;
  const int64_t plen = offsets[p + 1] - offsets[p];
  const int64_t nseg = (plen + kChecksumSegBytes - 1) / kChecksumSegBytes;
  const int64_t s_lo = (int64_t)gi * G;
  if (s_lo >= nseg) return;
  const int64_t s_hi = s_lo + G < nseg ? s_lo + G : nseg;
  const uint32_t* part = partial + 4 * (size_t)seg_start[p];
  uint32_t* o = partial2 + 4 * ((size_t)p * (size_t)max_groups + (size_t)gi);
  const int64_t gend = s_hi * kChecksumSegBytes < plen ? s_hi * kChecksumSegBytes : plen;  // (bytes from the range's start)
  if (ALGO == S3S_CHECKSUM_ADLER32) {
    uint32_t sa = 0, sb = 0;
    for (int64_t s = s_lo + tid; s < s_hi; s += kWave) {
      const uint32_t A = part[4 * s] % kAdlerMod, B = part[4 * s + 1] % kAdlerMod, len = part[4 * s + 2];
      const int64_t after = gend - (s * kChecksumSegBytes + len);
      sa = (sa + A) % kAdlerMod;
      sb = (uint32_t)((sb + B + (uint64_t)A * (uint64_t)(after % kAdlerMod)) % kAdlerMod);
    }
    uint32_t a = sa, b = sb;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      a += __shfl_xor(a, d);
      b += __shfl_xor(b, d);
    }
    if (tid == 0) {
      o[0] = a % kAdlerMod;
      o[1] = b % kAdlerMod;
      o[2] = (uint32_t)(gend - s_lo * kChecksumSegBytes);
      o[3] = 0;
    }
  } else {
    const uint32_t* x2n = tabs->x2n;
    const uint32_t poly = tabs->poly;
    const int64_t cnt = s_hi - s_lo, run = (cnt + kWave - 1) / kWave;
    const int64_t s0 = s_lo + (int64_t)tid * run, s1 = (s0 + run) < s_hi ? (s0 + run) : s_hi;
    uint32_t c = 0;
    if (s0 < s1) {
      const uint32_t xseg = x8n(x2n, kChecksumSegBytes, poly);
      int64_t end = 0;
      for (int64_t s = s0; s < s1; s++) {
        const uint32_t len = part[4 * s + 2];
        c = multmodp(c, len == (uint32_t)kChecksumSegBytes ? xseg : x8n(x2n, len, poly), poly) ^ part[4 * s];
        end = s * kChecksumSegBytes + len;
      }
      c = multmodp(c, x8n(x2n, (uint64_t)(gend - end), poly), poly);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c ^= __shfl_xor(c, d);
    if (tid == 0) {
      o[0] = c;
      o[1] = 0;
      o[2] = (uint32_t)(gend - s_lo * kChecksumSegBytes);
      o[3] = 0;
    }
  }
}

}  // namespace

size_t checksum_tables_bytes() { return 2 * sizeof(Tables); }  // [0] CRC-32, [1] CRC-32C

// host-side construction of the constant tables (uploaded once per context)
void checksum_tables_build(void* host_buf) {
  for (int which = 0; which < 2; which++) {
    Tables* t = static_cast<Tables*>(host_buf) + which;
    const uint32_t poly = which == 0 ? kPolyIeee : kPolyCastagnoli;
    t->poly = poly;
    t->pad[0] = t->pad[1] = t->pad[2] = 0;
    for (uint32_t i = 0; i < 256; i++) {
      uint32_t c = i;
      for (int k = 0; k < 8; k++) c = (c & 1) ? (poly ^ (c >> 1)) : (c >> 1);
      t->slice[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; i++)
      for (int s = 1; s < 4; s++)
        t->slice[s][i] = (t->slice[s - 1][i] >> 8) ^ t->slice[0][t->slice[s - 1][i] & 0xff];
    auto mul = [poly](uint32_t a, uint32_t b) {
      uint32_t p = 0;
      for (int i = 0; i < 32; i++) {
        if (a & (0x80000000u >> i)) p ^= b;
        b = (b >> 1) ^ (poly & (0u - (b & 1u)));
      }
      return p;
    };
    t->x2n[0] = 0x40000000u;  // x^1
    for (int k = 1; k < 32; k++) t->x2n[k] = mul(t->x2n[k - 1], t->x2n[k - 1]);
    // x^(8*64) = x^(2^9)
    const uint32_t xp = t->x2n[9];
    t->pow_piece[0] = 0x80000000u;  // 1
    for (int k = 1; k < 256; k++) t->pow_piece[k] = mul(t->pow_piece[k - 1], xp);
  }
}

void launch_checksum_with_tables(int algo, const uint8_t* d_data, const int64_t* d_offsets,
                                 int32_t n, const int32_t* d_seg_start, int32_t total_segs,
                                 const void* d_tables, uint32_t* d_partial, int64_t* d_out,
                                 int64_t data_len, hipStream_t st, int32_t max_segs_per_range, uint32_t* d_partial2) {
  if (n <= 0) return;
  const Tables* tabs = static_cast<const Tables*>(d_tables) + (algo == S3S_CHECKSUM_CRC32C ? 1 : 0);  // (the CRC kernels are the polynomial's tables away from each other)
  // ranges of more than kChecksumFoldFrom segments: fold groups of kChecksumFoldGroup segments first (d_partial2 holds
  // n x max_groups partials; the caller sizes it with checksum_fold_groups())
  const int32_t max_groups = d_partial2 ? checksum_fold_groups(n, max_segs_per_range) : 0;
  const int64_t unit = max_groups > 0 ? (int64_t)kChecksumFoldGroup * kChecksumSegBytes : (int64_t)kChecksumSegBytes;
  const uint32_t* comb_in = max_groups > 0 ? d_partial2 : d_partial;
  if (total_segs > 0) (void)hipMemsetAsync(d_partial, 0, 16 * (size_t)total_segs, st);  // wavefront results are added / xor-ed in atomically
  if (algo == S3S_CHECKSUM_ADLER32) {
    if (total_segs > 0)
      hipLaunchKernelGGL(checksum_segments_kernel<S3S_CHECKSUM_ADLER32>, dim3((unsigned)total_segs),
                         dim3(kThreads), 0, st, d_data, d_offsets, n, d_seg_start, tabs, d_partial, data_len);
    if (max_groups > 0)
      hipLaunchKernelGGL(checksum_fold_kernel<S3S_CHECKSUM_ADLER32>, dim3((unsigned)max_groups * (unsigned)n), dim3(kWave), 0, st,
                         d_offsets, n, d_seg_start, tabs, d_partial, d_partial2, kChecksumFoldGroup, max_groups);
    hipLaunchKernelGGL(checksum_combine_kernel<S3S_CHECKSUM_ADLER32>, dim3((unsigned)n),
                       dim3(kWave), 0, st, d_offsets, n, d_seg_start, tabs, comb_in, d_out, unit, max_groups);
  } else {
    if (total_segs > 0)
      hipLaunchKernelGGL(checksum_segments_kernel<S3S_CHECKSUM_CRC32>, dim3((unsigned)total_segs),
                         dim3(kThreads), 0, st, d_data, d_offsets, n, d_seg_start, tabs, d_partial, data_len);
    if (max_groups > 0)
      hipLaunchKernelGGL(checksum_fold_kernel<S3S_CHECKSUM_CRC32>, dim3((unsigned)max_groups * (unsigned)n), dim3(kWave), 0, st,
                         d_offsets, n, d_seg_start, tabs, d_partial, d_partial2, kChecksumFoldGroup, max_groups);
    hipLaunchKernelGGL(checksum_combine_kernel<S3S_CHECKSUM_CRC32>, dim3((unsigned)n),
                       dim3(kWave), 0, st, d_offsets, n, d_seg_start, tabs, comb_in, d_out, unit, max_groups);
  }
}

void launch_checksum_batch(int algo, const TaskTail* d_tails, int32_t n_tasks, int32_t total_segs, int32_t total_parts,
                           const int64_t* d_offsets, const int32_t* d_seg_start, const void* d_tables, uint32_t* d_partial,
                           int64_t* d_out, hipStream_t st) {
  if (n_tasks <= 0 || total_parts <= 0) return;
  const Tables* tabs = static_cast<const Tables*>(d_tables) + (algo == S3S_CHECKSUM_CRC32C ? 1 : 0);
  if (total_segs > 0) (void)hipMemsetAsync(d_partial, 0, 16 * (size_t)total_segs, st);
  if (algo == S3S_CHECKSUM_ADLER32) {
    if (total_segs > 0)
      hipLaunchKernelGGL(checksum_segments_batch_kernel<S3S_CHECKSUM_ADLER32>, dim3((unsigned)total_segs), dim3(kThreads), 0, st,
                         d_tails, n_tasks, d_offsets, d_seg_start, tabs, d_partial);
    hipLaunchKernelGGL(checksum_combine_batch_kernel<S3S_CHECKSUM_ADLER32>, dim3((unsigned)total_parts), dim3(kWave), 0, st, d_tails,
                       n_tasks, total_parts, d_offsets, d_seg_start, tabs, d_partial, d_out);
  } else {
    if (total_segs > 0)
      hipLaunchKernelGGL(checksum_segments_batch_kernel<S3S_CHECKSUM_CRC32>, dim3((unsigned)total_segs), dim3(kThreads), 0, st,
                         d_tails, n_tasks, d_offsets, d_seg_start, tabs, d_partial);
    hipLaunchKernelGGL(checksum_combine_batch_kernel<S3S_CHECKSUM_CRC32>, dim3((unsigned)total_parts), dim3(kWave), 0, st, d_tails,
                       n_tasks, total_parts, d_offsets, d_seg_start, tabs, d_partial, d_out);
  }
}

}  // namespace s3s
