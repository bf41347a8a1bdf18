// lz4_decompress.hip — reduce side: find the LZ4Block frames of a fetched range, decode them.
//
// Replaces the [EXT] LZ4BlockInputStream(stopOnEmptyBlock=false).refill() loop that
// serializerManager.wrapStream installs at S3ShuffleReader.scala:108 on top of the byte range
// S3ShuffleBlockStream exposes (S3ShuffleBlockStream.scala:36-40): parse a 21-byte header,
// check magic / token / lengths, LZ4-decode (or copy, method RAW) originalLen bytes, verify
// xxh32 & 0x0FFFFFFF, and on an end-of-stream frame keep going with the next concatenated
// stream (what makes batch fetch and multi-spill merge legal, S3ShuffleReader.scala:55-75).
//
// A frame's position is only known from its predecessor's compressedLen — a serial pointer
// chase of one HBM round trip per ~8 KiB.  For a 1 GiB single-partition block (BASELINE
// config 5) that is 30k+ dependent misses, so the chain is discovered speculatively:
//   tile_speculate  every 64 KiB tile (one wavefront) scans for the first plausible header
//                   ("LZ4Block" + sane fields) and walks the chain from there to the tile end
//   tile_resolve    one wavefront checks exit(k-1) == entry(k) for all tiles, 64 at a time;
//                   a tile whose speculation was wrong (magic bytes inside a payload) is
//                   re-walked from its true entry
//   tile_emit       every tile walks once more from its verified entry, validating headers
//                   exactly like refill() and writing Frame records at their scanned index
// then one workgroup per frame decodes into LDS and streams the 32 KiB result out coalesced,
// with the frame hash computed by a second wavefront trailing the decoder.
#include "s3s_internal.h"

#ifdef S3S_LZ4_TIMING
__device__ unsigned long long g_dec_dbg[16];
#define DDBG_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define DDBG_ADD(slot, val) ddbg[slot] += (val)
#else
#define DDBG_T(var)
#define DDBG_ADD(slot, val)
#endif

namespace s3s {
namespace {

constexpr int kTileBytes = 65536;
constexpr uint64_t kMagic = 0x6b636f6c42345a4cull;  // "LZ4Block" little-endian

__device__ __forceinline__ uint64_t ld64u(const uint8_t* p) {
  uint64_t v;
  __builtin_memcpy(&v, p, 8);
  return v;
}
__device__ __forceinline__ uint32_t ld32u(const uint8_t* p) {
  uint32_t v;
  __builtin_memcpy(&v, p, 4);
  return v;
}

struct Header {
  int32_t method, comp_len, orig_len;
  uint32_t check;
  bool ok;
};

// LZ4BlockInputStream.refill() header checks (magic excluded)
__device__ __forceinline__ Header parse_header(const uint8_t* h) {
  Header r;
  const uint32_t token = h[8];
  r.method = (int32_t)(token & 0xF0u);
  const int level = 10 + (int)(token & 0x0Fu);
  r.comp_len = (int32_t)ld32u(h + 9);
  r.orig_len = (int32_t)ld32u(h + 13);
  r.check = ld32u(h + 17);
  r.ok = (r.method == 0x10 || r.method == 0x20) && r.orig_len >= 0 && r.comp_len >= 0 &&
         r.orig_len <= (1 << level) && !(r.orig_len == 0 && r.comp_len != 0) &&
         !(r.orig_len != 0 && r.comp_len == 0) && !(r.method == 0x10 && r.orig_len != r.comp_len) &&
         !(r.orig_len == 0 && r.check != 0);
  return r;
}

// Sequential walk (one lane) from `pos` until the chain leaves [.., tile_end) or reaches
// comp_len.  Returns the exit position, or -1 on a malformed header / overrun.  Optionally
// writes Frame records.
__device__ int64_t walk_tile(const uint8_t* comp, int64_t comp_len, int64_t pos, int64_t tile_end,
                             int32_t* count_out, Frame* frames, uint32_t* frame_orig) {
  int32_t n = 0;
  while (pos < tile_end && pos < comp_len) {
    if (comp_len - pos < kLz4FrameHeader) return -1;  // "Stream ended prematurely"
    const uint8_t* h = comp + pos;
    if (ld64u(h) != kMagic) return -1;
    const Header hd = parse_header(h);
    if (!hd.ok) return -1;
    const int64_t next = pos + kLz4FrameHeader + hd.comp_len;
    if (next > comp_len) return -1;
    if (frames) {
      frames[n] = Frame{pos + kLz4FrameHeader, hd.comp_len, hd.orig_len, hd.check, hd.method};
      frame_orig[n] = (uint32_t)hd.orig_len;
    }
    n++;
    pos = next;
  }
  *count_out = n;
  return pos;
}

// per tile: spec[k] = {entry, exit, count}
__global__ __launch_bounds__(kWave) void tile_speculate_kernel(const uint8_t* __restrict__ comp,
                                                              int64_t comp_len, int32_t n_tiles,
                                                              int64_t* __restrict__ spec_entry,
                                                              int64_t* __restrict__ spec_exit,
                                                              int32_t* __restrict__ spec_count) {
  const int k = blockIdx.x, lane = threadIdx.x;
  if (k >= n_tiles) return;
  const int64_t t0 = (int64_t)k * kTileBytes;
  const int64_t t1 = (t0 + kTileBytes) < comp_len ? (t0 + kTileBytes) : comp_len;
  int64_t entry = -1;
  if (k == 0) {
    entry = 0;
  } else {
    // first plausible header in the tile: 64 positions per step
    for (int64_t p0 = t0; p0 < t1 && entry < 0; p0 += kWave) {
      const int64_t p = p0 + lane;
      bool hit = false;
      if (p < t1 && comp_len - p >= kLz4FrameHeader && ld64u(comp + p) == kMagic) {
        const Header hd = parse_header(comp + p);
        hit = hd.ok && p + kLz4FrameHeader + hd.comp_len <= comp_len;
      }
      const uint64_t m = __ballot(hit);
      if (m) entry = p0 + __builtin_ctzll(m);
    }
  }
  if (lane == 0) {
    int32_t cnt = 0;
    int64_t ex = -1;
    if (entry >= 0) ex = walk_tile(comp, comp_len, entry, t1, &cnt, nullptr, nullptr);
    spec_entry[k] = entry;
    spec_exit[k] = ex;
    spec_count[k] = cnt;
  }
}

// one wavefront: turn speculation into the true chain.  true_entry[k] = position where the
// chain enters tile k (-1: no frame starts in tile k), count[k] = frames starting in tile k.
__global__ __launch_bounds__(kWave) void tile_resolve_kernel(
    const uint8_t* __restrict__ comp, int64_t comp_len, int32_t n_tiles,
    const int64_t* __restrict__ spec_entry, int64_t* __restrict__ spec_exit,
    int32_t* __restrict__ spec_count, int64_t* __restrict__ true_entry, int32_t* __restrict__ status) {
  const int lane = threadIdx.x;
  int64_t e = 0;  // chain position entering the next unresolved tile
  int k = 0;
  while (k < n_tiles) {
    // fast path: up to 64 consecutive tiles whose speculation chains up
    const int kk = k + lane;
    bool ok = false;
    if (kk < n_tiles) {
      const int64_t want = (lane == 0) ? e : spec_exit[kk - 1];
      ok = (spec_entry[kk] == want) && spec_exit[kk] >= 0;
    }
    const uint64_t bad = ~__ballot(ok);
    const int good = bad ? __builtin_ctzll(bad) : kWave;
    if (lane < good) true_entry[k + lane] = spec_entry[k + lane];
    if (good > 0) {
      e = spec_exit[k + good - 1];
      k += good;
      continue;
    }
    // tile k: speculation does not apply
    const int64_t t0 = (int64_t)k * kTileBytes;
    const int64_t t1 = (t0 + kTileBytes) < comp_len ? (t0 + kTileBytes) : comp_len;
    if (e >= t1) {  // the chain skips this tile entirely (a frame larger than a tile)
      if (lane == 0) {
        true_entry[k] = -1;
        spec_count[k] = 0;
        spec_exit[k] = e;
      }
    } else {
      int32_t cnt = 0;
      const int64_t ex = walk_tile(comp, comp_len, e, t1, &cnt, nullptr, nullptr);  // uniform
      if (ex < 0) {
        if (lane == 0) atomicExch(status, S3S_E_BAD_FRAME);
        return;
      }
      if (lane == 0) {
        true_entry[k] = e;
        spec_count[k] = cnt;
        spec_exit[k] = ex;
      }
      e = ex;
    }
    __threadfence();  // later iterations read spec_exit[k] through other lanes
    k += 1;
  }
  if (e != comp_len && lane == 0) atomicExch(status, S3S_E_BAD_FRAME);
}

__global__ __launch_bounds__(kWave) void tile_emit_kernel(
    const uint8_t* __restrict__ comp, int64_t comp_len, int32_t n_tiles,
    const int64_t* __restrict__ true_entry, const int64_t* __restrict__ frame_base,
    Frame* __restrict__ frames, uint32_t* __restrict__ frame_orig, int32_t* __restrict__ status) {
  const int k = blockIdx.x * kWave + threadIdx.x;
  if (k >= n_tiles) return;
  const int64_t entry = true_entry[k];
  if (entry < 0) return;
  const int64_t t0 = (int64_t)k * kTileBytes;
  const int64_t t1 = (t0 + kTileBytes) < comp_len ? (t0 + kTileBytes) : comp_len;
  int32_t cnt = 0;
  const int64_t base = frame_base[k];
  if (walk_tile(comp, comp_len, entry, t1, &cnt, frames + base, frame_orig + base) < 0)
    atomicExch(status, S3S_E_BAD_FRAME);
}

// generic exclusive scan of uint32 -> int64 (n+1 outputs): ONE wavefront, NO LDS — the decode kernels of
// the other task threads book the whole LDS of every CU (20 x 8 KiB rings), and a workgroup that needs
// any of it waits for one of their frames to finish (see scan_items_kernel in assemble.hip)
__global__ __launch_bounds__(kWave) void scan_u32_kernel(const uint32_t* __restrict__ in, int64_t n,
                                                         int64_t* __restrict__ out) {
  const int lane = threadIdx.x;
  int64_t carry = 0;
  for (int64_t tile = 0; tile < n; tile += 4 * kWave) {
    const int64_t i0 = tile + 4 * lane;
    int64_t x[4];
#pragma unroll
    for (int k = 0; k < 4; k++) x[k] = i0 + k < n ? (int64_t)in[i0 + k] : 0;
    const int64_t mine = x[0] + x[1] + x[2] + x[3];
    int64_t inc = mine;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
      const int64_t y = __shfl_up(inc, d);
      if (lane >= d) inc += y;
    }
    int64_t off = carry + inc - mine;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (i0 + k < n) out[i0 + k] = off;
      off += x[k];
    }
    carry += __shfl(inc, kWave - 1);
  }
  if (lane == 0) out[n] = carry;
}

// ---- frame decode ----------------------------------------------------------------------------
constexpr int kDecThreads = 128;
constexpr uint32_t XXP1 = 2654435761u, XXP2 = 2246822519u, XXP3 = 3266489917u,
                   XXP4 = 668265263u, XXP5 = 374761393u;

template <int kCompCap>
struct __attribute__((aligned(16))) DecLds {
  uint8_t out[kMaxBlock + 64];
  uint8_t comp[kCompCap + 64];
  int progress;  // bytes of out[] final so far (decoder -> hasher)
  int error;
  uint32_t xxh;
};
constexpr int kSmallComp = 16384;  // frames that compress to <= 16 KiB: 48 KiB of LDS, 3 workgroups / CU

// wave-uniform unaligned 32-bit read from LDS into an SGPR
__device__ __forceinline__ uint32_t lds_u32_uniform(const uint8_t* base, int pos) {
  uint32_t v;
  __builtin_memcpy(&v, base + pos, 4);
  return __builtin_amdgcn_readfirstlane(v);
}

// LZ4 block decode of comp[0,clen) into out[0,olen), one wavefront, all control wave-uniform.
// Returns 0 or -1 (malformed).  Publishes progress for the trailing hasher.
//
// Fast path: one 16-byte window at ip holds token, <= 12 literals, the offset and the first
// match-length byte, so a sequence costs ONE LDS round trip; the copies are issued behind it and
// need no waits (LDS executes a wave's accesses in order, and the parse only reads comp[]).
__device__ int lz4_decode_wave(const uint8_t* comp, int clen, uint8_t* out, int olen,
                               volatile int* progress, int lane) {
  int ip = 0, op = 0, published = 0;
#ifdef S3S_LZ4_TIMING
  unsigned long long ddbg[8] = {0};
  struct Flush {
    unsigned long long* d;
    int lane;
    __device__ ~Flush() {
      if (lane == 0)
        for (int i = 0; i < 8; i++) atomicAdd(&g_dec_dbg[i], d[i]);
    }
  } flush{ddbg, lane};
  const unsigned long long tstart = __builtin_amdgcn_s_memtime();
#endif
  for (;;) {
    if (ip >= clen) return -1;
    DDBG_T(t0);
    DDBG_ADD(4, 1);
    int lit, ml, offset;
    bool fast = false;
    if (ip + 16 <= clen) {
      const uint32_t w0 = lds_u32_uniform(comp, ip), w1 = lds_u32_uniform(comp, ip + 4);
      const uint32_t w2 = lds_u32_uniform(comp, ip + 8), w3 = lds_u32_uniform(comp, ip + 12);
      lit = (int)((w0 >> 4) & 15u);
      ml = (int)(w0 & 15u);
      if (lit <= 12) {
        const int b = 1 + lit;  // byte index of the offset inside the window
        const int dw = b >> 2;
        const uint32_t lo = dw == 0 ? w0 : (dw == 1 ? w1 : (dw == 2 ? w2 : w3));
        const uint32_t hi = dw == 0 ? w1 : (dw == 1 ? w2 : (dw == 2 ? w3 : 0u));
        const uint32_t three = (uint32_t)((((uint64_t)hi << 32) | lo) >> (8 * (b & 3))) & 0xffffffu;
        offset = (int)(three & 0xffffu);
        const uint32_t ext = three >> 16;
        if (!(ml == 15 && ext == 255u)) {
          fast = true;
          if (lit > olen - op) return -1;
          if (lane < lit) out[op + lane] = comp[ip + 1 + lane];
          op += lit;
          ip += b + 2;
          if (ml == 15) {
            ml += (int)ext;
            ip += 1;
          }
        }
      }
    }
    if (!fast) {
      const uint32_t token = comp[ip++];
      lit = (int)(token >> 4);
      if (lit == 15) {
        uint32_t b;
        do {
          if (ip >= clen) return -1;
          b = comp[ip++];
          lit += (int)b;
        } while (b == 255);
      }
      if (lit > clen - ip || lit > olen - op) return -1;
      for (int j = lane; j < lit; j += kWave) out[op + j] = comp[ip + j];
      ip += lit;
      op += lit;
      if (ip == clen) break;  // last sequence carries literals only
      if (clen - ip < 2) return -1;
      offset = (int)comp[ip] | ((int)comp[ip + 1] << 8);
      ip += 2;
      ml = (int)(token & 15u);
      if (ml == 15) {
        uint32_t b;
        do {
          if (ip >= clen) return -1;
          b = comp[ip++];
          ml += (int)b;
        } while (b == 255);
      }
    }
    DDBG_T(t1);
    DDBG_ADD(0, t1 - t0);
    if (!fast) DDBG_ADD(5, 1);
    if (offset == 0 || offset > op) return -1;
    ml += 4;
    if (ml > olen - op) return -1;
    if (offset >= kWave) {
      // sources of a 64-byte round lie >= 64 bytes back: already written
      for (int j = lane; j < ml; j += kWave) out[op + j] = out[op - offset + j];
    } else {
      // overlapping copy = periodic pattern: every lane's source byte is round-invariant when
      // a round advances by a multiple of the period
      const int span = (kWave / offset) * offset;
      const uint32_t v = out[op - offset + (lane % offset)];
      if (lane < span)
        for (int j = lane; j < ml; j += span) out[op + j] = (uint8_t)v;
    }
    op += ml;
    DDBG_T(t2);
    DDBG_ADD(1, t2 - t1);
    if (op - published >= 512) {  // let the hasher trail in 512-byte steps
      published = op;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) *progress = op;
    }
  }
  if (op != olen) return -1;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if (lane == 0) *progress = olen;
#ifdef S3S_LZ4_TIMING
  ddbg[2] += __builtin_amdgcn_s_memtime() - tstart;
  ddbg[6] += 1;
#endif
  return 0;
}

// xxHash32 over out[0,olen) trailing the decoder: lanes 0..3 own the stripe accumulators
__device__ uint32_t xxh32_trailing(const uint8_t* out, int olen, volatile int* progress,
                                   volatile int* error, uint32_t seed, int lane) {
  uint32_t acc = lane == 0 ? seed + XXP1 + XXP2 : lane == 1 ? seed + XXP2 : lane == 2 ? seed : seed - XXP1;
  const int stripes = olen >> 4;
  int done = 0;
  const uint32_t* q = reinterpret_cast<const uint32_t*>(out) + (lane & 3);
  while (done < stripes) {
    int avail = *progress;
    if (*error) return 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    int upto = avail >> 4;
    if (upto > stripes) upto = stripes;
    if (upto == done) {
      __builtin_amdgcn_s_sleep(8);
      continue;
    }
    if (lane < 4)
      for (int j = done; j < upto; j++) acc = rotl32(acc + q[4 * j] * XXP2, 13) * XXP1;
    done = upto;
  }
  while (*progress < olen) {
    if (*error) return 0;
    __builtin_amdgcn_s_sleep(8);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  uint32_t h;
  if (olen >= 16) {
    const uint32_t v1 = __builtin_amdgcn_readlane(acc, 0), v2 = __builtin_amdgcn_readlane(acc, 1),
                   v3 = __builtin_amdgcn_readlane(acc, 2), v4 = __builtin_amdgcn_readlane(acc, 3);
    h = rotl32(v1, 1) + rotl32(v2, 7) + rotl32(v3, 12) + rotl32(v4, 18);
  } else {
    h = seed + XXP5;
  }
  h += (uint32_t)olen;
  int p = stripes << 4;
  for (; p + 4 <= olen; p += 4) h = rotl32(h + *reinterpret_cast<const uint32_t*>(out + p) * XXP3, 17) * XXP4;
  for (; p < olen; p++) h = rotl32(h + (uint32_t)out[p] * XXP5, 11) * XXP1;
  h ^= h >> 15;
  h *= XXP2;
  h ^= h >> 13;
  h *= XXP3;
  h ^= h >> 16;
  return h;
}

// Two instantiations share the frame list: kCompCap = kSmallComp takes the raw frames and the frames
// whose payload fits 16 KiB (3 workgroups per CU), kCompCap = kMaxBlock takes the rest.
template <int kCompCap>
__global__ __launch_bounds__(kDecThreads) void lz4_decompress_kernel(
    const uint8_t* __restrict__ comp, const Frame* __restrict__ frames, int32_t n_frames,
    const int64_t* __restrict__ frame_out, uint8_t* __restrict__ dst, int32_t* __restrict__ status) {
  __shared__ DecLds<kCompCap> s;
  const int f = blockIdx.x;
  if (f >= n_frames) return;
  const Frame fr = frames[f];
  const int olen = fr.orig_len, clen = fr.comp_len;
  if (olen == 0) return;  // end-of-stream frame: nothing to emit, decoding continues
  {
    const bool small = fr.method == 0x10 || clen <= kSmallComp;
    if (small != (kCompCap == kSmallComp)) return;  // the other instantiation's frame
  }
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (olen > kMaxBlock) {
    if (tid == 0) atomicExch(status, S3S_E_UNSUPPORTED);
    return;
  }
#ifdef S3S_LZ4_TIMING
  const unsigned long long kstart = __builtin_amdgcn_s_memtime();
#endif
  const uint8_t* g = comp + fr.comp_off;
  const bool raw = fr.method == 0x10;
  uint8_t* stage = raw ? s.out : s.comp;
  for (int i = tid * 16; i + 16 <= clen; i += kDecThreads * 16) {
    uint4 x;
    __builtin_memcpy(&x, g + i, 16);
    *reinterpret_cast<uint4*>(stage + i) = x;
  }
  for (int i = (clen & ~15) + tid; i < clen; i += kDecThreads) stage[i] = g[i];
  if (tid == 0) {
    s.progress = raw ? olen : 0;
    s.error = 0;
  }
  __syncthreads();
  if (wave == 0) {
    if (!raw) {
      const int rc = lz4_decode_wave(s.comp, clen, s.out, olen, &s.progress, lane);
      if (rc != 0 && lane == 0) {
        *(volatile int*)&s.error = 1;
        atomicExch(status, S3S_E_BAD_FRAME);
      }
    }
  } else {
    const uint32_t h = xxh32_trailing(s.out, olen, &s.progress, &s.error, kLz4BlockSeed, lane);
    if (lane == 0) s.xxh = h;
  }
  __syncthreads();
  if (s.error) return;
  if (tid == 0 && (s.xxh & 0x0FFFFFFFu) != fr.check) atomicExch(status, S3S_E_BAD_FRAME);
  // stream the decoded chunk out, 16-byte stores on the destination's alignment
  uint8_t* d = dst + frame_out[f];
  int head = (int)((16u - (uint32_t)(uintptr_t)d) & 15u);
  head = head < olen ? head : olen;
  if (tid < head) d[tid] = s.out[tid];
  const int nvec = (olen - head) >> 4;
  for (int v = tid; v < nvec; v += kDecThreads) {
    uint4 x;
    __builtin_memcpy(&x, s.out + head + 16 * v, 16);
    *reinterpret_cast<uint4*>(d + head + 16 * v) = x;
  }
  const int done = head + 16 * nvec;
  if (tid < olen - done) d[done + tid] = s.out[done + tid];
#ifdef S3S_LZ4_TIMING
  if (tid == 0) atomicAdd(&g_dec_dbg[3], __builtin_amdgcn_s_memtime() - kstart);
#endif
}


// ---- frame decode, variant 1: straight to global memory, no LDS -------------------------------
// One wavefront per frame and NOTHING in LDS, so residency is bounded by registers only (the
// LDS-staged kernel above fits 3 frames per CU; this one fits ~32).  The decode chain of one
// frame is as serial as ever — the throughput comes from the number of frames in flight.
//   parse     scalar loads (s_load: lgkmcnt, independent of the vector-memory queue) fetch a
//             20-byte window of the compressed stream per sequence
//   literals  comp -> out, plain byte loads / stores, fire and forget
//   matches   out -> out.  The source may have been written moments ago by this very wave, so
//             (a) loads are agent-scope relaxed atomics (sc1: served by L2, never by a stale L1
//             line) and (b) the wave drains its store queue (s_waitcnt vmcnt(0)) first — but only
//             when the source range reaches past the last drain point, which for shuffle data
//             (offsets of hundreds of bytes) is once every few dozen sequences.
//   check     xxHash32 re-reads the finished frame from L2 (lanes 0..3 own the accumulators)
__device__ __forceinline__ uint32_t ld_u8_l2(const uint8_t* p) {
  return (uint32_t)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t ld_u32_l2(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// 16 bytes at (arbitrarily aligned) address a, through the scalar cache: w[0..3]
__device__ __forceinline__ void sload16(const uint8_t* a, uint32_t& w0, uint32_t& w1, uint32_t& w2,
                                        uint32_t& w3) {
  const uint64_t addr = reinterpret_cast<uint64_t>(a);
  const uint32_t* base = reinterpret_cast<const uint32_t*>(addr & ~uint64_t(3));
  const uint32_t sh = (uint32_t)(addr & 3u) * 8u;
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  u32x4 q;
  uint32_t d4;
  asm volatile(
      "s_load_dwordx4 %0, %2, 0x0\n\t"
      "s_load_dword %1, %2, 0x10\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&s"(q), "=&s"(d4)
      : "s"(base)
      : "memory");
  const uint32_t d0 = q.x, d1 = q.y, d2 = q.z, d3 = q.w;
  w0 = (uint32_t)((((uint64_t)d1 << 32) | d0) >> sh);
  w1 = (uint32_t)((((uint64_t)d2 << 32) | d1) >> sh);
  w2 = (uint32_t)((((uint64_t)d3 << 32) | d2) >> sh);
  w3 = (uint32_t)((((uint64_t)d4 << 32) | d3) >> sh);
}

__global__ __launch_bounds__(kWave) void lz4_decompress_global_kernel(
    const uint8_t* __restrict__ comp, const Frame* __restrict__ frames, int32_t n_frames,
    const int64_t* __restrict__ frame_out, uint8_t* dst, int32_t* __restrict__ status) {
  const int f = blockIdx.x;
  if (f >= n_frames) return;
  const Frame fr = frames[f];
  const int olen = fr.orig_len, clen = fr.comp_len;
  if (olen == 0) return;
  const int lane = threadIdx.x;
  const uint8_t* c = comp + fr.comp_off;
  uint8_t* out = dst + frame_out[f];
  bool bad = false;
#ifdef S3S_LZ4_TIMING
  const unsigned long long gk0 = __builtin_amdgcn_s_memtime();
  unsigned long long g_seq = 0, g_flush = 0, g_drain = 0, g_slow = 0;
#endif
  if (fr.method == 0x10) {  // stored frame
    for (int j = lane * 4; j < olen; j += kWave * 4) {
      if (j + 4 <= olen) {
        uint32_t x;
        __builtin_memcpy(&x, c + j, 4);
        __builtin_memcpy(out + j, &x, 4);
      } else {
        for (int k = j; k < olen; k++) out[k] = c[k];
      }
    }
  } else {
    int ip = 0, op = 0, drained = 0;
    // 256-byte window of the compressed stream held across the wave: lane i owns the dword at WB+4i.
    // Parsing a sequence is then 5 v_readlane + scalar shifts, its literals one ds_bpermute — no
    // memory round trip.  The window is reloaded (one coalesced load) every ~240 stream bytes.
    const uint64_t c_addr = reinterpret_cast<uint64_t>(c);
    const uint64_t last_dw = (c_addr + (uint64_t)clen - 1u) & ~uint64_t(3);  // last dword holding stream bytes
    uint64_t WB = 0;
    uint32_t win = 0;
    bool have_win = false;
    // up to 4 match copies ride in registers between their load and their store, so the wave pays one
    // L2 round trip per 4 sequences instead of one per sequence.  A copy may join the flight only if
    // its source lies entirely below `drained` (= everything below is complete in L2).
    int npend = 0, pop0 = 0, pop1 = 0, pop2 = 0, pop3 = 0, pml0 = 0, pml1 = 0, pml2 = 0, pml3 = 0;
    uint32_t pv0 = 0, pv1 = 0, pv2 = 0, pv3 = 0;
    auto flush = [&]() {
      if (npend > 0) {
#ifdef S3S_LZ4_TIMING
        g_flush++;
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // loads landed; every earlier store is complete
        drained = pop0;
#ifndef S3S_ABL_DEC_NOSTORE
        if (lane < pml0) out[pop0 + lane] = (uint8_t)pv0;
        if (npend > 1 && lane < pml1) out[pop1 + lane] = (uint8_t)pv1;
        if (npend > 2 && lane < pml2) out[pop2 + lane] = (uint8_t)pv2;
        if (npend > 3 && lane < pml3) out[pop3 + lane] = (uint8_t)pv3;
#else
        asm volatile("" ::"v"(pv0), "v"(pv1), "v"(pv2), "v"(pv3));
#endif
        npend = 0;
      }
    };
    for (;;) {
      if (ip >= clen) { bad = true; break; }
#ifdef S3S_LZ4_TIMING
      g_seq++;
#endif
      int lit, ml, offset;
      bool fast = false;
      if (ip + 20 <= clen) {
        const uint64_t a = c_addr + (uint64_t)ip;
        if (!have_win || a + 20u > WB + 256u) {
          WB = a & ~uint64_t(3);
          uint64_t la = WB + 4u * (uint64_t)lane;
          la = la < last_dw ? la : last_dw;
          win = *reinterpret_cast<const uint32_t*>(la);
          have_win = true;
        }
        const int k = (int)((a - WB) >> 2);
        const uint32_t sh = (uint32_t)(a & 3u) * 8u;
        const uint32_t d0 = __builtin_amdgcn_readlane(win, k), d1 = __builtin_amdgcn_readlane(win, k + 1);
        const uint32_t d2 = __builtin_amdgcn_readlane(win, k + 2), d3 = __builtin_amdgcn_readlane(win, k + 3);
        const uint32_t d4 = __builtin_amdgcn_readlane(win, k + 4);
        const uint32_t w0 = (uint32_t)((((uint64_t)d1 << 32) | d0) >> sh);
        const uint32_t w1 = (uint32_t)((((uint64_t)d2 << 32) | d1) >> sh);
        const uint32_t w2 = (uint32_t)((((uint64_t)d3 << 32) | d2) >> sh);
        const uint32_t w3 = (uint32_t)((((uint64_t)d4 << 32) | d3) >> sh);
        lit = (int)((w0 >> 4) & 15u);
        ml = (int)(w0 & 15u);
        if (lit <= 12) {
          const int b = 1 + lit;
          const int dw = b >> 2;
          const uint32_t lo = dw == 0 ? w0 : (dw == 1 ? w1 : (dw == 2 ? w2 : w3));
          const uint32_t hi = dw == 0 ? w1 : (dw == 1 ? w2 : (dw == 2 ? w3 : 0u));
          const uint32_t three = (uint32_t)((((uint64_t)hi << 32) | lo) >> (8 * (b & 3))) & 0xffffffu;
          offset = (int)(three & 0xffffu);
          const uint32_t ext = three >> 16;
          if (!(ml == 15 && ext == 255u)) {
            fast = true;
            if (lit > olen - op) { bad = true; break; }
            if (lit > 0) {  // literal j is stream byte ip+1+j, i.e. a byte of the window
              const uint32_t la = (uint32_t)(a - WB) + 1u + (uint32_t)lane;
              const uint32_t dwv = (uint32_t)__shfl((int)win, (int)((la >> 2) & 63u));
#ifndef S3S_ABL_DEC_NOSTORE
              if (lane < lit) out[op + lane] = (uint8_t)(dwv >> (8u * (la & 3u)));
#else
              asm volatile("" ::"v"(dwv));
#endif
            }
            op += lit;
            ip += b + 2;
            if (ml == 15) {
              ml += (int)ext;
              ip += 1;
            }
          }
        }
      }
      if (!fast) {
        // byte-wise parse (long literal runs, long matches, the tail of the frame): wave-uniform
        // vector loads, rare enough not to matter
        flush();
        const uint32_t token = __builtin_amdgcn_readfirstlane((uint32_t)c[ip]);
        ip++;
        lit = (int)(token >> 4);
        if (lit == 15) {
          uint32_t b;
          do {
            if (ip >= clen) { bad = true; break; }
            b = __builtin_amdgcn_readfirstlane((uint32_t)c[ip]);
            ip++;
            lit += (int)b;
          } while (b == 255);
          if (bad) break;
        }
        if (lit > clen - ip || lit > olen - op) { bad = true; break; }
        for (int j = lane; j < lit; j += kWave) out[op + j] = c[ip + j];
        ip += lit;
        op += lit;
        if (ip == clen) break;  // last sequence carries literals only
        if (clen - ip < 2) { bad = true; break; }
        offset = (int)__builtin_amdgcn_readfirstlane((uint32_t)c[ip] | ((uint32_t)c[ip + 1] << 8));
        ip += 2;
        ml = (int)(token & 15u);
        if (ml == 15) {
          uint32_t b;
          do {
            if (ip >= clen) { bad = true; break; }
            b = __builtin_amdgcn_readfirstlane((uint32_t)c[ip]);
            ip++;
            ml += (int)b;
          } while (b == 255);
          if (bad) break;
        }
      }
      if (offset == 0 || offset > op) { bad = true; break; }
      ml += 4;
      if (ml > olen - op) { bad = true; break; }
      const uint8_t* src = out + op - offset;
      if (offset >= kWave && ml <= kWave) {
        // the common copy: one round, source at least a wave's width back
        const int src_end = op - offset + ml;
        if (src_end > drained) {
          flush();
          if (src_end > drained) {
#ifdef S3S_LZ4_TIMING
            g_drain++;
#endif
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            drained = op;
          }
        }
        if (npend == 4) flush();
        uint32_t val = 0;
        if (lane < ml) val = ld_u8_l2(src + lane);
        if (npend == 0) { pv0 = val; pop0 = op; pml0 = ml; }
        else if (npend == 1) { pv1 = val; pop1 = op; pml1 = ml; }
        else if (npend == 2) { pv2 = val; pop2 = op; pml2 = ml; }
        else { pv3 = val; pop3 = op; pml3 = ml; }
        npend++;
      } else if (offset >= kWave) {
        // long copy, 64 bytes per round; a round's source must be complete in L2 before it is read
        // (for offset < length that includes this copy's own earlier rounds)
        flush();
        for (int j0 = 0; j0 < ml; j0 += kWave) {
          const int need = op - offset + (ml - j0 < kWave ? ml : j0 + kWave);
          if (need > drained) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            drained = op + j0;
          }
          const int j = j0 + lane;
          if (j < ml) out[op + j] = (uint8_t)ld_u8_l2(src + j);
        }
      } else {
        // overlapping copy = periodic pattern of the `offset` bytes in front of op
        flush();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        drained = op;
        const int span = (kWave / offset) * offset;
        const uint32_t v = ld_u8_l2(src + (lane % offset));
        if (lane < span)
          for (int j = lane; j < ml; j += span) out[op + j] = (uint8_t)v;
      }
      op += ml;
    }
    flush();
    if (!bad && op != olen) bad = true;
  }
  if (bad) {
    if (lane == 0) atomicExch(status, S3S_E_BAD_FRAME);
    return;
  }
  // ---- LZ4Block check: xxHash32(seed 0x9747b28c) & 0x0FFFFFFF over the decoded frame ------------
  // all stores complete in L2, then drop this CU's (possibly stale) L1 lines: plain loads below
  // are served with what L2 holds.  The frame is streamed 256 bytes per coalesced load; the four
  // stripe accumulators live in lanes 0..3 and pull their words out of the block by cross-lane reads.
#ifdef S3S_LZ4_TIMING
  const unsigned long long gk1 = __builtin_amdgcn_s_memtime();
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  {
    const uint32_t seed = kLz4BlockSeed;
    uint32_t acc = lane == 0 ? seed + XXP1 + XXP2 : lane == 1 ? seed + XXP2 : lane == 2 ? seed : seed - XXP1;
    const int stripes = olen >> 4;
    const int nblk = olen >> 8;
    auto ld32u = [&](int byte_pos) -> uint32_t {
      uint32_t x;
      __builtin_memcpy(&x, out + byte_pos, 4);  // unaligned global_load_dword
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
    for (int j = nblk * 16; j < stripes; j++) {  // < 16 leftover stripes
      const uint32_t wv = ld32u(16 * j + 4 * (lane & 3));
      acc = rotl32(acc + wv * XXP2, 13) * XXP1;
    }
    uint32_t h;
    if (olen >= 16) {
      const uint32_t v1 = __builtin_amdgcn_readlane(acc, 0), v2 = __builtin_amdgcn_readlane(acc, 1),
                     v3 = __builtin_amdgcn_readlane(acc, 2), v4 = __builtin_amdgcn_readlane(acc, 3);
      h = rotl32(v1, 1) + rotl32(v2, 7) + rotl32(v3, 12) + rotl32(v4, 18);
    } else {
      h = seed + XXP5;
    }
    h += (uint32_t)olen;
    int p = stripes << 4;
    for (; p + 4 <= olen; p += 4) h = rotl32(h + ld32u(p) * XXP3, 17) * XXP4;
    for (; p < olen; p++) h = rotl32(h + (uint32_t)out[p] * XXP5, 11) * XXP1;
    h ^= h >> 15;
    h *= XXP2;
    h ^= h >> 13;
    h *= XXP3;
    h ^= h >> 16;
    if (lane == 0 && (h & 0x0FFFFFFFu) != fr.check) atomicExch(status, S3S_E_BAD_FRAME);
  }
#ifdef S3S_LZ4_TIMING
  if (lane == 0) {
    const unsigned long long gk2 = __builtin_amdgcn_s_memtime();
    atomicAdd(&g_dec_dbg[8], gk1 - gk0);
    atomicAdd(&g_dec_dbg[9], gk2 - gk1);
    atomicAdd(&g_dec_dbg[10], 1ull);
    atomicAdd(&g_dec_dbg[11], g_seq);
    atomicAdd(&g_dec_dbg[12], g_flush);
    atomicAdd(&g_dec_dbg[13], g_drain);
  }
#endif
}

// ---- frame decode, variant 2: variant 1 plus an LDS ring of the last 8 KiB of output -------------
// Near matches (the bulk for shuffle data) are LDS -> LDS and never wait for memory; only far
// matches take the L2 round trip of variant 1.  8 KiB per frame keeps 20 frames per CU resident.
constexpr int kRing = 8192;
__global__ __launch_bounds__(kWave) void lz4_decompress_ring_kernel(
    const uint8_t* __restrict__ comp, const Frame* __restrict__ frames, int32_t n_frames,
    const int64_t* __restrict__ frame_out, uint8_t* dst, int32_t* __restrict__ status) {
  const int f = blockIdx.x;
  if (f >= n_frames) return;
  const Frame fr = frames[f];
  const int olen = fr.orig_len, clen = fr.comp_len;
  if (olen == 0) return;
  const int lane = threadIdx.x;
  // the last kRing output bytes of the frame, so that near matches never leave the CU
  __shared__ uint8_t ring[kRing];
  const uint8_t* c = comp + fr.comp_off;
  uint8_t* out = dst + frame_out[f];
  bool bad = false;
  if (fr.method == 0x10) {  // stored frame
    for (int j = lane * 4; j < olen; j += kWave * 4) {
      if (j + 4 <= olen) {
        uint32_t x;
        __builtin_memcpy(&x, c + j, 4);
        __builtin_memcpy(out + j, &x, 4);
      } else {
        for (int k = j; k < olen; k++) out[k] = c[k];
      }
    }
  } else {
    int ip = 0, op = 0, drained = 0;
    // 256-byte window of the compressed stream held across the wave: lane i owns the dword at WB+4i.
    // Parsing a sequence is then 5 v_readlane + scalar shifts, its literals one ds_bpermute — no
    // memory round trip.  The window is reloaded (one coalesced load) every ~240 stream bytes.
    const uint64_t c_addr = reinterpret_cast<uint64_t>(c);
    const uint64_t last_dw = (c_addr + (uint64_t)clen - 1u) & ~uint64_t(3);  // last dword holding stream bytes
    uint64_t WB = 0;
    uint32_t win = 0;
    bool have_win = false;
    // up to 4 match copies ride in registers between their load and their store, so the wave pays one
    // L2 round trip per 4 sequences instead of one per sequence.  A copy may join the flight only if
    // its source lies entirely below `drained` (= everything below is complete in L2).
    int npend = 0, pop0 = 0, pop1 = 0, pop2 = 0, pop3 = 0, pml0 = 0, pml1 = 0, pml2 = 0, pml3 = 0;
    uint32_t pv0 = 0, pv1 = 0, pv2 = 0, pv3 = 0;
    auto flush = [&]() {
      if (npend > 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // loads landed; every earlier store is complete
        drained = pop0;
        if (lane < pml0) { out[pop0 + lane] = (uint8_t)pv0; ring[(pop0 + lane) & (kRing - 1)] = (uint8_t)pv0; }
        if (npend > 1 && lane < pml1) { out[pop1 + lane] = (uint8_t)pv1; ring[(pop1 + lane) & (kRing - 1)] = (uint8_t)pv1; }
        if (npend > 2 && lane < pml2) { out[pop2 + lane] = (uint8_t)pv2; ring[(pop2 + lane) & (kRing - 1)] = (uint8_t)pv2; }
        if (npend > 3 && lane < pml3) { out[pop3 + lane] = (uint8_t)pv3; ring[(pop3 + lane) & (kRing - 1)] = (uint8_t)pv3; }
        npend = 0;
      }
    };
    for (;;) {
      if (ip >= clen) { bad = true; break; }
      // a copy in flight owns ring slots it has not written yet: land it before the ring wraps onto them
      if (npend > 0 && op - pop0 > kRing / 2) flush();
      int lit, ml, offset;
      bool fast = false;
      if (ip + 20 <= clen) {
        const uint64_t a = c_addr + (uint64_t)ip;
        if (!have_win || a + 20u > WB + 256u) {
          WB = a & ~uint64_t(3);
          uint64_t la = WB + 4u * (uint64_t)lane;
          la = la < last_dw ? la : last_dw;
          win = *reinterpret_cast<const uint32_t*>(la);
          have_win = true;
        }
        const int k = (int)((a - WB) >> 2);
        const uint32_t sh = (uint32_t)(a & 3u) * 8u;
        const uint32_t d0 = __builtin_amdgcn_readlane(win, k), d1 = __builtin_amdgcn_readlane(win, k + 1);
        const uint32_t d2 = __builtin_amdgcn_readlane(win, k + 2), d3 = __builtin_amdgcn_readlane(win, k + 3);
        const uint32_t d4 = __builtin_amdgcn_readlane(win, k + 4);
        const uint32_t w0 = (uint32_t)((((uint64_t)d1 << 32) | d0) >> sh);
        const uint32_t w1 = (uint32_t)((((uint64_t)d2 << 32) | d1) >> sh);
        const uint32_t w2 = (uint32_t)((((uint64_t)d3 << 32) | d2) >> sh);
        const uint32_t w3 = (uint32_t)((((uint64_t)d4 << 32) | d3) >> sh);
        lit = (int)((w0 >> 4) & 15u);
        ml = (int)(w0 & 15u);
        if (lit <= 12) {
          const int b = 1 + lit;
          const int dw = b >> 2;
          const uint32_t lo = dw == 0 ? w0 : (dw == 1 ? w1 : (dw == 2 ? w2 : w3));
          const uint32_t hi = dw == 0 ? w1 : (dw == 1 ? w2 : (dw == 2 ? w3 : 0u));
          const uint32_t three = (uint32_t)((((uint64_t)hi << 32) | lo) >> (8 * (b & 3))) & 0xffffffu;
          offset = (int)(three & 0xffffu);
          const uint32_t ext = three >> 16;
          if (!(ml == 15 && ext == 255u)) {
            fast = true;
            if (lit > olen - op) { bad = true; break; }
            if (lit > 0) {  // literal j is stream byte ip+1+j, i.e. a byte of the window
              const uint32_t la = (uint32_t)(a - WB) + 1u + (uint32_t)lane;
              const uint32_t dwv = (uint32_t)__shfl((int)win, (int)((la >> 2) & 63u));
              if (lane < lit) {
                const uint8_t bv = (uint8_t)(dwv >> (8u * (la & 3u)));
                out[op + lane] = bv;
                ring[(op + lane) & (kRing - 1)] = bv;
              }
            }
            op += lit;
            ip += b + 2;
            if (ml == 15) {
              ml += (int)ext;
              ip += 1;
            }
          }
        }
      }
      if (!fast) {
        // byte-wise parse (long literal runs, long matches, the tail of the frame): wave-uniform
        // vector loads, rare enough not to matter
        flush();
        const uint32_t token = __builtin_amdgcn_readfirstlane((uint32_t)c[ip]);
        ip++;
        lit = (int)(token >> 4);
        if (lit == 15) {
          uint32_t b;
          do {
            if (ip >= clen) { bad = true; break; }
            b = __builtin_amdgcn_readfirstlane((uint32_t)c[ip]);
            ip++;
            lit += (int)b;
          } while (b == 255);
          if (bad) break;
        }
        if (lit > clen - ip || lit > olen - op) { bad = true; break; }
        for (int j = lane; j < lit; j += kWave) {
          const uint8_t bv = c[ip + j];
          out[op + j] = bv;
          ring[(op + j) & (kRing - 1)] = bv;
        }
        ip += lit;
        op += lit;
        if (ip == clen) break;  // last sequence carries literals only
        if (clen - ip < 2) { bad = true; break; }
        offset = (int)__builtin_amdgcn_readfirstlane((uint32_t)c[ip] | ((uint32_t)c[ip + 1] << 8));
        ip += 2;
        ml = (int)(token & 15u);
        if (ml == 15) {
          uint32_t b;
          do {
            if (ip >= clen) { bad = true; break; }
            b = __builtin_amdgcn_readfirstlane((uint32_t)c[ip]);
            ip++;
            ml += (int)b;
          } while (b == 255);
          if (bad) break;
        }
      }
      if (offset == 0 || offset > op) { bad = true; break; }
      ml += 4;
      if (ml > olen - op) { bad = true; break; }
      const uint8_t* src = out + op - offset;
      if (offset + kWave <= kRing) {
        // ---- near match: the source is in the ring (LDS executes this wave's accesses in order, so
        // rounds of a long copy and overlapping copies need no waits)
        if (npend > 0 && op - offset + ml > pop0) flush();  // ... unless it is still in flight
        if (offset >= kWave) {
          for (int j0 = 0; j0 < ml; j0 += kWave) {
            const int j = j0 + lane;
            if (j < ml) {
              const uint8_t bv = ring[(op - offset + j) & (kRing - 1)];
              ring[(op + j) & (kRing - 1)] = bv;
              out[op + j] = bv;
            }
          }
        } else {  // overlap: periodic in the last `offset` bytes; one register byte per lane
          const int span = (kWave / offset) * offset;
          const uint8_t bv = ring[(op - offset + (lane % offset)) & (kRing - 1)];
          if (lane < span)
            for (int j = lane; j < ml; j += span) {
              ring[(op + j) & (kRing - 1)] = bv;
              out[op + j] = bv;
            }
        }
      } else if (ml <= kWave) {
        // ---- far match, one round: rides in registers between its L2 load and its stores
        const int src_end = op - offset + ml;
        if (src_end > drained) {
          flush();
          if (src_end > drained) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            drained = op;
          }
        }
        if (npend == 4) flush();
        uint32_t val = 0;
        if (lane < ml) val = ld_u8_l2(src + lane);
        if (npend == 0) { pv0 = val; pop0 = op; pml0 = ml; }
        else if (npend == 1) { pv1 = val; pop1 = op; pml1 = ml; }
        else if (npend == 2) { pv2 = val; pop2 = op; pml2 = ml; }
        else { pv3 = val; pop3 = op; pml3 = ml; }
        npend++;
      } else {
        // ---- far match, several rounds (offset > kRing - 64 >= 64: rounds never overlap their source)
        flush();
        for (int j0 = 0; j0 < ml; j0 += kWave) {
          const int need = op - offset + (ml - j0 < kWave ? ml : j0 + kWave);
          if (need > drained) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            drained = op + j0;
          }
          const int j = j0 + lane;
          if (j < ml) {
            const uint8_t bv = (uint8_t)ld_u8_l2(src + j);
            out[op + j] = bv;
            ring[(op + j) & (kRing - 1)] = bv;
          }
        }
      }
      op += ml;
    }
    flush();
    if (!bad && op != olen) bad = true;
  }
  if (bad) {
    if (lane == 0) atomicExch(status, S3S_E_BAD_FRAME);
    return;
  }
  // ---- LZ4Block check: xxHash32(seed 0x9747b28c) & 0x0FFFFFFF over the decoded frame ------------
  // all stores complete in L2, then drop this CU's (possibly stale) L1 lines: plain loads below
  // are served with what L2 holds.  The frame is streamed 256 bytes per coalesced load; the four
  // stripe accumulators live in lanes 0..3 and pull their words out of the block by cross-lane reads.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  {
    const uint32_t seed = kLz4BlockSeed;
    uint32_t acc = lane == 0 ? seed + XXP1 + XXP2 : lane == 1 ? seed + XXP2 : lane == 2 ? seed : seed - XXP1;
    const int stripes = olen >> 4;
    const int nblk = olen >> 8;
    auto ld32u = [&](int byte_pos) -> uint32_t {
      uint32_t x;
      __builtin_memcpy(&x, out + byte_pos, 4);  // unaligned global_load_dword
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
    for (int j = nblk * 16; j < stripes; j++) {  // < 16 leftover stripes
      const uint32_t wv = ld32u(16 * j + 4 * (lane & 3));
      acc = rotl32(acc + wv * XXP2, 13) * XXP1;
    }
    uint32_t h;
    if (olen >= 16) {
      const uint32_t v1 = __builtin_amdgcn_readlane(acc, 0), v2 = __builtin_amdgcn_readlane(acc, 1),
                     v3 = __builtin_amdgcn_readlane(acc, 2), v4 = __builtin_amdgcn_readlane(acc, 3);
      h = rotl32(v1, 1) + rotl32(v2, 7) + rotl32(v3, 12) + rotl32(v4, 18);
    } else {
      h = seed + XXP5;
    }
    h += (uint32_t)olen;
    int p = stripes << 4;
    for (; p + 4 <= olen; p += 4) h = rotl32(h + ld32u(p) * XXP3, 17) * XXP4;
    for (; p < olen; p++) h = rotl32(h + (uint32_t)out[p] * XXP5, 11) * XXP1;
    h ^= h >> 15;
    h *= XXP2;
    h ^= h >> 13;
    h *= XXP3;
    h ^= h >> 16;
    if (lane == 0 && (h & 0x0FFFFFFFu) != fr.check) atomicExch(status, S3S_E_BAD_FRAME);
  }
}

// ---- frame decode, variant 3: ring decoder written for the VALU ---------------------------------
// PMC on variants 1/2 shows ~160 scalar instructions per LZ4 sequence and the CU's single scalar
// ALU ~75 % busy: wave-serial decoders on CDNA are bounded by SALU issue, not by memory.  Here
// every wave-uniform quantity of the parse (token fields, offset, positions) is computed
// redundantly by all 64 lanes on the vector ALU — values come out of ds_bpermute, so the compiler
// keeps them in VGPRs — and only branch conditions are turned scalar (v_readfirstlane).  Far
// matches (source older than the ring) are rare and handled synchronously.
__global__ __launch_bounds__(kWave) void lz4_decompress_valu_kernel(
    const uint8_t* __restrict__ comp, const Frame* __restrict__ frames, int32_t n_frames,
    const int64_t* __restrict__ frame_out, uint8_t* dst, int32_t* __restrict__ status) {
  __shared__ uint8_t ring[kRing];
  const int f = blockIdx.x;
  if (f >= n_frames) return;
  const Frame fr = frames[f];
  const int olen = fr.orig_len, clen = fr.comp_len;
  if (olen == 0) return;
  const int lane = threadIdx.x;
  const uint8_t* c = comp + fr.comp_off;
  uint8_t* out = dst + frame_out[f];
  int bad = 0;
  if (fr.method == 0x10) {  // stored frame
    for (int j = lane * 4; j < olen; j += kWave * 4) {
      if (j + 4 <= olen) {
        uint32_t x;
        __builtin_memcpy(&x, c + j, 4);
        __builtin_memcpy(out + j, &x, 4);
      } else {
        for (int k = j; k < olen; k++) out[k] = c[k];
      }
    }
  } else {
    // per-lane copies of the wave-uniform state (kept in VGPRs on purpose)
    int ip = 0, op = 0;
    asm volatile("" : "+v"(ip), "+v"(op));
    int drained = 0;  // scalar: only the rare far path uses it
    const uint32_t c_lo = (uint32_t)(reinterpret_cast<uint64_t>(c) & 3u);  // stream byte s sits at dword-space byte s + c_lo
    const uint8_t* c_al = c - c_lo;                                        // dword-aligned base of the stream
    const int last_dw = (int)((c_lo + (uint32_t)clen - 1u) >> 2);          // last dword index holding stream bytes
    int wb = -1 << 20;   // dword index (relative to c_al) of window lane 0; VGPR-uniform
    asm volatile("" : "+v"(wb));
    uint32_t win = 0;
#ifdef S3S_LZ4_TIMING
    unsigned long long ddbg[8] = {0};
    const unsigned long long tk0 = __builtin_amdgcn_s_memtime();
#endif
    for (;;) {
      DDBG_T(tv0);
      DDBG_ADD(4, 1);
      // ---- loop control: everything scalar the loop needs, from one readfirstlane -----------------
      const int ipu = __builtin_amdgcn_readfirstlane(ip);
      if (ipu >= clen) { bad = 1; break; }
      int lit, ml, offset;
      bool fast = false;
      if (ipu + 20 <= clen) {
        const int ab = ip + (int)c_lo;        // byte position in dword space
        int k = (ab >> 2) - wb;               // window lane of the token's dword
        if (__builtin_amdgcn_readfirstlane((int)(k < 0 || k > 58))) {  // (re)load the 256-byte window
          wb = ab >> 2;
          int di = wb + lane;
          di = di < last_dw ? di : last_dw;
          win = reinterpret_cast<const uint32_t*>(c_al)[di];
          k = 0;
          DDBG_ADD(7, 1);
        }
        const uint32_t d0 = (uint32_t)__shfl((int)win, k), d1 = (uint32_t)__shfl((int)win, k + 1);
        const uint32_t d2 = (uint32_t)__shfl((int)win, k + 2), d3 = (uint32_t)__shfl((int)win, k + 3);
        const uint32_t d4 = (uint32_t)__shfl((int)win, k + 4);
        const uint32_t sh = ((uint32_t)ab & 3u) * 8u;
        const uint32_t w0 = __builtin_amdgcn_alignbit(d1, d0, sh), w1 = __builtin_amdgcn_alignbit(d2, d1, sh);
        const uint32_t w2 = __builtin_amdgcn_alignbit(d3, d2, sh), w3 = __builtin_amdgcn_alignbit(d4, d3, sh);
        lit = (int)((w0 >> 4) & 15u);
        ml = (int)(w0 & 15u);
        const int b = 1 + lit;  // byte index of the offset inside the 16-byte view (valid for lit <= 12)
        const int dw = b >> 2;
        const uint32_t lo = dw == 0 ? w0 : (dw == 1 ? w1 : (dw == 2 ? w2 : w3));
        const uint32_t hi = dw == 0 ? w1 : (dw == 1 ? w2 : (dw == 2 ? w3 : 0u));
        const uint32_t three = __builtin_amdgcn_alignbit(hi, lo, 8u * ((uint32_t)b & 3u)) & 0xffffffu;
        offset = (int)(three & 0xffffu);
        const uint32_t ext = three >> 16;
        const int okv = (lit <= 12) & !((ml == 15) & (ext == 255u)) & (lit <= olen - op);
        if (__builtin_amdgcn_readfirstlane(okv)) {
          fast = true;
          // literal j is stream byte ip+1+j: a byte of the window
          const uint32_t la = (uint32_t)(ab - 4 * wb) + 1u + (uint32_t)lane;
          const uint32_t dwv = (uint32_t)__shfl((int)win, (int)((la >> 2) & 63u));
          if (lane < lit) {
            const uint8_t bv = (uint8_t)(dwv >> (8u * (la & 3u)));
            out[op + lane] = bv;
            ring[(op + lane) & (kRing - 1)] = bv;
          }
          op += lit;
          const int m15 = (ml == 15) ? 1 : 0;
          ml += m15 ? (int)ext : 0;
          ip += b + 2 + m15;
        }
      }
      if (!fast) {
        DDBG_ADD(5, 1);
        // byte-wise parse (long literal runs, long matches, the tail of the frame): rare
        int ips = ipu, ops = __builtin_amdgcn_readfirstlane(op);
        const uint32_t token = __builtin_amdgcn_readfirstlane((uint32_t)c[ips]);
        ips++;
        lit = (int)(token >> 4);
        if (lit == 15) {
          uint32_t bb;
          do {
            if (ips >= clen) { bad = 1; break; }
            bb = __builtin_amdgcn_readfirstlane((uint32_t)c[ips]);
            ips++;
            lit += (int)bb;
          } while (bb == 255);
          if (bad) break;
        }
        if (lit > clen - ips || lit > olen - ops) { bad = 1; break; }
        for (int j = lane; j < lit; j += kWave) {
          const uint8_t bv = c[ips + j];
          out[ops + j] = bv;
          ring[(ops + j) & (kRing - 1)] = bv;
        }
        ips += lit;
        ops += lit;
        ip = ips;
        op = ops;
        if (ips == clen) break;  // last sequence carries literals only
        if (clen - ips < 2) { bad = 1; break; }
        offset = (int)__builtin_amdgcn_readfirstlane((uint32_t)c[ips] | ((uint32_t)c[ips + 1] << 8));
        ips += 2;
        ml = (int)(token & 15u);
        if (ml == 15) {
          uint32_t bb;
          do {
            if (ips >= clen) { bad = 1; break; }
            bb = __builtin_amdgcn_readfirstlane((uint32_t)c[ips]);
            ips++;
            ml += (int)bb;
          } while (bb == 255);
          if (bad) break;
        }
        ip = ips;
      }
      ml += 4;
      DDBG_T(tv1);
      DDBG_ADD(0, tv1 - tv0);
      // one scalar decision word: bit0 malformed, bit1 near (source in the ring), bit2 single round
      const int near = (offset + kWave <= kRing);
      const int dec = ((offset == 0) | (offset > op) | (ml > olen - op)) | (near << 1) | ((ml <= kWave) << 2) |
                      ((offset >= kWave) << 3);
      const int decu = __builtin_amdgcn_readfirstlane(dec);
      if (decu & 1) { bad = 1; break; }
      if ((decu & 14) == 14) {
        // near, one round, no overlap (the common copy): LDS -> LDS + global
        if (lane < ml) {
          const uint8_t bv = ring[(op - offset + lane) & (kRing - 1)];
          ring[(op + lane) & (kRing - 1)] = bv;
          out[op + lane] = bv;
        }
      } else if ((decu & 6) == 6) {
        // near, one round, overlapping: byte j of the copy is source byte j mod offset
        const int sj = lane % offset;
        if (lane < ml) {
          const uint8_t bv = ring[(op - offset + sj) & (kRing - 1)];
          ring[(op + lane) & (kRing - 1)] = bv;
          out[op + lane] = bv;
        }
      } else if (decu & 2) {
        const int mlu = __builtin_amdgcn_readfirstlane(ml);
        const int offu = __builtin_amdgcn_readfirstlane(offset);
        if (offu >= kWave) {
          // rounds of 64: the source trails the write head by `offset` <= kRing - 64 bytes
          for (int j0 = 0; j0 < mlu; j0 += kWave) {
            const int j = j0 + lane;
            if (j < ml) {
              const uint8_t bv = ring[(op - offset + j) & (kRing - 1)];
              ring[(op + j) & (kRing - 1)] = bv;
              out[op + j] = bv;
            }
          }
        } else {
          // overlapping copy of any length = periodic pattern: each lane keeps ONE byte of the period
          // in a register (the copy may be longer than the ring, which would eat its own source)
          const int span = (kWave / offu) * offu;
          const uint8_t bv = ring[(op - offset + (lane % offu)) & (kRing - 1)];
          if (lane < span)
            for (int j = lane; j < mlu; j += span) {
              ring[(op + j) & (kRing - 1)] = bv;
              out[op + j] = bv;
            }
        }
      } else {
        // far: the source left the ring; read it back from L2 once this wave's stores have landed
        const int opu = __builtin_amdgcn_readfirstlane(op), offu = __builtin_amdgcn_readfirstlane(offset);
        const int mlu = __builtin_amdgcn_readfirstlane(ml);
        const uint8_t* src = out + opu - offu;
        for (int j0 = 0; j0 < mlu; j0 += kWave) {
          const int need = opu - offu + (mlu - j0 < kWave ? mlu : j0 + kWave);
          if (need > drained) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            drained = opu + j0;
          }
          const int j = j0 + lane;
          if (j < mlu) {
            const uint8_t bv = (uint8_t)ld_u8_l2(src + j);
            out[opu + j] = bv;
            ring[(opu + j) & (kRing - 1)] = bv;
          }
        }
      }
      op += ml;
      DDBG_T(tv2);
      DDBG_ADD(1, tv2 - tv1);
    }
#ifdef S3S_LZ4_TIMING
    ddbg[2] = __builtin_amdgcn_s_memtime() - tk0;
    ddbg[6] = 1;
    if (lane == 0)
      for (int i = 0; i < 8; i++) atomicAdd(&g_dec_dbg[i], ddbg[i]);
#endif
    if (!bad && __builtin_amdgcn_readfirstlane(op) != olen) bad = 1;
  }
  if (bad) {
    if (lane == 0) atomicExch(status, S3S_E_BAD_FRAME);
    return;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  {
    const uint32_t seed = kLz4BlockSeed;
    uint32_t acc = lane == 0 ? seed + XXP1 + XXP2 : lane == 1 ? seed + XXP2 : lane == 2 ? seed : seed - XXP1;
    const int stripes = olen >> 4;
    const int nblk = olen >> 8;
    auto ld32u = [&](int byte_pos) -> uint32_t {
      uint32_t x;
      __builtin_memcpy(&x, out + byte_pos, 4);
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
    for (int j = nblk * 16; j < stripes; j++) {
      const uint32_t wv = ld32u(16 * j + 4 * (lane & 3));
      acc = rotl32(acc + wv * XXP2, 13) * XXP1;
    }
    uint32_t h;
    if (olen >= 16) {
      const uint32_t v1 = __builtin_amdgcn_readlane(acc, 0), v2 = __builtin_amdgcn_readlane(acc, 1),
                     v3 = __builtin_amdgcn_readlane(acc, 2), v4 = __builtin_amdgcn_readlane(acc, 3);
      h = rotl32(v1, 1) + rotl32(v2, 7) + rotl32(v3, 12) + rotl32(v4, 18);
    } else {
      h = seed + XXP5;
    }
    h += (uint32_t)olen;
    int p = stripes << 4;
    for (; p + 4 <= olen; p += 4) h = rotl32(h + ld32u(p) * XXP3, 17) * XXP4;
    for (; p < olen; p++) h = rotl32(h + (uint32_t)out[p] * XXP5, 11) * XXP1;
    h ^= h >> 15;
    h *= XXP2;
    h ^= h >> 13;
    h *= XXP3;
    h ^= h >> 16;
    if (lane == 0 && (h & 0x0FFFFFFFu) != fr.check) atomicExch(status, S3S_E_BAD_FRAME);
  }
}

}  // namespace

int32_t lz4_tile_count(int64_t comp_len) { return (int32_t)((comp_len + kTileBytes - 1) / kTileBytes); }

void launch_lz4_discover(const uint8_t* d_comp, int64_t comp_len, int32_t n_tiles,
                         int64_t* d_spec_entry, int64_t* d_spec_exit, int32_t* d_spec_count,
                         int64_t* d_true_entry, int64_t* d_frame_base, int32_t* d_status,
                         hipStream_t st) {
  if (n_tiles <= 0) return;
  hipLaunchKernelGGL(tile_speculate_kernel, dim3((unsigned)n_tiles), dim3(kWave), 0, st, d_comp,
                     comp_len, n_tiles, d_spec_entry, d_spec_exit, d_spec_count);
  hipLaunchKernelGGL(tile_resolve_kernel, dim3(1), dim3(kWave), 0, st, d_comp, comp_len, n_tiles,
                     d_spec_entry, d_spec_exit, d_spec_count, d_true_entry, d_status);
  hipLaunchKernelGGL(scan_u32_kernel, dim3(1), dim3(kWave), 0, st,
                     reinterpret_cast<const uint32_t*>(d_spec_count), (int64_t)n_tiles, d_frame_base);
}

void launch_lz4_emit_frames(const uint8_t* d_comp, int64_t comp_len, int32_t n_tiles,
                            const int64_t* d_true_entry, const int64_t* d_frame_base,
                            Frame* d_frames, uint32_t* d_frame_orig, int64_t n_frames,
                            int64_t* d_frame_out, int32_t* d_status, hipStream_t st) {
  if (n_tiles <= 0) return;
  hipLaunchKernelGGL(tile_emit_kernel, dim3((unsigned)((n_tiles + kWave - 1) / kWave)), dim3(kWave),
                     0, st, d_comp, comp_len, n_tiles, d_true_entry, d_frame_base, d_frames,
                     d_frame_orig, d_status);
  hipLaunchKernelGGL(scan_u32_kernel, dim3(1), dim3(kWave), 0, st, d_frame_orig, n_frames,
                     d_frame_out);
}

void launch_scan_u32(const uint32_t* d_in, int64_t n, int64_t* d_out, hipStream_t st) {
  hipLaunchKernelGGL(scan_u32_kernel, dim3(1), dim3(kWave), 0, st, d_in, n, d_out);
}

void launch_lz4_decompress(const uint8_t* d_comp, const Frame* d_frames, int32_t n_frames,
                           const int64_t* d_frame_out, uint8_t* d_dst, int32_t* d_status,
                           int variant, hipStream_t st) {
  if (n_frames <= 0) return;
  if (variant == 4) {
    launch_lz4_decompress_batch(d_comp, d_frames, n_frames, d_frame_out, d_dst, d_status, st);
    return;
  }
  if (variant == 3) {
    hipLaunchKernelGGL(lz4_decompress_valu_kernel, dim3((unsigned)n_frames), dim3(kWave), 0, st, d_comp,
                       d_frames, n_frames, d_frame_out, d_dst, d_status);
    return;
  }
  if (variant == 2) {
    hipLaunchKernelGGL(lz4_decompress_ring_kernel, dim3((unsigned)n_frames), dim3(kWave), 0, st, d_comp,
                       d_frames, n_frames, d_frame_out, d_dst, d_status);
    return;
  }
  if (variant == 1) {
    hipLaunchKernelGGL(lz4_decompress_global_kernel, dim3((unsigned)n_frames), dim3(kWave), 0, st, d_comp,
                       d_frames, n_frames, d_frame_out, d_dst, d_status);
    return;
  }
  hipLaunchKernelGGL(lz4_decompress_kernel<kSmallComp>, dim3((unsigned)n_frames), dim3(kDecThreads), 0, st,
                     d_comp, d_frames, n_frames, d_frame_out, d_dst, d_status);
  hipLaunchKernelGGL(lz4_decompress_kernel<kMaxBlock>, dim3((unsigned)n_frames), dim3(kDecThreads), 0, st,
                     d_comp, d_frames, n_frames, d_frame_out, d_dst, d_status);
}

}  // namespace s3s
