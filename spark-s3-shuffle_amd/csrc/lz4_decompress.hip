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

// generic single-workgroup exclusive scan of uint32 -> int64 (n+1 outputs)
constexpr int kScanThreads = 1024;
__global__ __launch_bounds__(kScanThreads) void scan_u32_kernel(const uint32_t* __restrict__ in,
                                                               int64_t n,
                                                               int64_t* __restrict__ out) {
  __shared__ int64_t wave_sum[kScanThreads / kWave];
  __shared__ int64_t carry;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int64_t tile = 0; tile < n; tile += kScanThreads) {
    const int64_t i = tile + tid;
    const int64_t x = i < n ? (int64_t)in[i] : 0;
    int64_t inc = x;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
      const int64_t y = __shfl_up(inc, d);
      if (lane >= d) inc += y;
    }
    if (lane == kWave - 1) wave_sum[wave] = inc;
    __syncthreads();
    int64_t before = carry;
    for (int wv = 0; wv < wave; wv++) before += wave_sum[wv];
    if (i < n) out[i] = before + inc - x;
    __syncthreads();
    if (tid == kScanThreads - 1) carry = before + inc;
    __syncthreads();
  }
  if (tid == 0) out[n] = carry;
}

// ---- frame decode ----------------------------------------------------------------------------
constexpr int kDecThreads = 128;
constexpr uint32_t XXP1 = 2654435761u, XXP2 = 2246822519u, XXP3 = 3266489917u,
                   XXP4 = 668265263u, XXP5 = 374761393u;

struct __attribute__((aligned(16))) DecLds {
  uint8_t out[kMaxBlock + 64];
  uint8_t comp[kMaxBlock + 64];
  int progress;  // bytes of out[] final so far (decoder -> hasher)
  int error;
  uint32_t xxh;
};

// LZ4 block decode of comp[0,clen) into out[0,olen), one wavefront, all control wave-uniform.
// Returns 0 or -1 (malformed).  Publishes progress for the trailing hasher.
__device__ int lz4_decode_wave(const uint8_t* comp, int clen, uint8_t* out, int olen,
                               volatile int* progress, int lane) {
  int ip = 0, op = 0;
  for (;;) {
    if (ip >= clen) return -1;
    const uint32_t token = comp[ip++];
    int lit = (int)(token >> 4);
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
    const int offset = (int)comp[ip] | ((int)comp[ip + 1] << 8);
    ip += 2;
    if (offset == 0 || offset > op) return -1;
    int ml = (int)(token & 15u);
    if (ml == 15) {
      uint32_t b;
      do {
        if (ip >= clen) return -1;
        b = comp[ip++];
        ml += (int)b;
      } while (b == 255);
    }
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
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) *progress = op;
  }
  if (op != olen) return -1;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if (lane == 0) *progress = olen;
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

__global__ __launch_bounds__(kDecThreads) void lz4_decompress_kernel(
    const uint8_t* __restrict__ comp, const Frame* __restrict__ frames, int32_t n_frames,
    const int64_t* __restrict__ frame_out, uint8_t* __restrict__ dst, int32_t* __restrict__ status) {
  __shared__ DecLds s;
  const int f = blockIdx.x;
  if (f >= n_frames) return;
  const Frame fr = frames[f];
  const int olen = fr.orig_len, clen = fr.comp_len;
  if (olen == 0) return;  // end-of-stream frame: nothing to emit, decoding continues
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (olen > kMaxBlock) {
    if (tid == 0) atomicExch(status, S3S_E_UNSUPPORTED);
    return;
  }
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
  hipLaunchKernelGGL(scan_u32_kernel, dim3(1), dim3(kScanThreads), 0, st,
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
  hipLaunchKernelGGL(scan_u32_kernel, dim3(1), dim3(kScanThreads), 0, st, d_frame_orig, n_frames,
                     d_frame_out);
}

void launch_scan_u32(const uint32_t* d_in, int64_t n, int64_t* d_out, hipStream_t st) {
  hipLaunchKernelGGL(scan_u32_kernel, dim3(1), dim3(kScanThreads), 0, st, d_in, n, d_out);
}

void launch_lz4_decompress(const uint8_t* d_comp, const Frame* d_frames, int32_t n_frames,
                           const int64_t* d_frame_out, uint8_t* d_dst, int32_t* d_status,
                           hipStream_t st) {
  if (n_frames <= 0) return;
  hipLaunchKernelGGL(lz4_decompress_kernel, dim3((unsigned)n_frames), dim3(kDecThreads), 0, st,
                     d_comp, d_frames, n_frames, d_frame_out, d_dst, d_status);
}

}  // namespace s3s
