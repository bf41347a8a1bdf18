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
// then one wavefront per frame decodes it: the batch decoder (lz4_decode_batch.hip, default) or the ring decoder
// below (S3S_OPT_LZ4_DECODE_VARIANT 3); frame hashes are checked by lz4_verify_frames_kernel / in the ring kernel.
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
__device__ __forceinline__ void speculate_tile(const uint8_t* __restrict__ comp, int64_t comp_len, int k, int lane,
                                               int64_t* __restrict__ spec_entry, int64_t* __restrict__ spec_exit,
                                               int32_t* __restrict__ spec_count) {
  const int64_t t0 = (int64_t)k * kTileBytes;
  const int64_t t1 = (t0 + kTileBytes) < comp_len ? (t0 + kTileBytes) : comp_len;
  int64_t entry = -1;
  if (k == 0) {
    entry = 0;
  } else {
    // 256 positions per step (round 4, with the wide rebase +1.2 % on the reduce side, profiles/r04a_first_call.txt): the four loads of a step are in flight together instead of one
    // load -> ballot round trip per 64 positions (a tile's first header is ~8 KiB in: ~130 of those round trips today)
    for (int64_t p0 = t0; p0 < t1 && entry < 0; p0 += 4 * kWave) {
      uint64_t w[4];
      bool in[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int64_t p = p0 + u * kWave + lane;
        in[u] = p < t1 && comp_len - p >= kLz4FrameHeader;
        w[u] = in[u] ? ld64u(comp + p) : 0ull;
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int64_t p = p0 + u * kWave + lane;
        bool hit = false;
        if (in[u] && w[u] == kMagic) {
          const Header hd = parse_header(comp + p);
          hit = hd.ok && p + kLz4FrameHeader + hd.comp_len <= comp_len;
        }
        const uint64_t m = __ballot(hit);
        if (m && entry < 0) entry = p0 + u * kWave + __builtin_ctzll(m);
      }
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

__global__ __launch_bounds__(kWave) void tile_speculate_kernel(const uint8_t* __restrict__ comp,
                                                              int64_t comp_len, int32_t n_tiles,
                                                              int64_t* __restrict__ spec_entry,
                                                              int64_t* __restrict__ spec_exit,
                                                              int32_t* __restrict__ spec_count) {
  const int k = blockIdx.x;
  if (k >= n_tiles) return;
  speculate_tile(comp, comp_len, k, threadIdx.x, spec_entry, spec_exit, spec_count);
}

__global__ __launch_bounds__(kWave) void tile_speculate_batch_kernel(const LzRange* __restrict__ ranges,
                                                                    const int32_t* __restrict__ tile_range,
                                                                    int32_t total_tiles) {
  const int b = blockIdx.x;
  if (b >= total_tiles) return;
  const LzRange r = ranges[tile_range[b]];
  speculate_tile(r.comp, r.comp_len, b - r.tile0, threadIdx.x, r.spec_entry, r.spec_exit, r.spec_count);
}

// one wavefront: turn speculation into the true chain.  true_entry[k] = position where the
// chain enters tile k (-1: no frame starts in tile k), count[k] = frames starting in tile k.
__device__ __forceinline__ void resolve_chain(
    const uint8_t* __restrict__ comp, int64_t comp_len, int32_t n_tiles,
    const int64_t* __restrict__ spec_entry, int64_t* __restrict__ spec_exit,
    int32_t* __restrict__ spec_count, int64_t* __restrict__ true_entry, int32_t* __restrict__ status, int lane) {
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

__global__ __launch_bounds__(kWave) void tile_resolve_kernel(
    const uint8_t* __restrict__ comp, int64_t comp_len, int32_t n_tiles,
    const int64_t* __restrict__ spec_entry, int64_t* __restrict__ spec_exit,
    int32_t* __restrict__ spec_count, int64_t* __restrict__ true_entry, int32_t* __restrict__ status) {
  resolve_chain(comp, comp_len, n_tiles, spec_entry, spec_exit, spec_count, true_entry, status, threadIdx.x);
}

__device__ __forceinline__ void emit_tile(const uint8_t* __restrict__ comp, int64_t comp_len, int k,
                                          const int64_t* __restrict__ true_entry, const int64_t* __restrict__ frame_base,
                                          Frame* __restrict__ frames, uint32_t* __restrict__ frame_orig,
                                          int32_t* __restrict__ status) {
  const int64_t entry = true_entry[k];
  if (entry < 0) return;
  const int64_t t0 = (int64_t)k * kTileBytes;
  const int64_t t1 = (t0 + kTileBytes) < comp_len ? (t0 + kTileBytes) : comp_len;
  int32_t cnt = 0;
  const int64_t base = frame_base[k];
  if (walk_tile(comp, comp_len, entry, t1, &cnt, frames + base, frame_orig + base) < 0)
    atomicExch(status, S3S_E_BAD_FRAME);
}

__global__ __launch_bounds__(kWave) void tile_emit_kernel(
    const uint8_t* __restrict__ comp, int64_t comp_len, int32_t n_tiles,
    const int64_t* __restrict__ true_entry, const int64_t* __restrict__ frame_base,
    Frame* __restrict__ frames, uint32_t* __restrict__ frame_orig, int32_t* __restrict__ status) {
  const int k = blockIdx.x * kWave + threadIdx.x;
  if (k >= n_tiles) return;
  emit_tile(comp, comp_len, k, true_entry, frame_base, frames, frame_orig, status);
}

__global__ __launch_bounds__(kWave) void tile_emit_batch_kernel(const LzRange* __restrict__ ranges,
                                                               const int32_t* __restrict__ tile_range,
                                                               int32_t total_tiles) {
  const int b = blockIdx.x * kWave + threadIdx.x;
  if (b >= total_tiles) return;
  const LzRange r = ranges[tile_range[b]];
  if (r.skip || r.n_frames == 0) return;
  emit_tile(r.comp, r.comp_len, b - r.tile0, r.true_entry, r.frame_base, r.frames, r.frame_orig, r.status);
}

// generic exclusive scan of uint32 -> int64 (n+1 outputs): ONE wavefront, NO LDS — the decode kernels of
// the other task threads book the whole LDS of every CU (20 x 8 KiB rings), and a workgroup that needs
// any of it waits for one of their frames to finish (see scan_items_kernel in assemble.hip)
__device__ __forceinline__ void scan_u32_wave(const uint32_t* __restrict__ in, int64_t n, int64_t* __restrict__ out,
                                              int lane) {
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

__global__ __launch_bounds__(kWave) void scan_u32_kernel(const uint32_t* __restrict__ in, int64_t n,
                                                         int64_t* __restrict__ out) {
  scan_u32_wave(in, n, out, threadIdx.x);
}

// batched call, phase 1 (one wavefront per range): speculation -> true chain -> frames per tile -> frame count
__global__ __launch_bounds__(kWave) void tile_resolve_batch_kernel(const LzRange* __restrict__ ranges, int32_t n_ranges) {
  const int ri = blockIdx.x;
  if (ri >= n_ranges) return;
  const LzRange r = ranges[ri];
  const int lane = threadIdx.x;
  if (r.n_tiles <= 0) {
    if (lane == 0) r.result[0] = 0;
    return;
  }
  resolve_chain(r.comp, r.comp_len, r.n_tiles, r.spec_entry, r.spec_exit, r.spec_count, r.true_entry, r.status, lane);
  __threadfence();
  scan_u32_wave(reinterpret_cast<const uint32_t*>(r.spec_count), (int64_t)r.n_tiles, r.frame_base, lane);
  __threadfence();
  if (lane == 0) r.result[0] = __builtin_nontemporal_load(&r.frame_base[r.n_tiles]);
}

// batched call, phase 2 (one wavefront per range): output offsets of the range's frames, its decoded size, the
// capacity check, and the rebase of its frame records to absolute addresses (empty frames for a range that failed)
__global__ __launch_bounds__(kWave) void frames_finish_batch_kernel(LzRange* __restrict__ ranges, int32_t n_ranges) {
  const int ri = blockIdx.x;
  if (ri >= n_ranges) return;
  const LzRange r = ranges[ri];
  const int lane = threadIdx.x;
  if (r.n_frames <= 0) {
    if (lane == 0) r.result[1] = 0;
    return;
  }
  int64_t total = 0;
  bool skip = r.skip != 0;
  if (!skip) {
    scan_u32_wave(r.frame_orig, r.n_frames, r.frame_out, lane);
    __threadfence();
    total = __builtin_nontemporal_load(&r.frame_out[r.n_frames]);
    skip = total > r.dst_capacity || __builtin_nontemporal_load(r.status) != 0;
  }
  if (lane == 0) r.result[1] = total;
  const int64_t comp_base = (int64_t)reinterpret_cast<uintptr_t>(r.comp);
  // (round 4) the rebase is one wavefront per range on the critical path in front of the decode launch
  // (50 us for four 128 MiB ranges, 3 % of the reduce-side step): four records per lane and iteration, their loads issued
  // together, instead of one load -> store round trip per record
  if (!skip) {
    for (int64_t i0 = 4 * (int64_t)lane; i0 < r.n_frames; i0 += 4 * kWave) {
      int64_t co[4], fo[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int64_t i = i0 + k < r.n_frames ? i0 + k : r.n_frames - 1;
        co[k] = r.frames[i].comp_off;
        fo[k] = __builtin_nontemporal_load(&r.frame_out[i]);
      }
#pragma unroll
      for (int k = 0; k < 4; k++)
        if (i0 + k < r.n_frames) {
          r.frames[i0 + k].comp_off = co[k] + comp_base;
          r.out_abs[i0 + k] = fo[k] + r.dst_base;
        }
    }
    return;
  }
  for (int64_t i = lane; i < r.n_frames; i += kWave) {
    if (skip) {
      r.frames[i] = Frame{0, 0, 0, 0u, 0x10};
      r.out_abs[i] = r.dst_base;
    } else {
      r.frames[i].comp_off += comp_base;
      r.out_abs[i] = __builtin_nontemporal_load(&r.frame_out[i]) + r.dst_base;
    }
  }
}

// ---- frame decode: ring decoder written for the VALU (decode variant 3; the batch decoder, variant 4 and
// default, is lz4_decode_batch.hip).  Round 1's other decoders (frame staged in LDS, straight to global
// memory, scalar-parse ring) are in the git history and DESIGN.md §6.
constexpr uint32_t XXP1 = 2654435761u, XXP2 = 2246822519u, XXP3 = 3266489917u,
                   XXP4 = 668265263u, XXP5 = 374761393u;

__device__ __forceinline__ uint32_t ld_u8_l2(const uint8_t* p) {
  return (uint32_t)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

constexpr int kRing = 8192;
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
    int wb = -(1 << 20);  // dword index (relative to c_al) of window lane 0; VGPR-uniform
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

void launch_lz4_discover_batch(const LzRange* d_ranges, int32_t n_ranges, const int32_t* d_tile_range,
                               int32_t total_tiles, hipStream_t st) {
  if (total_tiles > 0)
    hipLaunchKernelGGL(tile_speculate_batch_kernel, dim3((unsigned)total_tiles), dim3(kWave), 0, st, d_ranges,
                       d_tile_range, total_tiles);
  if (n_ranges > 0)
    hipLaunchKernelGGL(tile_resolve_batch_kernel, dim3((unsigned)n_ranges), dim3(kWave), 0, st, d_ranges, n_ranges);
}

void launch_lz4_frames_batch(LzRange* d_ranges, int32_t n_ranges, const int32_t* d_tile_range, int32_t total_tiles,
                             hipStream_t st) {
  if (total_tiles > 0)
    hipLaunchKernelGGL(tile_emit_batch_kernel, dim3((unsigned)((total_tiles + kWave - 1) / kWave)), dim3(kWave), 0, st,
                       d_ranges, d_tile_range, total_tiles);
  if (n_ranges > 0)
    hipLaunchKernelGGL(frames_finish_batch_kernel, dim3((unsigned)n_ranges), dim3(kWave), 0, st, d_ranges, n_ranges);
}

void launch_scan_u32(const uint32_t* d_in, int64_t n, int64_t* d_out, hipStream_t st) {
  hipLaunchKernelGGL(scan_u32_kernel, dim3(1), dim3(kWave), 0, st, d_in, n, d_out);
}

void launch_lz4_decompress(const uint8_t* d_comp, const Frame* d_frames, int32_t n_frames,
                           const int64_t* d_frame_out, uint8_t* d_dst, int32_t* d_status,
                           int variant, hipStream_t st, hipEvent_t after_decode) {
  if (n_frames <= 0) {
    if (after_decode) (void)hipEventRecord(after_decode, st);
    return;
  }
  if (variant == 3) {
    hipLaunchKernelGGL(lz4_decompress_valu_kernel, dim3((unsigned)n_frames), dim3(kWave), 0, st, d_comp,
                       d_frames, n_frames, d_frame_out, d_dst, d_status);
    if (after_decode) (void)hipEventRecord(after_decode, st);
    return;
  }
  launch_lz4_decompress_batch(d_comp, d_frames, n_frames, d_frame_out, d_dst, d_status, st, after_decode);
}

}  // namespace s3s
