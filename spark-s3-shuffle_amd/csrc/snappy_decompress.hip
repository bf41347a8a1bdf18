// snappy_decompress.hip — reduce side of the Snappy leg: walk the SnappyInputStream framing of
// every fetched partition, then decode one raw snappy block per workgroup.
//
// Replaces the [EXT] serializerManager.wrapStream -> SnappyInputStream stage behind
// storage/S3ShuffleReader.scala:98-110 (snappy-java: 16-byte stream header, then
// i32 BE compressedLength | raw snappy per chunk; concatenated streams are accepted, which is
// what makes batch fetch and multi-spill merge legal, S3ShuffleReader.scala:55-75).
#include "s3s_internal.h"

namespace s3s {
namespace {


__device__ __forceinline__ bool is_stream_header(const uint8_t* c) {
  return c[0] == 0x82 && c[1] == 'S' && c[2] == 'N' && c[3] == 'A' && c[4] == 'P' && c[5] == 'P' &&
         c[6] == 'Y' && c[7] == 0;
}

// LZFInputStream (compress-lzf) framing: 'Z' 'V' 0 | len u16 BE | bytes   or   'Z' 'V' 1 | clen u16 BE | ulen u16 BE | LZF block;
// chunks until the end of the partition (concatenated streams are simply more chunks).
__device__ int walk_partition_lzf(const uint8_t* comp, int64_t beg, int64_t end, bool emit, Frame* frames, uint32_t* frame_orig) {
  int64_t ip = beg;
  int n = 0;
  while (ip < end) {
    if (end - ip < 5 || comp[ip] != 'Z' || comp[ip + 1] != 'V' || comp[ip + 2] > 1) return -1;
    const int type = comp[ip + 2];
    const uint32_t len = (uint32_t)comp[ip + 3] << 8 | comp[ip + 4];
    ip += 5;
    uint32_t ulen = len;
    if (type == 1) {
      if (end - ip < 2) return -1;
      ulen = (uint32_t)comp[ip] << 8 | comp[ip + 1];
      ip += 2;
      if (len == 0 || ulen == 0) return -1;  // (a compressed chunk of nothing is not something the encoder writes)
    }
    if ((int64_t)len > end - ip) return -1;
    if (emit) {
      Frame f;
      f.comp_off = ip;
      f.comp_len = (int32_t)len;
      f.orig_len = (int32_t)ulen;
      f.check = 0;
      f.method = type == 1 ? 2 : 0x10;
      frames[n] = f;
      frame_orig[n] = ulen;
    }
    n++;
    ip += len;
  }
  return n;
}

// Walks partition p.  emit == false: counts chunks.  emit == true: writes frames.
// Returns the chunk count or -1 on a malformed stream.
__device__ int walk_partition(const uint8_t* comp, int64_t beg, int64_t end, bool emit, Frame* frames,
                              uint32_t* frame_orig) {
  int64_t ip = beg;
  int n = 0;
  if (ip == end) return 0;
  if (end - ip < kSnappyStreamHeader || !is_stream_header(comp + ip)) return -1;
  ip += kSnappyStreamHeader;
  while (ip < end) {
    if (end - ip < 4) return -1;
    const uint32_t cl = (uint32_t)comp[ip] << 24 | (uint32_t)comp[ip + 1] << 16 |
                        (uint32_t)comp[ip + 2] << 8 | (uint32_t)comp[ip + 3];
    if (cl == 0x82534e41u) {  // the next concatenated stream starts here
      if (end - ip < kSnappyStreamHeader || !is_stream_header(comp + ip)) return -1;
      ip += kSnappyStreamHeader;
      continue;
    }
    ip += 4;
    if ((int64_t)cl > end - ip || cl == 0) return -1;
    uint32_t ulen = 0;
    int sh = 0;
    uint32_t i = 0;
    for (;; i++, sh += 7) {
      if (i >= cl || sh > 28) return -1;
      const uint32_t b = comp[ip + i];
      ulen |= (b & 0x7fu) << sh;
      if (!(b & 0x80u)) break;
    }
    if (emit) {
      Frame f;
      f.comp_off = ip;
      f.comp_len = (int32_t)cl;
      f.orig_len = (int32_t)ulen;
      f.check = 0;
      f.method = 1;
      frames[n] = f;
      frame_orig[n] = ulen;
    }
    n++;
    ip += cl;
  }
  return n;
}

__global__ void snappy_count_kernel(const uint8_t* __restrict__ comp, const int64_t* __restrict__ part_off,
                                    int32_t n_parts, uint32_t* __restrict__ part_nframes,
                                    int32_t* __restrict__ status, int chunk_format) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_parts) return;
  const int n = chunk_format == kChunkLzf ? walk_partition_lzf(comp, part_off[p], part_off[p + 1], false, nullptr, nullptr)
                                          : walk_partition(comp, part_off[p], part_off[p + 1], false, nullptr, nullptr);
  if (n < 0) {
    atomicExch(status, S3S_E_BAD_FRAME);
    part_nframes[p] = 0;
  } else {
    part_nframes[p] = (uint32_t)n;
  }
}

__global__ void snappy_emit_kernel(const uint8_t* __restrict__ comp, const int64_t* __restrict__ part_off,
                                   int32_t n_parts, const int64_t* __restrict__ frame_base,
                                   Frame* __restrict__ frames, uint32_t* __restrict__ frame_orig,
                                   int32_t* __restrict__ status, int chunk_format) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_parts) return;
  const int64_t b = frame_base[p];
  if ((chunk_format == kChunkLzf ? walk_partition_lzf(comp, part_off[p], part_off[p + 1], true, frames + b, frame_orig + b)
                                 : walk_partition(comp, part_off[p], part_off[p + 1], true, frames + b, frame_orig + b)) < 0)
    atomicExch(status, S3S_E_BAD_FRAME);
}

// ---- block decode, variant 1: ring decoder written for the VALU (same design as
// lz4_decompress_valu_kernel: no staging of the block, 256-byte stream window across the wave,
// 8 KiB LDS ring of recent output, wave-uniform parse values kept in VGPRs) ----
constexpr int kSnRing = 8192;

__device__ __forceinline__ uint32_t sn_ld_u8_l2(const uint8_t* p) {
  return (uint32_t)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(kWave) void snappy_decompress_valu_kernel(
    const uint8_t* __restrict__ comp, const Frame* __restrict__ frames, int32_t n_frames,
    const int64_t* __restrict__ frame_out, uint8_t* dst, int32_t* __restrict__ status) {
  __shared__ uint8_t ring[kSnRing];
  const int f = blockIdx.x;
  if (f >= n_frames) return;
  const Frame fr = frames[f];
  const int olen = fr.orig_len, clen = fr.comp_len;
  const int lane = threadIdx.x;
  if (olen == 0 && clen == 0) return;  // a frame the batched call skips
  if (olen > kMaxBlock) {
    if (lane == 0) atomicExch(status, S3S_E_UNSUPPORTED);
    return;
  }
  const uint8_t* c = comp + fr.comp_off;
  uint8_t* out = dst + frame_out[f];
  int bad = 0;
  // preamble: varint32 uncompressed length (scalar, once per block)
  int ip0 = 0;
  {
    uint32_t ulen = 0;
    int sh = 0;
    for (;;) {
      if (ip0 >= clen || sh > 28) { bad = 1; break; }
      const uint32_t b = __builtin_amdgcn_readfirstlane((uint32_t)c[ip0]);
      ip0++;
      ulen |= (b & 0x7fu) << sh;
      if (!(b & 0x80u)) break;
      sh += 7;
    }
    if (!bad && (int)ulen != olen) bad = 1;
  }
  if (!bad) {
    int ip = ip0, op = 0;
    asm volatile("" : "+v"(ip), "+v"(op));
    int drained = 0;
    const uint32_t c_lo = (uint32_t)(reinterpret_cast<uint64_t>(c) & 3u);
    const uint8_t* c_al = c - c_lo;
    const int last_dw = (int)((c_lo + (uint32_t)clen - 1u) >> 2);
    int wb = -(1 << 20);
    asm volatile("" : "+v"(wb));
    uint32_t win = 0;
    for (;;) {
      const int ipu = __builtin_amdgcn_readfirstlane(ip);
      if (ipu >= clen) break;
      int len = 0, offset = 0;
      int kind = 3;  // 0 literal done, 1 copy parsed, 3 slow
      if (ipu + 72 <= clen) {
        const int ab = ip + (int)c_lo;
        int k = (ab >> 2) - wb;
        if (__builtin_amdgcn_readfirstlane((int)(k < 0 || k > 45))) {  // window must hold tag + 64 literal bytes
          wb = ab >> 2;
          int di = wb + lane;
          di = di < last_dw ? di : last_dw;
          win = reinterpret_cast<const uint32_t*>(c_al)[di];
          k = 0;
        }
        const uint32_t d0 = (uint32_t)__shfl((int)win, k), d1 = (uint32_t)__shfl((int)win, k + 1);
        const uint32_t w0 = __builtin_amdgcn_alignbit(d1, d0, ((uint32_t)ab & 3u) * 8u);  // tag + 3 bytes
        const uint32_t tag = w0 & 0xffu;
        const uint32_t ty = tag & 3u;
        const int lit_len = (int)(tag >> 2) + 1;
        const int lit_ok = (ty == 0u) & (lit_len <= 60) & (lit_len <= olen - op);
        const int cp_ok = (ty == 1u) | (ty == 2u);
        const int len_c = ty == 1u ? 4 + (int)((tag >> 2) & 7u) : (int)(tag >> 2) + 1;
        const int off_c = ty == 1u ? (int)(((tag >> 5) << 8) | ((w0 >> 8) & 0xffu)) : (int)((w0 >> 8) & 0xffffu);
        const int sel = __builtin_amdgcn_readfirstlane(lit_ok | (cp_ok << 1));
        if (sel & 1) {
          // literal of <= 60 bytes: stream bytes ip+1 .. ip+len are bytes of the window
          const uint32_t la = (uint32_t)(ab - 4 * wb) + 1u + (uint32_t)lane;
          const uint32_t dwv = (uint32_t)__shfl((int)win, (int)((la >> 2) & 63u));
          if (lane < lit_len) {
            const uint8_t bv = (uint8_t)(dwv >> (8u * (la & 3u)));
            out[op + lane] = bv;
            ring[(op + lane) & (kSnRing - 1)] = bv;
          }
          op += lit_len;
          ip += 1 + lit_len;
          kind = 0;
        } else if (sel & 2) {
          len = len_c;
          offset = off_c;
          ip += ty == 1u ? 2 : 3;
          kind = 1;
        }
      }
      if (kind == 3) {
        // byte-wise element parse: long literals, 4-byte-offset copies, the tail of the block (rare)
        int ips = ipu, ops = __builtin_amdgcn_readfirstlane(op);
        const uint32_t tag = __builtin_amdgcn_readfirstlane((uint32_t)c[ips]);
        ips++;
        if ((tag & 3u) == 0u) {
          int l = (int)(tag >> 2) + 1;
          if (l > 60) {
            const int nb = l - 60;
            if (clen - ips < nb) { bad = 1; break; }
            uint32_t lv = 0;
            for (int i = 0; i < nb; i++) lv |= __builtin_amdgcn_readfirstlane((uint32_t)c[ips + i]) << (8 * i);
            ips += nb;
            if (lv >= (uint32_t)kMaxBlock) { bad = 1; break; }
            l = (int)lv + 1;
          }
          if (l > clen - ips || l > olen - ops) { bad = 1; break; }
          for (int j = lane; j < l; j += kWave) {
            const uint8_t bv = c[ips + j];
            out[ops + j] = bv;
            ring[(ops + j) & (kSnRing - 1)] = bv;
          }
          ip = ips + l;
          op = ops + l;
          continue;
        }
        if ((tag & 3u) == 1u) {
          if (ips >= clen) { bad = 1; break; }
          len = 4 + (int)((tag >> 2) & 7u);
          offset = (int)((tag >> 5) << 8) | (int)__builtin_amdgcn_readfirstlane((uint32_t)c[ips]);
          ips += 1;
        } else if ((tag & 3u) == 2u) {
          if (clen - ips < 2) { bad = 1; break; }
          len = (int)(tag >> 2) + 1;
          offset = (int)__builtin_amdgcn_readfirstlane((uint32_t)c[ips] | ((uint32_t)c[ips + 1] << 8));
          ips += 2;
        } else {
          if (clen - ips < 4) { bad = 1; break; }
          len = (int)(tag >> 2) + 1;
          const uint32_t o = __builtin_amdgcn_readfirstlane((uint32_t)c[ips] | ((uint32_t)c[ips + 1] << 8) |
                                                            ((uint32_t)c[ips + 2] << 16) | ((uint32_t)c[ips + 3] << 24));
          ips += 4;
          if (o > (uint32_t)kMaxBlock) { bad = 1; break; }
          offset = (int)o;
        }
        ip = ips;
        kind = 1;
      }
      if (kind == 0) continue;
      // ---- copy of len <= 64 bytes ------------------------------------------------------------------
      const int near = (offset + kWave <= kSnRing);
      const int dec = ((offset == 0) | (offset > op) | (len > olen - op)) | (near << 1) | ((offset >= kWave) << 2);
      const int decu = __builtin_amdgcn_readfirstlane(dec);
      if (decu & 1) { bad = 1; break; }
      if ((decu & 6) == 6) {
        if (lane < len) {  // the common copy: no overlap
          const uint8_t bv = ring[(op - offset + lane) & (kSnRing - 1)];
          ring[(op + lane) & (kSnRing - 1)] = bv;
          out[op + lane] = bv;
        }
      } else if (decu & 2) {
        const int sj = lane % offset;  // overlap: periodic
        if (lane < len) {
          const uint8_t bv = ring[(op - offset + sj) & (kSnRing - 1)];
          ring[(op + lane) & (kSnRing - 1)] = bv;
          out[op + lane] = bv;
        }
      } else {
        const int opu = __builtin_amdgcn_readfirstlane(op), offu = __builtin_amdgcn_readfirstlane(offset);
        const int lenu = __builtin_amdgcn_readfirstlane(len);
        if (opu - offu + lenu > drained) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          drained = opu;
        }
        if (lane < lenu) {
          const uint8_t bv = (uint8_t)sn_ld_u8_l2(out + opu - offu + lane);
          out[opu + lane] = bv;
          ring[(opu + lane) & (kSnRing - 1)] = bv;
        }
      }
      op += len;
    }
    if (!bad && __builtin_amdgcn_readfirstlane(op) != olen) bad = 1;
  }
  if (bad && lane == 0) atomicExch(status, S3S_E_BAD_FRAME);
}

}  // namespace

void launch_snappy_count_frames(const uint8_t* d_comp, const int64_t* d_part_off, int32_t n_parts,
                                uint32_t* d_part_nframes, int32_t* d_status, hipStream_t st, int chunk_format) {
  if (n_parts <= 0) return;
  hipLaunchKernelGGL(snappy_count_kernel, dim3((unsigned)((n_parts + 63) / 64)), dim3(64), 0, st, d_comp,
                     d_part_off, n_parts, d_part_nframes, d_status, chunk_format);
}

void launch_snappy_emit_frames(const uint8_t* d_comp, const int64_t* d_part_off, int32_t n_parts,
                               const int64_t* d_frame_base, Frame* d_frames, uint32_t* d_frame_orig,
                               int32_t* d_status, hipStream_t st, int chunk_format) {
  if (n_parts <= 0) return;
  hipLaunchKernelGGL(snappy_emit_kernel, dim3((unsigned)((n_parts + 63) / 64)), dim3(64), 0, st, d_comp,
                     d_part_off, n_parts, d_frame_base, d_frames, d_frame_orig, d_status, chunk_format);
}

void launch_snappy_decompress(const uint8_t* d_comp, const Frame* d_frames, int32_t n_frames,
                              const int64_t* d_frame_out, uint8_t* d_dst, int32_t* d_status,
                              int variant, hipStream_t st, int chunk_format) {
  if (n_frames <= 0) return;
  if (chunk_format == kChunkLzf) {  // (LZF has the batch decoder only)
    launch_lzf_decompress_batch(d_comp, d_frames, n_frames, d_frame_out, d_dst, d_status, st);
    return;
  }
  if (variant == 3) {  // ring decoder on the vector ALU (round 1)
    hipLaunchKernelGGL(snappy_decompress_valu_kernel, dim3((unsigned)n_frames), dim3(kWave), 0, st, d_comp,
                       d_frames, n_frames, d_frame_out, d_dst, d_status);
    return;
  }
  launch_snappy_decompress_batch(d_comp, d_frames, n_frames, d_frame_out, d_dst, d_status, st);  // default
}

}  // namespace s3s
