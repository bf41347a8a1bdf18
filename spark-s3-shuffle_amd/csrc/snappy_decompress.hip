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

constexpr int kSnMaxComp = 32 + kMaxBlock + kMaxBlock / 6;  // snappy MaxCompressedLength(32 KiB)
constexpr int kDecThreads = 128;

__device__ __forceinline__ bool is_stream_header(const uint8_t* c) {
  return c[0] == 0x82 && c[1] == 'S' && c[2] == 'N' && c[3] == 'A' && c[4] == 'P' && c[5] == 'P' &&
         c[6] == 'Y' && c[7] == 0;
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
                                    int32_t* __restrict__ status) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_parts) return;
  const int n = walk_partition(comp, part_off[p], part_off[p + 1], false, nullptr, nullptr);
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
                                   int32_t* __restrict__ status) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_parts) return;
  const int64_t b = frame_base[p];
  if (walk_partition(comp, part_off[p], part_off[p + 1], true, frames + b, frame_orig + b) < 0)
    atomicExch(status, S3S_E_BAD_FRAME);
}

struct __attribute__((aligned(16))) SnDecLds {
  uint8_t out[kMaxBlock + 64];
  uint8_t comp[kSnMaxComp + 64];
  int error;
};

// raw snappy decode of comp[0,clen) into out[0,olen), one wavefront, control wave-uniform
__device__ int snappy_decode_wave(const uint8_t* comp, int clen, uint8_t* out, int olen, int lane) {
  int ip = 0, op = 0;
  {  // preamble
    uint32_t ulen = 0;
    int sh = 0;
    for (;;) {
      if (ip >= clen || sh > 28) return -1;
      const uint32_t b = comp[ip++];
      ulen |= (b & 0x7fu) << sh;
      if (!(b & 0x80u)) break;
      sh += 7;
    }
    if ((int)ulen != olen) return -1;
  }
  while (ip < clen) {
    const uint32_t tag = comp[ip++];
    int len, offset;
    if ((tag & 3u) == 0u) {
      len = (int)(tag >> 2) + 1;
      if (len > 60) {
        const int nb = len - 60;
        if (clen - ip < nb) return -1;
        uint32_t l = 0;
        for (int i = 0; i < nb; i++) l |= (uint32_t)comp[ip + i] << (8 * i);
        ip += nb;
        if (l >= (uint32_t)kMaxBlock) return -1;
        len = (int)l + 1;
      }
      if (len > clen - ip || len > olen - op) return -1;
      for (int j = lane; j < len; j += kWave) out[op + j] = comp[ip + j];
      ip += len;
      op += len;
      continue;
    }
    if ((tag & 3u) == 1u) {
      if (ip >= clen) return -1;
      len = 4 + (int)((tag >> 2) & 7u);
      offset = (int)((tag >> 5) << 8) | (int)comp[ip++];
    } else if ((tag & 3u) == 2u) {
      if (clen - ip < 2) return -1;
      len = (int)(tag >> 2) + 1;
      offset = (int)comp[ip] | ((int)comp[ip + 1] << 8);
      ip += 2;
    } else {
      if (clen - ip < 4) return -1;
      len = (int)(tag >> 2) + 1;
      const uint32_t o = (uint32_t)comp[ip] | (uint32_t)comp[ip + 1] << 8 | (uint32_t)comp[ip + 2] << 16 |
                         (uint32_t)comp[ip + 3] << 24;
      ip += 4;
      if (o > (uint32_t)kMaxBlock) return -1;
      offset = (int)o;
    }
    if (offset == 0 || offset > op || len > olen - op) return -1;
    if (offset >= kWave) {
      for (int j = lane; j < len; j += kWave) out[op + j] = out[op - offset + j];
    } else {
      // overlapping copy = periodic pattern (same trick as the LZ4 decoder)
      const int span = (kWave / offset) * offset;
      const uint32_t v = out[op - offset + (lane % offset)];
      if (lane < span)
        for (int j = lane; j < len; j += span) out[op + j] = (uint8_t)v;
    }
    op += len;
  }
  return op == olen ? 0 : -1;
}

__global__ __launch_bounds__(kDecThreads) void snappy_decompress_kernel(
    const uint8_t* __restrict__ comp, const Frame* __restrict__ frames, int32_t n_frames,
    const int64_t* __restrict__ frame_out, uint8_t* __restrict__ dst, int32_t* __restrict__ status) {
  __shared__ SnDecLds s;
  const int f = blockIdx.x;
  if (f >= n_frames) return;
  const Frame fr = frames[f];
  const int olen = fr.orig_len, clen = fr.comp_len;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (olen > kMaxBlock || clen > kSnMaxComp) {
    if (tid == 0) atomicExch(status, S3S_E_UNSUPPORTED);
    return;
  }
  const uint8_t* g = comp + fr.comp_off;
  for (int i = tid * 16; i + 16 <= clen; i += kDecThreads * 16) {
    uint4 x;
    __builtin_memcpy(&x, g + i, 16);
    *reinterpret_cast<uint4*>(s.comp + i) = x;
  }
  for (int i = (clen & ~15) + tid; i < clen; i += kDecThreads) s.comp[i] = g[i];
  if (tid == 0) s.error = 0;
  __syncthreads();
  if (wave == 0) {
    const int rc = snappy_decode_wave(s.comp, clen, s.out, olen, lane);
    if (rc != 0 && lane == 0) {
      s.error = 1;
      atomicExch(status, S3S_E_BAD_FRAME);
    }
  }
  __syncthreads();
  if (s.error || olen == 0) return;
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

void launch_snappy_count_frames(const uint8_t* d_comp, const int64_t* d_part_off, int32_t n_parts,
                                uint32_t* d_part_nframes, int32_t* d_status, hipStream_t st) {
  if (n_parts <= 0) return;
  hipLaunchKernelGGL(snappy_count_kernel, dim3((unsigned)((n_parts + 63) / 64)), dim3(64), 0, st, d_comp,
                     d_part_off, n_parts, d_part_nframes, d_status);
}

void launch_snappy_emit_frames(const uint8_t* d_comp, const int64_t* d_part_off, int32_t n_parts,
                               const int64_t* d_frame_base, Frame* d_frames, uint32_t* d_frame_orig,
                               int32_t* d_status, hipStream_t st) {
  if (n_parts <= 0) return;
  hipLaunchKernelGGL(snappy_emit_kernel, dim3((unsigned)((n_parts + 63) / 64)), dim3(64), 0, st, d_comp,
                     d_part_off, n_parts, d_frame_base, d_frames, d_frame_orig, d_status);
}

void launch_snappy_decompress(const uint8_t* d_comp, const Frame* d_frames, int32_t n_frames,
                              const int64_t* d_frame_out, uint8_t* d_dst, int32_t* d_status,
                              hipStream_t st) {
  if (n_frames <= 0) return;
  hipLaunchKernelGGL(snappy_decompress_kernel, dim3((unsigned)n_frames), dim3(kDecThreads), 0, st, d_comp,
                     d_frames, n_frames, d_frame_out, d_dst, d_status);
}

}  // namespace s3s
