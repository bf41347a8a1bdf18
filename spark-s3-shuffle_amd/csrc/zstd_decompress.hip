// zstd_decompress.hip — reduce side for spark.io.compression.codec=zstd: verify + decode the Zstandard
// frames of fetched block ranges.  The decoder itself is zstd_decode_core.h (shared with its host model); this file
// is the kernel around it and the orchestration behind s3s_decompress_range* / s3s_decompressed_size for
// S3S_CODEC_ZSTD.  One sequence wavefront per partition plus a literal wavefront per two partitions (see the kernel), one pass
// at guessed capacities (zstd_decompress_ranges) or, when a partition outgrows its guess, two passes over the compressed bytes:
//   pass 1  sizes: headers, table descriptions and the sequence streams are decoded, nothing is written — a Spark
//           writer's frames carry no content size (streaming API), and the partitions must land back to back in dst;
//   pass 2  decode into dst + the partition's offset, Huffman literals through a per-partition scratch buffer sized
//           by pass 1 (the largest regenerated literals section of its blocks).
// Compression with this codec stays on the JVM: s3s_compress_* answer S3S_E_UNSUPPORTED.
#define S3S_ZSTD_DEVICE 1
#include "s3s_ctx.h"
#include "zstd_decode_core.h"

#include <vector>

namespace s3s {
namespace {

struct ZPart {       // one partition of one range
  const uint8_t* src;
  int64_t size;
  uint8_t* dst;      // pass 2: where its decoded bytes go
  int64_t cap;       // pass 2: bytes available there
  int64_t lit_off;   // pass 2: offset of its literals scratch: two buffers lit_stride apart (block k uses buffer k & 1)
  int64_t lit_stride;
};
struct ZRes {
  int64_t total;     // decoded bytes
  int64_t lit_need;  // scratch the partition needs in pass 2
  int32_t rc;        // ZS_OK / ZS_BAD / ZS_CAPACITY / ZS_UNSUPPORTED (= the S3S_E_* values)
  int32_t pad;
};

// Three wavefronts per workgroup, two partitions (round 4).  Wavefronts 0 and 1 are the SEQUENCE sides of partitions 2b and
// 2b + 1: headers, FSE tables, the sequence loop and its copies.  Wavefront 2 is the LITERAL side of both: it walks the same
// headers and regenerates the Huffman-coded literals of a partition's NEXT block while its sequence wavefront executes this
// one (a level-1 TeraSort frame spent 8.5 of its 38 ms in Huffman streams on 4 of 64 lanes, and the rest never needed those
// lanes' results before the next block; one literal wavefront keeps up with two sequence sides).  They meet through four
// counters in LDS per partition (s3s_zstd::LitPipe) and two literal buffers per partition.  Registers: the sequence side
// needs 186 VGPRs when left alone; three wavefronts per SIMD (12 per CU = 4 workgroups = 8 partitions in flight, what the
// one-wavefront-per-partition kernel had) leave it 168.
constexpr int kZstdThreads = 3 * kWave;
__global__ __launch_bounds__(kZstdThreads, 3) void zstd_partitions_kernel(const ZPart* __restrict__ parts, int32_t n, int execute,
                                                                         uint8_t* __restrict__ lit_base, ZRes* __restrict__ res) {
  __shared__ s3s_zstd::Work w[2];
  __shared__ s3s_zstd::LitPipe lp[2];
  __shared__ s3s_zstd::LitWalker ks[2];  // (the literal wavefront's own state: indexed by partition, so not in registers)
  const int p0 = 2 * (int)blockIdx.x;
  if (p0 >= n) return;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  s3s_zstd::Lanes L{(int)(threadIdx.x & (kWave - 1)), kWave};
  if (threadIdx.x < 2) lp[threadIdx.x].ready = lp[threadIdx.x].consumed = lp[threadIdx.x].err = lp[threadIdx.x].quit = 0;
  __syncthreads();
  if (wave == 2) {
    if (execute == 0) return;
    const int np = n - p0 < 2 ? n - p0 : 2;
    for (int t = 0; t < np; t++) {
      const ZPart zp = parts[p0 + t];
      ks[t].src = zp.src;
      ks[t].size = zp.size;
      ks[t].ip = 0;
      ks[t].lit_buf = lit_base ? lit_base + zp.lit_off : nullptr;
      ks[t].lit_stride = zp.lit_stride;
      ks[t].hblock = 0;
      ks[t].in_frame = ks[t].has_check = ks[t].have = 0;
      ks[t].done = zp.size > 0 ? 0 : 1;
    }
    s3s_zstd::literal_side(lp, ks, np, L);
    return;
  }
  const int p = p0 + wave;
  if (p >= n) return;
  __builtin_amdgcn_s_setprio(3);  // (the sequence side is the partition's critical path, the literal wavefront has slack: + 4.7 %)
  const ZPart zp = parts[p];
  int64_t total = 0, need = 0;
  int rc = s3s_zstd::ZS_OK;
  if (zp.size > 0)
    rc = s3s_zstd::decode_partition(w[wave], lp[wave], zp.src, zp.size, zp.dst, zp.cap, execute != 0,
                                    lit_base ? lit_base + zp.lit_off : nullptr, zp.lit_stride, L, &total, &need);
  s3s_zstd::pipe_set(&lp[wave].quit, 1, L);  // (a literal side still waiting for a buffer leaves)
  if (L.lane == 0) {
    ZRes r;
    r.total = total;
    r.lit_need = need;
    r.rc = rc;
    r.pad = 0;
    res[p] = r;
  }
}

// ---- single pass (round 4): partitions decoded at guessed capacities are moved back to back ------------------------------
struct ZPiece {
  const uint8_t* src;
  uint8_t* dst;
  int64_t n;
};
constexpr int kPieceBytes = 1 << 16;
constexpr int kCopyThreads = 256;

__global__ __launch_bounds__(kCopyThreads) void zstd_compact_kernel(const ZPiece* __restrict__ pieces, int32_t n_pieces) {
  const int i = blockIdx.x;
  if (i >= n_pieces) return;
  const ZPiece pc = pieces[i];
  const int tid = threadIdx.x, n = (int)pc.n;
  uint8_t* dst = pc.dst;
  const uint8_t* src = pc.src;
  int head = (int)((16u - (uint32_t)(uintptr_t)dst) & 15u);  // 16-byte stores on the destination's alignment
  head = head < n ? head : n;
  if (tid < head) dst[tid] = src[tid];
  const int nvec = (n - head) >> 4;
  uint4* d16 = reinterpret_cast<uint4*>(dst + head);
  const uint8_t* sv = src + head;
  for (int v = tid; v < nvec; v += kCopyThreads) {
    uint4 x;
    __builtin_memcpy(&x, sv + 16 * v, 16);
    d16[v] = x;
  }
  const int done = head + 16 * nvec;
  if (tid < n - done) dst[done + tid] = src[done + tid];
}

}  // namespace

static_assert(s3s_zstd::ZS_BAD == S3S_E_BAD_FRAME && s3s_zstd::ZS_CAPACITY == S3S_E_CAPACITY &&
                  s3s_zstd::ZS_UNSUPPORTED == S3S_E_UNSUPPORTED, "decoder codes are the ABI's");

// Verify + decode n_ranges fetched ranges (device buffers).  size_only: pass 1 alone, out_len = decoded bytes.
// Per range: status / out_len / bad_partition as s3s_decompress_range_device reports them.  Returns the first error.
int zstd_decompress_ranges(s3s_ctx* ctx, int checksum_algo, s3s_fetch_range* R, int32_t n_ranges, bool size_only, bool* regular_end) {
  if (regular_end) *regular_end = false;  // true: every range has its own verdict (the return value is the first bad one's)
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  for (auto& v : ctx->stage_ms) v = 0;
  int64_t n_parts64 = 0, n_segs64 = 0;
  for (int32_t r = 0; r < n_ranges; r++) {
    s3s_fetch_range& k = R[r];
    k.status = S3S_OK;
    k.out_len = 0;
    k.bad_partition = -1;
    n_parts64 += k.num_partitions;
    for (int32_t p = 0; p < k.num_partitions; p++) n_segs64 += worst_segs(k.part_offsets[p + 1] - k.part_offsets[p]);
  }
  if (n_parts64 > 0x7fffff00ll || n_segs64 > 0x7fffff00ll) return fail(ctx, S3S_E_UNSUPPORTED, "batch too large for one call");
  const int32_t n_parts = (int32_t)n_parts64;
  auto first_error = [&]() -> int {
    for (int32_t r = 0; r < n_ranges; r++)
      if (R[r].status != S3S_OK) {
        const s3s_fetch_range& k = R[r];
        if (k.status == S3S_E_CHECKSUM) return fail(ctx, k.status, "Invalid checksum detected for partition %d of range %d", k.bad_partition, r);
        if (k.status == S3S_E_CAPACITY) return fail(ctx, k.status, "range %d: dst_capacity %lld < %lld decoded bytes", r, (long long)k.dst_capacity, (long long)k.out_len);
        if (k.status == S3S_E_UNSUPPORTED) return fail(ctx, k.status, "range %d: zstd frame with a dictionary id", r);
        return fail(ctx, k.status, "Stream is corrupted (zstd, range %d)", r);
      }
    return S3S_OK;
  };
  if (n_parts == 0) {
    if (regular_end) *regular_end = true;
    return S3S_OK;
  }
  // pinned staging: [ZPart n_parts][ZRes n_parts][offsets + seg starts + sums per range]
  auto al = [](size_t x) { return (x + 15) & ~size_t(15); };
  const size_t o_parts = 0, o_res = al(o_parts + sizeof(ZPart) * (size_t)n_parts), o_off = al(o_res + sizeof(ZRes) * (size_t)n_parts),
               o_seg = al(o_off + 8 * ((size_t)n_parts + (size_t)n_ranges)), o_sums = al(o_seg + 4 * ((size_t)n_parts + (size_t)n_ranges)),
               stage_total = o_sums + 8 * (size_t)n_parts + 64;
  int rc;
  if ((rc = ensure_stage(ctx, stage_total))) return rc;
  uint8_t* hs = static_cast<uint8_t*>(ctx->h_stage);
  ZPart* h_parts = reinterpret_cast<ZPart*>(hs + o_parts);
  ZRes* h_res = reinterpret_cast<ZRes*>(hs + o_res);
  int64_t* h_off = reinterpret_cast<int64_t*>(hs + o_off);
  int32_t* h_seg = reinterpret_cast<int32_t*>(hs + o_seg);
  int64_t* h_sums = reinterpret_cast<int64_t*>(hs + o_sums);
  record(ctx, 0);
  // ---- per-partition Adler32 / CRC32 over the compressed bytes (S3ChecksumValidationStream) ---------------------
  if (checksum_algo != S3S_CHECKSUM_NONE) {
    if ((rc = ensure(ctx, B_OFFSETS, 8 * ((size_t)n_parts + (size_t)n_ranges)))) return rc;
    if ((rc = ensure(ctx, B_SUMS, 8 * (size_t)n_parts))) return rc;
    // (run_checksum re-sizes B_SEG_START / B_PARTIAL per call: size them once for the largest range, so that no
    //  reallocation synchronises in the middle of the queue)
    int64_t max_np = 0, max_segs = 0;
    for (int32_t r = 0; r < n_ranges; r++) {
      int64_t s = 0;
      for (int32_t p = 0; p < R[r].num_partitions; p++) s += worst_segs(R[r].part_offsets[p + 1] - R[r].part_offsets[p]);
      max_segs = s > max_segs ? s : max_segs;
      max_np = R[r].num_partitions > max_np ? R[r].num_partitions : max_np;
    }
    if ((rc = ensure(ctx, B_SEG_START, 4 * (size_t)(max_np + 1)))) return rc;
    if ((rc = ensure(ctx, B_PARTIAL, 16 * (size_t)(max_segs > 0 ? max_segs : 1)))) return rc;
    int32_t pp = 0;
    for (int32_t r = 0; r < n_ranges; r++) {
      const s3s_fetch_range& k = R[r];
      const int32_t np = k.num_partitions;
      if (np == 0) continue;
      int64_t* ho = h_off + pp + r;
      int32_t* hg = h_seg + pp + r;
      int32_t segs = 0;
      for (int32_t p = 0; p < np; p++) {
        ho[p] = k.part_offsets[p];
        hg[p] = segs;
        segs += worst_segs(k.part_offsets[p + 1] - k.part_offsets[p]);
      }
      ho[np] = k.part_offsets[np];
      hg[np] = segs;
      int64_t* d_off = dev<int64_t>(ctx, B_OFFSETS) + pp + r;
      HIP_TRY(ctx, hipMemcpyAsync(d_off, ho, 8 * (size_t)(np + 1), hipMemcpyHostToDevice, ctx->stream));
      if ((rc = run_checksum(ctx, checksum_algo, k.d_comp, d_off, np, hg, dev<int64_t>(ctx, B_SUMS) + pp, k.comp_len))) return rc;
      pp += np;
    }
    HIP_TRY(ctx, hipMemcpyAsync(h_sums, ctx->buf[B_SUMS].p, 8 * (size_t)n_parts, hipMemcpyDeviceToHost, ctx->stream));
  }
  record(ctx, 1);
  // ---- single pass (round 4) -----------------------------------------------------------------------------------------
  // A Spark writer's frames carry no content size, so the two-pass form below decodes every entropy stream TWICE (sizes,
  // then bytes: 20 ms + 52 ms for 1 600 TeraSort frames).  Here every partition is decoded ONCE into a scratch area at a
  // GUESSED capacity (kGuess x its compressed size: shuffle data compresses 3 - 6 x under level 1), and a copy kernel moves
  // the partitions back to back into the caller's buffer (1 GiB: < 1 ms).  A partition that needs more than the guess
  // answers ZS_CAPACITY and the call falls back to the two-pass form as a whole (rare: runs of zeros, constant columns).
  static const int kGuess = [] {
    const char* e = getenv("S3S_ZSTD_GUESS");  // 0: always two passes
    return e ? atoi(e) : 8;
  }();
  // scratch of the single pass is bounded (advisor r4): it exists per context and never shrinks, so a call whose guess would
  // take more than the budget (default 4 GiB, S3S_ZSTD_SCRATCH_MIB) - or whose allocation fails - takes the two-pass form,
  // which sizes its literal scratch by the measured need and decodes straight into the caller's buffer
  static const int64_t kScratchBudget = [] {
    const char* e = getenv("S3S_ZSTD_SCRATCH_MIB");
    return (int64_t)(e ? atoll(e) : 4096) << 20;
  }();
  bool single = !size_only && kGuess > 0;
  if (single) {
    int64_t scr_total = 0, lit_total1 = 0;
    {
      int32_t pp = 0;
      for (int32_t r = 0; r < n_ranges; r++)
        for (int32_t p = 0; p < R[r].num_partitions; p++, pp++) {
          ZPart& z = h_parts[pp];
          z.src = R[r].d_comp + R[r].part_offsets[p];
          z.size = R[r].part_offsets[p + 1] - R[r].part_offsets[p];
          z.cap = z.size > 0 ? (int64_t)kGuess * z.size + 4096 : 0;
          z.dst = reinterpret_cast<uint8_t*>((uintptr_t)scr_total);  // (offset for now: the buffer may still move)
          z.lit_off = lit_total1;
          // the largest regenerated (Huffman-coded) literals section a block of this partition can have: a block's limit, and
          // at most 8 symbols per compressed byte (raw / RLE literals never pass through this scratch)
          const int64_t by_size = 8 * z.size + 256;
          z.lit_stride = (((by_size < (int64_t)s3s_zstd::kMaxBlock ? by_size : (int64_t)s3s_zstd::kMaxBlock) + 64) + 15) & ~int64_t(15);
          scr_total += (z.cap + 255) & ~int64_t(255);
          if (z.size > 0) lit_total1 += 2 * z.lit_stride;
        }
    }
    if (scr_total + lit_total1 > kScratchBudget) single = false;
    else {
      int e = ensure(ctx, B_ZSCRATCH, (size_t)scr_total + 256);
      if (e == S3S_OK) e = ensure(ctx, B_SLOTS, (size_t)lit_total1 + 64);
      if (e == S3S_E_NOMEM) {
        // an allocation that did not fit is not this call's verdict: the two-pass form needs less.  Give the (multi-GiB) scratch
        // back first, so that its allocations find room.  Any OTHER failure (a stream error inside ensure) is the call's verdict.
        (void)hipGetLastError();
        ctx->err[0] = 0;
        DevBuf& zb = ctx->buf[B_ZSCRATCH];
        if (zb.p) {
          (void)hipFree(zb.p);
          zb.p = nullptr;
          zb.cap = 0;
        }
        single = false;
      } else if (e != S3S_OK) {
        return e;
      }
    }
  }
  if (single) {
    if ((rc = ensure(ctx, B_RANGES, sizeof(ZPart) * (size_t)n_parts))) return rc;
    if ((rc = ensure(ctx, B_FRAMES, sizeof(ZRes) * (size_t)n_parts))) return rc;
    uint8_t* scr = dev<uint8_t>(ctx, B_ZSCRATCH);
    for (int32_t q = 0; q < n_parts; q++) h_parts[q].dst = scr + (uintptr_t)h_parts[q].dst;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->buf[B_RANGES].p, h_parts, sizeof(ZPart) * (size_t)n_parts, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(zstd_partitions_kernel, dim3((unsigned)((n_parts + 1) / 2)), dim3(kZstdThreads), 0, ctx->stream, dev<ZPart>(ctx, B_RANGES),
                       n_parts, 1, dev<uint8_t>(ctx, B_SLOTS), dev<ZRes>(ctx, B_FRAMES));
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(h_res, ctx->buf[B_FRAMES].p, sizeof(ZRes) * (size_t)n_parts, hipMemcpyDeviceToHost, ctx->stream));
    record(ctx, 2);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    // verdicts; a partition that outgrew its guess sends the whole call to the two-pass form
    bool outgrown = false;
    int64_t n_pieces = 0;
    {
      int32_t pp = 0;
      for (int32_t r = 0; r < n_ranges && !outgrown; r++) {
        s3s_fetch_range& k = R[r];
        if (checksum_algo != S3S_CHECKSUM_NONE)
          for (int32_t p = 0; p < k.num_partitions && k.status == S3S_OK; p++)
            if (h_sums[pp + p] != k.ref_checksums[p]) {
              k.status = S3S_E_CHECKSUM;
              k.bad_partition = p;
            }
        int64_t total = 0;
        for (int32_t p = 0; p < k.num_partitions; p++, pp++) {
          if (k.status != S3S_OK) continue;
          if (h_res[pp].rc == S3S_E_CAPACITY) {
            outgrown = true;
            break;
          }
          if (h_res[pp].rc != 0) {
            k.status = h_res[pp].rc;
            continue;
          }
          total += h_res[pp].total;
        }
        if (outgrown) break;
        if (k.status == S3S_OK) {
          k.out_len = total;
          if (total > k.dst_capacity) k.status = S3S_E_CAPACITY;
          else n_pieces += (total + kPieceBytes - 1) / kPieceBytes + k.num_partitions;
        }
      }
    }
    if (!outgrown) {
      std::vector<ZPiece> pcs;  // (source of an asynchronous copy: lives until the stream has been synchronised below)
      if (n_pieces > 0) {
        if ((rc = ensure(ctx, B_ZPIECES, sizeof(ZPiece) * (size_t)n_pieces))) return rc;
        pcs.reserve((size_t)n_pieces);
        int32_t pp = 0;
        for (int32_t r = 0; r < n_ranges; r++) {
          const s3s_fetch_range& k = R[r];
          int64_t at = 0;
          for (int32_t p = 0; p < k.num_partitions; p++, pp++) {
            if (k.status != S3S_OK) continue;
            for (int64_t o = 0; o < h_res[pp].total; o += kPieceBytes) {
              const int64_t n = h_res[pp].total - o < kPieceBytes ? h_res[pp].total - o : kPieceBytes;
              pcs.push_back(ZPiece{h_parts[pp].dst + o, k.d_dst + at + o, n});
            }
            at += h_res[pp].total;
          }
        }
        if (!pcs.empty()) {
          // (pageable source: the copy is staged by the runtime before the call returns)
          HIP_TRY(ctx, hipMemcpyAsync(ctx->buf[B_ZPIECES].p, pcs.data(), sizeof(ZPiece) * pcs.size(), hipMemcpyHostToDevice, ctx->stream));
          hipLaunchKernelGGL(zstd_compact_kernel, dim3((unsigned)pcs.size()), dim3(kCopyThreads), 0, ctx->stream,
                             dev<ZPiece>(ctx, B_ZPIECES), (int32_t)pcs.size());
          HIP_TRY(ctx, hipGetLastError());
        }
      }
      record(ctx, 3);
      HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
      if (ctx->profile) {
        float ms = 0;
        hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[3]); ctx->stage_ms[S3S_STAGE_TOTAL] = ms;
        hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]); ctx->stage_ms[S3S_STAGE_CHECKSUM] = ms;
        hipEventElapsedTime(&ms, ctx->ev[1], ctx->ev[2]); ctx->stage_ms[S3S_STAGE_CODEC] = ms;
        hipEventElapsedTime(&ms, ctx->ev[2], ctx->ev[3]); ctx->stage_ms[S3S_STAGE_ASSEMBLE] = ms;
      }
      if (regular_end) *regular_end = true;
      return first_error();
    }
    // outgrown: start over with the size pass (statuses back to "nothing known")
    for (int32_t r = 0; r < n_ranges; r++) {
      R[r].status = S3S_OK;
      R[r].out_len = 0;
      R[r].bad_partition = -1;
    }
    record(ctx, 1);
  }
  // ---- pass 1: sizes -------------------------------------------------------------------------------------------------
  {
    int32_t pp = 0;
    for (int32_t r = 0; r < n_ranges; r++)
      for (int32_t p = 0; p < R[r].num_partitions; p++, pp++) {
        ZPart& z = h_parts[pp];
        z.src = R[r].d_comp + R[r].part_offsets[p];
        z.size = R[r].part_offsets[p + 1] - R[r].part_offsets[p];
        z.dst = nullptr;
        z.cap = 0;
        z.lit_off = 0;
        z.lit_stride = 0;
      }
  }
  if ((rc = ensure(ctx, B_RANGES, sizeof(ZPart) * (size_t)n_parts))) return rc;
  if ((rc = ensure(ctx, B_FRAMES, sizeof(ZRes) * (size_t)n_parts))) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(ctx->buf[B_RANGES].p, h_parts, sizeof(ZPart) * (size_t)n_parts, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(zstd_partitions_kernel, dim3((unsigned)((n_parts + 1) / 2)), dim3(kZstdThreads), 0, ctx->stream, dev<ZPart>(ctx, B_RANGES),
                     n_parts, 0, (uint8_t*)nullptr, dev<ZRes>(ctx, B_FRAMES));
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(h_res, ctx->buf[B_FRAMES].p, sizeof(ZRes) * (size_t)n_parts, hipMemcpyDeviceToHost, ctx->stream));
  record(ctx, 2);
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  // ---- verdicts of the checksum and size passes, destination layout -------------------------------------------------------
  int64_t lit_total = 0;
  {
    int32_t pp = 0;
    for (int32_t r = 0; r < n_ranges; r++) {
      s3s_fetch_range& k = R[r];
      const int32_t p0 = pp;
      if (checksum_algo != S3S_CHECKSUM_NONE)
        for (int32_t p = 0; p < k.num_partitions && k.status == S3S_OK; p++)
          if (h_sums[p0 + p] != k.ref_checksums[p]) {
            k.status = S3S_E_CHECKSUM;
            k.bad_partition = p;
          }
      int64_t total = 0;
      for (int32_t p = 0; p < k.num_partitions; p++, pp++) {
        if (k.status == S3S_OK && h_res[pp].rc != 0) k.status = h_res[pp].rc;
        if (k.status != S3S_OK) continue;
        h_parts[pp].dst = k.d_dst ? k.d_dst + total : nullptr;
        h_parts[pp].cap = h_res[pp].total;
        h_parts[pp].lit_off = lit_total;
        h_parts[pp].lit_stride = (h_res[pp].lit_need + 64 + 15) & ~int64_t(15);
        total += h_res[pp].total;
        lit_total += 2 * h_parts[pp].lit_stride;
      }
      if (k.status == S3S_OK) {
        k.out_len = total;
        if (!size_only && total > k.dst_capacity) k.status = S3S_E_CAPACITY;
      }
      if (k.status != S3S_OK)  // none of its partitions takes part in pass 2
        for (int32_t q = p0; q < pp; q++) h_parts[q].size = 0;
    }
  }
  if (size_only) {
    if (ctx->profile) {
      float ms = 0;
      hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[2]); ctx->stage_ms[S3S_STAGE_TOTAL] = ms;
      hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]); ctx->stage_ms[S3S_STAGE_CHECKSUM] = ms;
      hipEventElapsedTime(&ms, ctx->ev[1], ctx->ev[2]); ctx->stage_ms[S3S_STAGE_DISCOVER] = ms;
    }
    if (regular_end) *regular_end = true;
    return first_error();
  }
  // ---- pass 2: decode ------------------------------------------------------------------------------------------------------
  if ((rc = ensure(ctx, B_SLOTS, (size_t)lit_total + 64))) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(ctx->buf[B_RANGES].p, h_parts, sizeof(ZPart) * (size_t)n_parts, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(zstd_partitions_kernel, dim3((unsigned)((n_parts + 1) / 2)), dim3(kZstdThreads), 0, ctx->stream, dev<ZPart>(ctx, B_RANGES),
                     n_parts, 1, dev<uint8_t>(ctx, B_SLOTS), dev<ZRes>(ctx, B_FRAMES));
  HIP_TRY(ctx, hipGetLastError());
  record(ctx, 3);
  HIP_TRY(ctx, hipMemcpyAsync(h_res, ctx->buf[B_FRAMES].p, sizeof(ZRes) * (size_t)n_parts, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->profile) {
    float ms = 0;
    hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[3]); ctx->stage_ms[S3S_STAGE_TOTAL] = ms;
    hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]); ctx->stage_ms[S3S_STAGE_CHECKSUM] = ms;
    hipEventElapsedTime(&ms, ctx->ev[1], ctx->ev[2]); ctx->stage_ms[S3S_STAGE_DISCOVER] = ms;
    hipEventElapsedTime(&ms, ctx->ev[2], ctx->ev[3]); ctx->stage_ms[S3S_STAGE_CODEC] = ms;
  }
  {
    int32_t pp = 0;
    for (int32_t r = 0; r < n_ranges; r++) {
      s3s_fetch_range& k = R[r];
      for (int32_t p = 0; p < k.num_partitions; p++, pp++)
        if (k.status == S3S_OK && h_parts[pp].size > 0 && (h_res[pp].rc != 0 || h_res[pp].total != h_parts[pp].cap))
          k.status = h_res[pp].rc != 0 ? h_res[pp].rc : S3S_E_BAD_FRAME;
    }
  }
  if (regular_end) *regular_end = true;
  return first_error();
}

}  // namespace s3s
