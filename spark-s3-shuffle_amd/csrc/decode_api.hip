// decode_api.hip — reduce side of the C-ABI: verify per-partition checksums, decode the codec
// streams of one fetched block range (ShuffleBlockId / ShuffleBlockBatchId).
//
// Order of checks mirrors what a reduce task observes in the reference: the checksum stream
// (S3ChecksumValidationStream.scala:54-86) sits below the decompressor
// (S3ShuffleReader.scala:102-108), so a partition whose bytes do not match referenceChecksums
// raises "Invalid checksum detected" — reported here as S3S_E_CHECKSUM with the partition
// number — and a corrupted frame raises "Stream is corrupted" (S3S_E_BAD_FRAME).
#include "s3s_ctx.h"

using namespace s3s;

namespace {

// Host walk over LZ4Block headers: decoded size of concatenated streams (sizing helper only).
int lz4block_decoded_size_host(const uint8_t* c, int64_t n, int64_t* out) {
  static const uint8_t magic[8] = {'L', 'Z', '4', 'B', 'l', 'o', 'c', 'k'};
  int64_t ip = 0, total = 0;
  while (ip < n) {
    if (n - ip < kLz4FrameHeader || memcmp(c + ip, magic, 8) != 0) return S3S_E_BAD_FRAME;
    const uint8_t* h = c + ip;
    const int method = h[8] & 0xF0, level = 10 + (h[8] & 0x0F);
    const int32_t cl = (int32_t)((uint32_t)h[9] | (uint32_t)h[10] << 8 | (uint32_t)h[11] << 16 | (uint32_t)h[12] << 24);
    const int32_t ol = (int32_t)((uint32_t)h[13] | (uint32_t)h[14] << 8 | (uint32_t)h[15] << 16 | (uint32_t)h[16] << 24);
    if ((method != 0x10 && method != 0x20) || ol < 0 || cl < 0 || ol > (1 << level) ||
        (ol == 0) != (cl == 0) || (method == 0x10 && ol != cl) || n - ip - kLz4FrameHeader < cl)
      return S3S_E_BAD_FRAME;
    total += ol;
    ip += kLz4FrameHeader + cl;
  }
  *out = total;
  return S3S_OK;
}

int lzf_decoded_size_host(const uint8_t* c, int64_t n, int64_t* out) {  // LZFInputStream: chunk headers carry both lengths
  int64_t ip = 0, total = 0;
  while (ip < n) {
    if (n - ip < 5 || c[ip] != 'Z' || c[ip + 1] != 'V' || c[ip + 2] > 1) return S3S_E_BAD_FRAME;
    const int type = c[ip + 2];
    const int64_t len = (int64_t)c[ip + 3] << 8 | c[ip + 4];
    ip += 5;
    int64_t ulen = len;
    if (type == 1) {
      if (n - ip < 2) return S3S_E_BAD_FRAME;
      ulen = (int64_t)c[ip] << 8 | c[ip + 1];
      ip += 2;
      if (len == 0 || ulen == 0) return S3S_E_BAD_FRAME;
    }
    if (len > n - ip) return S3S_E_BAD_FRAME;
    ip += len;
    total += ulen;
  }
  *out = total;
  return S3S_OK;
}

int snappy_decoded_size_host(const uint8_t* c, int64_t n, int64_t* out) {
  static const uint8_t hdr[8] = {0x82, 'S', 'N', 'A', 'P', 'P', 'Y', 0};
  int64_t ip = 0, total = 0;
  if (n == 0) {
    *out = 0;
    return S3S_OK;
  }
  if (n < kSnappyStreamHeader || memcmp(c, hdr, 8) != 0) return S3S_E_BAD_FRAME;
  ip = kSnappyStreamHeader;
  while (ip < n) {
    if (n - ip < 4) return S3S_E_BAD_FRAME;
    const uint32_t cl = (uint32_t)c[ip] << 24 | (uint32_t)c[ip + 1] << 16 | (uint32_t)c[ip + 2] << 8 | c[ip + 3];
    if (cl == 0x82534e41u) {  // concatenated stream header
      if (n - ip < kSnappyStreamHeader || memcmp(c + ip, hdr, 8) != 0) return S3S_E_BAD_FRAME;
      ip += kSnappyStreamHeader;
      continue;
    }
    ip += 4;
    if ((int64_t)cl > n - ip) return S3S_E_BAD_FRAME;
    uint32_t ulen = 0;
    int sh = 0;
    uint32_t i = 0;
    for (;; i++, sh += 7) {
      if (i >= cl || sh > 28) return S3S_E_BAD_FRAME;
      ulen |= (uint32_t)(c[ip + i] & 0x7f) << sh;
      if (!(c[ip + i] & 0x80)) break;
    }
    total += ulen;
    ip += cl;
  }
  *out = total;
  return S3S_OK;
}

}  // namespace

extern "C" {

int s3s_decompressed_size(s3s_ctx* ctx, int codec, const uint8_t* comp, int64_t comp_len,
                          int64_t* out_len) {
  if (!ctx) return S3S_E_INVALID;
  ctx->err[0] = 0;
  if (comp_len < 0 || (comp_len > 0 && !comp) || !out_len) return fail(ctx, S3S_E_INVALID, "null/invalid argument");
  int rc;
  switch (codec) {
    case S3S_CODEC_NONE:
      *out_len = comp_len;
      return S3S_OK;
    case S3S_CODEC_LZ4:
      rc = lz4block_decoded_size_host(comp, comp_len, out_len);
      break;
    case S3S_CODEC_SNAPPY:
      rc = snappy_decoded_size_host(comp, comp_len, out_len);
      break;
    case S3S_CODEC_LZF:
      rc = lzf_decoded_size_host(comp, comp_len, out_len);
      break;
    case S3S_CODEC_ZSTD: {
      // a Spark writer's frames carry no content size: the size pass of the device decoder walks them (pass 1 of
      // zstd_decompress.hip); the whole buffer is one "partition" of concatenated frames
      *out_len = 0;
      if (comp_len == 0) return S3S_OK;
      HIP_TRY(ctx, hipSetDevice(ctx->device));
      if ((rc = ensure(ctx, B_SRC, (size_t)comp_len + 64))) return rc;
      HIP_TRY(ctx, hipMemcpyAsync(ctx->buf[B_SRC].p, comp, (size_t)comp_len, hipMemcpyHostToDevice, ctx->stream));
      const int64_t offs[2] = {0, comp_len};
      s3s_fetch_range k{};
      k.d_comp = dev<uint8_t>(ctx, B_SRC);
      k.comp_len = comp_len;
      k.part_offsets = offs;
      k.num_partitions = 1;
      rc = zstd_decompress_ranges(ctx, S3S_CHECKSUM_NONE, &k, 1, true);
      *out_len = k.out_len;
      return rc;
    }
    default:
      return fail(ctx, S3S_E_INVALID, "unknown codec %d", codec);
  }
  if (rc != S3S_OK) return fail(ctx, rc, "Stream is corrupted");
  return S3S_OK;
}

int s3s_decompress_range_device(s3s_ctx* ctx, int codec, int checksum_algo, const uint8_t* d_comp,
                                int64_t comp_len, const int64_t* part_offsets,
                                const int64_t* ref_checksums, int32_t nparts, uint8_t* d_dst,
                                int64_t dst_capacity, int64_t* out_len, int32_t* out_bad_partition) {
  if (!ctx) return S3S_E_INVALID;
  ctx->err[0] = 0;
  if (out_bad_partition) *out_bad_partition = -1;
  if (out_len) *out_len = 0;
  if (nparts < 0 || !part_offsets || comp_len < 0 || dst_capacity < 0)
    return fail(ctx, S3S_E_INVALID, "null/invalid argument");
  if (codec != S3S_CODEC_NONE && codec != S3S_CODEC_LZ4 && codec != S3S_CODEC_SNAPPY && codec != S3S_CODEC_ZSTD && codec != S3S_CODEC_LZF)
    return fail(ctx, S3S_E_INVALID, "unknown codec %d", codec);
  if (checksum_algo != S3S_CHECKSUM_NONE && checksum_algo != S3S_CHECKSUM_ADLER32 && checksum_algo != S3S_CHECKSUM_CRC32 &&
      checksum_algo != S3S_CHECKSUM_CRC32C)
    return fail(ctx, S3S_E_INVALID, "Unsupported shuffle checksum algorithm: %d", checksum_algo);
  if (part_offsets[0] != 0 || part_offsets[nparts] != comp_len)
    return fail(ctx, S3S_E_INVALID, "part_offsets must span [0, comp_len]");
  for (int32_t p = 0; p < nparts; p++)
    if (part_offsets[p + 1] < part_offsets[p]) return fail(ctx, S3S_E_INVALID, "part_offsets not monotonic at %d", p);
  if (checksum_algo != S3S_CHECKSUM_NONE && nparts > 0 && !ref_checksums)
    return fail(ctx, S3S_E_INVALID, "ref_checksums is null but a checksum algorithm is selected");
  if ((comp_len > 0 && !d_comp) || (dst_capacity > 0 && !d_dst)) return fail(ctx, S3S_E_INVALID, "null data pointer");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  for (auto& v : ctx->stage_ms) v = 0;
  if (codec == S3S_CODEC_ZSTD) {  // one wavefront per partition, two passes (zstd_decompress.hip)
    s3s_fetch_range k{};
    k.d_comp = d_comp;
    k.comp_len = comp_len;
    k.part_offsets = part_offsets;
    k.ref_checksums = ref_checksums;
    k.num_partitions = nparts;
    k.d_dst = d_dst;
    k.dst_capacity = dst_capacity;
    const int zrc = zstd_decompress_ranges(ctx, checksum_algo, &k, 1, false);
    if (out_len) *out_len = k.out_len;
    if (out_bad_partition) *out_bad_partition = k.bad_partition;
    return zrc;
  }

  const int32_t n = nparts;
  const int32_t n_tiles = codec == S3S_CODEC_LZ4 ? lz4_tile_count(comp_len) : 0;
  // pinned staging: [offsets n+1][seg_start n+1][sums n][misc 8 x int64]
  const size_t off_bytes = sizeof(int64_t) * (size_t)(n + 1), seg_bytes = sizeof(int32_t) * (size_t)(n + 1);
  const size_t o_seg = (off_bytes + 15) & ~size_t(15), o_sums = (o_seg + seg_bytes + 15) & ~size_t(15),
               o_misc = (o_sums + sizeof(int64_t) * (size_t)(n > 0 ? n : 1) + 15) & ~size_t(15);
  int rc;
  if ((rc = ensure_stage(ctx, o_misc + 64))) return rc;
  uint8_t* hs = static_cast<uint8_t*>(ctx->h_stage);
  int64_t* h_off = reinterpret_cast<int64_t*>(hs);
  int32_t* h_seg = reinterpret_cast<int32_t*>(hs + o_seg);
  int64_t* h_sums = reinterpret_cast<int64_t*>(hs + o_sums);
  int64_t* h_misc = reinterpret_cast<int64_t*>(hs + o_misc);  // [0] n_frames/total, [1] status
  if ((rc = ensure(ctx, B_STATUS, 16))) return rc;
  HIP_TRY(ctx, hipMemsetAsync(ctx->buf[B_STATUS].p, 0, 16, ctx->stream));
  record(ctx, 0);

  // ---- per-partition checksum over the compressed bytes ---------------------------------------
  const bool do_sum = checksum_algo != S3S_CHECKSUM_NONE && n > 0;
  if (do_sum) {
    int64_t segs = 0;
    for (int32_t p = 0; p < n; p++) {
      h_off[p] = part_offsets[p];
      h_seg[p] = (int32_t)segs;
      segs += worst_segs(part_offsets[p + 1] - part_offsets[p]);
      if (segs > 0x7fffff00ll) return fail(ctx, S3S_E_UNSUPPORTED, "range too large for one call");
    }
    h_off[n] = part_offsets[n];
    h_seg[n] = (int32_t)segs;
    if ((rc = ensure(ctx, B_OFFSETS, off_bytes))) return rc;
    if ((rc = ensure(ctx, B_SUMS, sizeof(int64_t) * (size_t)n))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->buf[B_OFFSETS].p, h_off, off_bytes, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = run_checksum(ctx, checksum_algo, d_comp, dev<int64_t>(ctx, B_OFFSETS), n, h_seg,
                           dev<int64_t>(ctx, B_SUMS), comp_len)))
      return rc;
    HIP_TRY(ctx, hipMemcpyAsync(h_sums, ctx->buf[B_SUMS].p, sizeof(int64_t) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  }
  record(ctx, 1);  // ev1: checksum done

  auto verify_sums = [&]() -> int {
    if (!do_sum) return S3S_OK;
    for (int32_t p = 0; p < n; p++)
      if (h_sums[p] != ref_checksums[p]) {
        if (out_bad_partition) *out_bad_partition = p;
        return fail(ctx, S3S_E_CHECKSUM, "Invalid checksum detected for partition %d of the range", p);
      }
    return S3S_OK;
  };
  auto finish_profile = [&](int last_ev) {
    if (!ctx->profile) return;
    float ms = 0;
    hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[last_ev]); ctx->stage_ms[S3S_STAGE_TOTAL] = ms;
    hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]); ctx->stage_ms[S3S_STAGE_CHECKSUM] = ms;
    if (last_ev >= 2) { hipEventElapsedTime(&ms, ctx->ev[1], ctx->ev[2]); ctx->stage_ms[S3S_STAGE_DISCOVER] = ms; }
    if (last_ev >= 3) { hipEventElapsedTime(&ms, ctx->ev[2], ctx->ev[3]); ctx->stage_ms[S3S_STAGE_CODEC] = ms; }
  };

  if (codec == S3S_CODEC_NONE) {
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if ((rc = verify_sums())) return rc;
    if (comp_len > dst_capacity) return fail(ctx, S3S_E_CAPACITY, "dst_capacity %lld < %lld", (long long)dst_capacity, (long long)comp_len);
    if (comp_len > 0) HIP_TRY(ctx, hipMemcpyAsync(d_dst, d_comp, (size_t)comp_len, hipMemcpyDeviceToDevice, ctx->stream));
    record(ctx, 2);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    finish_profile(2);
    if (out_len) *out_len = comp_len;
    return S3S_OK;
  }

  if (codec == S3S_CODEC_SNAPPY || codec == S3S_CODEC_LZF) {
    // ---- SnappyInputStream / LZFInputStream framing: count chunks per partition, scan, emit frames, decode ---------
    const int cf = codec == S3S_CODEC_LZF ? kChunkLzf : kChunkSnappy;
    if (comp_len == 0 || n == 0) {
      HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
      if ((rc = verify_sums())) return rc;
      return S3S_OK;
    }
    if (!do_sum) {
      for (int32_t p = 0; p <= n; p++) h_off[p] = part_offsets[p];
      if ((rc = ensure(ctx, B_OFFSETS, off_bytes))) return rc;
      HIP_TRY(ctx, hipMemcpyAsync(ctx->buf[B_OFFSETS].p, h_off, off_bytes, hipMemcpyHostToDevice, ctx->stream));
    }
    const size_t cnt_bytes = (sizeof(uint32_t) * (size_t)(n + 1) + 15) & ~size_t(15);
    if ((rc = ensure(ctx, B_PART_NFRAMES, cnt_bytes + sizeof(int64_t) * (size_t)(n + 2)))) return rc;
    uint32_t* d_cnt = dev<uint32_t>(ctx, B_PART_NFRAMES);
    int64_t* d_base = reinterpret_cast<int64_t*>(dev<uint8_t>(ctx, B_PART_NFRAMES) + cnt_bytes);
    launch_snappy_count_frames(d_comp, dev<int64_t>(ctx, B_OFFSETS), n, d_cnt, dev<int32_t>(ctx, B_STATUS), ctx->stream, cf);
    launch_scan_u32(d_cnt, n, d_base, ctx->stream);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(&h_misc[0], d_base + n, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(&h_misc[1], ctx->buf[B_STATUS].p, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if ((rc = verify_sums())) return rc;
    if (*reinterpret_cast<int32_t*>(&h_misc[1]) != 0) return fail(ctx, S3S_E_BAD_FRAME, "Stream is corrupted (%s chunk chain)", cf == kChunkLzf ? "lzf" : "snappy");
    const int64_t n_frames = h_misc[0];
    if (n_frames > 0x7fffff00ll) return fail(ctx, S3S_E_UNSUPPORTED, "too many frames in one call");
    if ((rc = ensure(ctx, B_FRAMES, sizeof(Frame) * (size_t)(n_frames + 1)))) return rc;
    if ((rc = ensure(ctx, B_ITEM_SIZE, sizeof(uint32_t) * (size_t)(n_frames + 1)))) return rc;
    if ((rc = ensure(ctx, B_FRAME_OUT, sizeof(int64_t) * (size_t)(n_frames + 1)))) return rc;
    launch_snappy_emit_frames(d_comp, dev<int64_t>(ctx, B_OFFSETS), n, d_base, dev<Frame>(ctx, B_FRAMES),
                              dev<uint32_t>(ctx, B_ITEM_SIZE), dev<int32_t>(ctx, B_STATUS), ctx->stream, cf);
    launch_scan_u32(dev<uint32_t>(ctx, B_ITEM_SIZE), n_frames, dev<int64_t>(ctx, B_FRAME_OUT), ctx->stream);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(&h_misc[0], dev<int64_t>(ctx, B_FRAME_OUT) + n_frames, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    record(ctx, 2);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const int64_t total = h_misc[0];
    if (out_len) *out_len = total;
    if (total > dst_capacity)
      return fail(ctx, S3S_E_CAPACITY, "dst_capacity %lld < %lld decoded bytes", (long long)dst_capacity, (long long)total);
    launch_snappy_decompress(d_comp, dev<Frame>(ctx, B_FRAMES), (int32_t)n_frames, dev<int64_t>(ctx, B_FRAME_OUT),
                             d_dst, dev<int32_t>(ctx, B_STATUS), ctx->lz4_decode_variant, ctx->stream, cf);
    HIP_TRY(ctx, hipGetLastError());
    record(ctx, 3);
    HIP_TRY(ctx, hipMemcpyAsync(&h_misc[1], ctx->buf[B_STATUS].p, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    finish_profile(3);
    const int32_t st = *reinterpret_cast<int32_t*>(&h_misc[1]);
    if (st == S3S_E_UNSUPPORTED) return fail(ctx, S3S_E_UNSUPPORTED, "snappy block larger than %d bytes", kMaxBlock);
    if (st != 0) return fail(ctx, S3S_E_BAD_FRAME, "Stream is corrupted");
    return S3S_OK;
  }

  // ---- LZ4Block: discover the frame chain --------------------------------------------------------
  if (comp_len == 0) {
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if ((rc = verify_sums())) return rc;
    return S3S_OK;
  }
  // tile workspace lives in B_FRAME_OUT's neighbours: [spec_entry][spec_exit][true_entry][frame_base] int64, [spec_count] int32
  const size_t tile_i64 = sizeof(int64_t) * (size_t)(n_tiles + 1);
  if ((rc = ensure(ctx, B_PART_NFRAMES, 4 * tile_i64 + sizeof(int32_t) * (size_t)(n_tiles + 1)))) return rc;
  int64_t* d_spec_entry = dev<int64_t>(ctx, B_PART_NFRAMES);
  int64_t* d_spec_exit = d_spec_entry + (n_tiles + 1);
  int64_t* d_true_entry = d_spec_exit + (n_tiles + 1);
  int64_t* d_frame_base = d_true_entry + (n_tiles + 1);
  int32_t* d_spec_count = reinterpret_cast<int32_t*>(d_frame_base + (n_tiles + 1));
  launch_lz4_discover(d_comp, comp_len, n_tiles, d_spec_entry, d_spec_exit, d_spec_count, d_true_entry,
                      d_frame_base, dev<int32_t>(ctx, B_STATUS), ctx->stream);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(&h_misc[0], d_frame_base + n_tiles, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(&h_misc[1], ctx->buf[B_STATUS].p, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if ((rc = verify_sums())) return rc;
  if (*reinterpret_cast<int32_t*>(&h_misc[1]) != 0) return fail(ctx, S3S_E_BAD_FRAME, "Stream is corrupted (frame chain)");
  const int64_t n_frames = h_misc[0];
  if (n_frames > 0x7fffff00ll) return fail(ctx, S3S_E_UNSUPPORTED, "too many frames in one call");

  if ((rc = ensure(ctx, B_FRAMES, sizeof(Frame) * (size_t)(n_frames + 1)))) return rc;
  if ((rc = ensure(ctx, B_ITEM_SIZE, sizeof(uint32_t) * (size_t)(n_frames + 1)))) return rc;
  if ((rc = ensure(ctx, B_FRAME_OUT, sizeof(int64_t) * (size_t)(n_frames + 1)))) return rc;
  launch_lz4_emit_frames(d_comp, comp_len, n_tiles, d_true_entry, d_frame_base, dev<Frame>(ctx, B_FRAMES),
                         dev<uint32_t>(ctx, B_ITEM_SIZE), n_frames, dev<int64_t>(ctx, B_FRAME_OUT),
                         dev<int32_t>(ctx, B_STATUS), ctx->stream);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(&h_misc[0], dev<int64_t>(ctx, B_FRAME_OUT) + n_frames, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(&h_misc[1], ctx->buf[B_STATUS].p, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  record(ctx, 2);  // ev2: discovery done
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (*reinterpret_cast<int32_t*>(&h_misc[1]) != 0) return fail(ctx, S3S_E_BAD_FRAME, "Stream is corrupted (frame header)");
  const int64_t total = h_misc[0];
  if (out_len) *out_len = total;
  if (total > dst_capacity)
    return fail(ctx, S3S_E_CAPACITY, "dst_capacity %lld < %lld decoded bytes", (long long)dst_capacity, (long long)total);

  launch_lz4_decompress(d_comp, dev<Frame>(ctx, B_FRAMES), (int32_t)n_frames, dev<int64_t>(ctx, B_FRAME_OUT),
                        d_dst, dev<int32_t>(ctx, B_STATUS), ctx->lz4_decode_variant, ctx->stream,
                        ctx->profile ? ctx->ev_hash : nullptr);
  HIP_TRY(ctx, hipGetLastError());
  record(ctx, 3);  // ev3: decode done
  HIP_TRY(ctx, hipMemcpyAsync(&h_misc[1], ctx->buf[B_STATUS].p, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  finish_profile(3);
  if (ctx->profile) {  // codec = the decode kernel alone, hash = the frame checks (ev_hash sits between them)
    float ms = 0;
    if (hipEventElapsedTime(&ms, ctx->ev[2], ctx->ev_hash) == hipSuccess) ctx->stage_ms[S3S_STAGE_CODEC] = ms;
    if (hipEventElapsedTime(&ms, ctx->ev_hash, ctx->ev[3]) == hipSuccess) ctx->stage_ms[S3S_STAGE_HASH] = ms;
  }
  int32_t st = *reinterpret_cast<int32_t*>(&h_misc[1]);
  if (st == S3S_E_UNSUPPORTED && ctx->lz4_decode_variant != 3) {
    // a frame above kBatchMaxBlock (32 MiB: larger than any LZ4BlockOutputStream block; frames of 32 KiB .. 32 MiB written
    // with a larger spark.io.compression.lz4.blockSize are the batch decoder's since round 4): the ring decoder takes any size
    HIP_TRY(ctx, hipMemsetAsync(ctx->buf[B_STATUS].p, 0, 16, ctx->stream));
    launch_lz4_decompress(d_comp, dev<Frame>(ctx, B_FRAMES), (int32_t)n_frames, dev<int64_t>(ctx, B_FRAME_OUT), d_dst,
                          dev<int32_t>(ctx, B_STATUS), 3, ctx->stream);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(&h_misc[1], ctx->buf[B_STATUS].p, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    st = *reinterpret_cast<int32_t*>(&h_misc[1]);
  }
  if (st == S3S_E_UNSUPPORTED) return fail(ctx, S3S_E_UNSUPPORTED, "LZ4Block frame larger than %d bytes", kBatchMaxBlock);
  if (st != 0) return fail(ctx, S3S_E_BAD_FRAME, "Stream is corrupted");
  return S3S_OK;
}

int s3s_decompress_range(s3s_ctx* ctx, int codec, int checksum_algo, const uint8_t* comp,
                         int64_t comp_len, const int64_t* part_offsets, const int64_t* ref_checksums,
                         int32_t nparts, uint8_t* dst, int64_t dst_capacity, int64_t* out_len,
                         int32_t* out_bad_partition) {
  if (!ctx) return S3S_E_INVALID;
  ctx->err[0] = 0;
  if (comp_len < 0 || (comp_len > 0 && !comp) || dst_capacity < 0 || (dst_capacity > 0 && !dst))
    return fail(ctx, S3S_E_INVALID, "null/invalid host buffer");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int rc;
  // size the device destination from the DECODED size (a walk over the frame headers on the host), not from the
  // caller's capacity: a large reusable output buffer must not turn into an equally large hipMalloc
  int64_t need = dst_capacity;
  {
    int64_t decoded = 0;
    if (codec == S3S_CODEC_ZSTD)
      ;  // (its size pass runs on the device inside the call below; the caller sized dst with s3s_decompressed_size)
    else if (codec != S3S_CODEC_NONE && s3s_decompressed_size(ctx, codec, comp, comp_len, &decoded) == S3S_OK && decoded < need)
      need = decoded;
    else if (codec == S3S_CODEC_NONE && comp_len < need)
      need = comp_len;
    ctx->err[0] = 0;  // (a corrupt chain is reported by the device path below, with its own message)
  }
  if ((rc = ensure(ctx, B_SRC, (size_t)comp_len + 64))) return rc;
  if ((rc = ensure(ctx, B_DST, (size_t)need + 64))) return rc;
  if (comp_len > 0)
    HIP_TRY(ctx, hipMemcpyAsync(ctx->buf[B_SRC].p, comp, (size_t)comp_len, hipMemcpyHostToDevice, ctx->stream));
  int64_t total = 0;
  rc = s3s_decompress_range_device(ctx, codec, checksum_algo, dev<uint8_t>(ctx, B_SRC), comp_len, part_offsets,
                                   ref_checksums, nparts, dev<uint8_t>(ctx, B_DST), need, &total,
                                   out_bad_partition);
  if (rc == S3S_E_CAPACITY && need < dst_capacity) rc = fail(ctx, S3S_E_BAD_FRAME, "Stream is corrupted (decoded size changed)");
  if (out_len) *out_len = total;
  if (rc != S3S_OK) return rc;
  if (total > 0) HIP_TRY(ctx, hipMemcpy(dst, ctx->buf[B_DST].p, (size_t)total, hipMemcpyDeviceToHost));
  return S3S_OK;
}

}  // extern "C"

// ---- batched reduce side -----------------------------------------------------------------------------------
namespace {
// frames of one range: payload offsets and output offsets become absolute addresses, so that ONE decode launch
// (comp = dst = nullptr) covers the frames of every range; a range that failed an earlier check gets empty frames
__global__ void rebase_frames_kernel(Frame* frames, const int64_t* out_rel, int64_t* out_abs, int32_t n,
                                     int64_t comp_base, int64_t dst_base, int32_t skip) {
  const int32_t i = (int32_t)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= n) return;
  if (skip) {
    frames[i].comp_off = 0;
    frames[i].comp_len = 0;
    frames[i].orig_len = 0;
    frames[i].method = 0x10;
    out_abs[i] = dst_base;
    return;
  }
  frames[i].comp_off += comp_base;
  out_abs[i] = out_rel[i] + dst_base;
}
}  // namespace

extern "C" int s3s_decompress_ranges_batch_device(s3s_ctx* ctx, int codec, int checksum_algo, s3s_fetch_range* R,
                                                  int32_t n_ranges) {
  if (!ctx) return S3S_E_INVALID;
  ctx->err[0] = 0;
  if (n_ranges < 0 || (n_ranges > 0 && !R)) return fail(ctx, S3S_E_INVALID, "null range array or negative count");
  BatchVerdict<s3s_fetch_range> verdict(R, n_ranges);  // stamped before the argument checks (advisor r4)
  if (codec != S3S_CODEC_NONE && codec != S3S_CODEC_LZ4 && codec != S3S_CODEC_SNAPPY && codec != S3S_CODEC_ZSTD && codec != S3S_CODEC_LZF)
    return fail(ctx, S3S_E_INVALID, "unknown codec %d", codec);
  if (checksum_algo != S3S_CHECKSUM_NONE && checksum_algo != S3S_CHECKSUM_ADLER32 && checksum_algo != S3S_CHECKSUM_CRC32 &&
      checksum_algo != S3S_CHECKSUM_CRC32C)
    return fail(ctx, S3S_E_INVALID, "Unsupported shuffle checksum algorithm: %d", checksum_algo);
  auto single = [&](s3s_fetch_range& r) {
    r.status = s3s_decompress_range_device(ctx, codec, checksum_algo, r.d_comp, r.comp_len, r.part_offsets,
                                           r.ref_checksums, r.num_partitions, r.d_dst, r.dst_capacity, &r.out_len,
                                           &r.bad_partition);
  };
  auto first_error = [&]() -> int {
    for (int32_t r = 0; r < n_ranges; r++)
      if (R[r].status != S3S_OK)
        return fail(ctx, R[r].status, "range %d of the batch failed (%s)", r,
                    R[r].status == S3S_E_CHECKSUM ? "Invalid checksum detected" : R[r].status == S3S_E_CAPACITY ? "dst_capacity too small" : "Stream is corrupted");
    return S3S_OK;
  };
  if (n_ranges == 0) return S3S_OK;
  if (codec == S3S_CODEC_NONE || n_ranges == 1) {  // nothing to batch
    for (int32_t r = 0; r < n_ranges; r++) single(R[r]);
    return verdict.finish(first_error());
  }
  // ---- validate, count ------------------------------------------------------------------------------------------
  int64_t n_parts = 0, n_segs = 0;
  for (int32_t r = 0; r < n_ranges; r++) {
    s3s_fetch_range& k = R[r];
    k.status = S3S_OK;
    k.out_len = 0;
    k.bad_partition = -1;
    if (k.num_partitions < 0 || !k.part_offsets || k.comp_len < 0 || k.dst_capacity < 0)
      return fail(ctx, S3S_E_INVALID, "range %d: null/invalid argument", r);
    if (k.part_offsets[0] != 0 || k.part_offsets[k.num_partitions] != k.comp_len)
      return fail(ctx, S3S_E_INVALID, "range %d: part_offsets must span [0, comp_len]", r);
    for (int32_t p = 0; p < k.num_partitions; p++) {
      if (k.part_offsets[p + 1] < k.part_offsets[p]) return fail(ctx, S3S_E_INVALID, "range %d: part_offsets not monotonic at %d", r, p);
      n_segs += worst_segs(k.part_offsets[p + 1] - k.part_offsets[p]);
    }
    if (checksum_algo != S3S_CHECKSUM_NONE && k.num_partitions > 0 && !k.ref_checksums)
      return fail(ctx, S3S_E_INVALID, "range %d: ref_checksums is null but a checksum algorithm is selected", r);
    if ((k.comp_len > 0 && !k.d_comp) || (k.dst_capacity > 0 && !k.d_dst)) return fail(ctx, S3S_E_INVALID, "range %d: null data pointer", r);
    n_parts += k.num_partitions;
  }
  if (n_segs > 0x7fffff00ll || n_parts > 0x7fffff00ll) return fail(ctx, S3S_E_UNSUPPORTED, "batch too large for one call");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (codec == S3S_CODEC_ZSTD)  // the partitions of every range in one launch per pass: frames in flight are the throughput
  {
    bool regular = false;
    const int zrc = zstd_decompress_ranges(ctx, checksum_algo, R, n_ranges, false, &regular);
    return regular ? verdict.finish(zrc) : zrc;
  }
  for (auto& v : ctx->stage_ms) v = 0;
  const bool do_sum = checksum_algo != S3S_CHECKSUM_NONE;
  const size_t np1 = (size_t)n_parts + (size_t)n_ranges;  // sum of (n_r + 1)
  auto al = [](size_t x) { return (x + 15) & ~size_t(15); };
  // pinned staging: [offsets][seg_start][sums][n_frames per range][total per range][status: n_ranges + 1]
  //                 [range descriptors][tile -> range map][results: 2 per range]      (LZ4: batched discovery)
  int64_t total_tiles64 = 0;
  if (codec == S3S_CODEC_LZ4)
    for (int32_t r = 0; r < n_ranges; r++) total_tiles64 += lz4_tile_count(R[r].comp_len);
  if (total_tiles64 > 0x7fffff00ll) return fail(ctx, S3S_E_UNSUPPORTED, "batch too large for one call");
  const int32_t total_tiles = (int32_t)total_tiles64;
  const size_t o_off = 0, o_seg = al(o_off + 8 * np1), o_sums = al(o_seg + 4 * np1), o_nf = al(o_sums + 8 * ((size_t)n_parts + 1)),
               o_tot = al(o_nf + 8 * (size_t)n_ranges), o_st = al(o_tot + 8 * (size_t)n_ranges),
               o_desc = al(o_st + 4 * ((size_t)n_ranges + 1)), o_map = al(o_desc + sizeof(LzRange) * (size_t)n_ranges),
               o_res = al(o_map + 4 * ((size_t)total_tiles + 1)), o_tails = al(o_res + 16 * (size_t)n_ranges + 16),
               stage_total = o_tails + sizeof(TaskTail) * (size_t)n_ranges + 16;
  int rc;
  if ((rc = ensure_stage(ctx, stage_total))) return rc;
  uint8_t* hs = static_cast<uint8_t*>(ctx->h_stage);
  int64_t* h_off = reinterpret_cast<int64_t*>(hs + o_off);
  int32_t* h_seg = reinterpret_cast<int32_t*>(hs + o_seg);
  int64_t* h_sums = reinterpret_cast<int64_t*>(hs + o_sums);
  int64_t* h_nf = reinterpret_cast<int64_t*>(hs + o_nf);
  int64_t* h_tot = reinterpret_cast<int64_t*>(hs + o_tot);
  int32_t* h_st = reinterpret_cast<int32_t*>(hs + o_st);
  LzRange* h_desc = reinterpret_cast<LzRange*>(hs + o_desc);
  int32_t* h_map = reinterpret_cast<int32_t*>(hs + o_map);
  int64_t* h_res = reinterpret_cast<int64_t*>(hs + o_res);
  // device workspaces, all sized up front (no reallocation between the queued kernels)
  // discovery workspace per range: LZ4 4 x (tiles + 1) int64 + (tiles + 1) int32; Snappy (n + 1) u32 + (n + 2) int64
  std::vector<size_t> ws_off((size_t)n_ranges + 1), first_part((size_t)n_ranges + 1), first_seg((size_t)n_ranges + 1);
  {
    size_t w = 0, pp = 0, sg = 0;
    for (int32_t r = 0; r < n_ranges; r++) {
      ws_off[(size_t)r] = w;
      first_part[(size_t)r] = pp;
      first_seg[(size_t)r] = sg;
      const s3s_fetch_range& k = R[r];
      if (codec == S3S_CODEC_LZ4) {
        const size_t t1 = (size_t)lz4_tile_count(k.comp_len) + 1;
        w += al(4 * 8 * t1 + 4 * t1);
      } else {
        w += al(al(4 * ((size_t)k.num_partitions + 1)) + 8 * ((size_t)k.num_partitions + 2));
      }
      int32_t seg_r = 0;
      for (int32_t p = 0; p < k.num_partitions; p++) {
        h_off[pp + (size_t)r + (size_t)p] = k.part_offsets[p];
        h_seg[pp + (size_t)r + (size_t)p] = seg_r;
        seg_r += worst_segs(k.part_offsets[p + 1] - k.part_offsets[p]);
      }
      h_off[pp + (size_t)r + (size_t)k.num_partitions] = k.comp_len;
      h_seg[pp + (size_t)r + (size_t)k.num_partitions] = seg_r;
      pp += (size_t)k.num_partitions;
      sg += (size_t)seg_r;
    }
    ws_off[(size_t)n_ranges] = w;
    first_part[(size_t)n_ranges] = pp;
    first_seg[(size_t)n_ranges] = sg;
  }
  if ((rc = ensure(ctx, B_OFFSETS, 8 * np1))) return rc;
  if ((rc = ensure(ctx, B_SEG_START, 4 * np1))) return rc;
  if ((rc = ensure(ctx, B_SUMS, 8 * ((size_t)n_parts + 1)))) return rc;
  if ((rc = ensure(ctx, B_PARTIAL, 16 * (first_seg[(size_t)n_ranges] > 0 ? first_seg[(size_t)n_ranges] : 1)))) return rc;
  if ((rc = ensure(ctx, B_PART_NFRAMES, ws_off[(size_t)n_ranges] + 16))) return rc;
  if ((rc = ensure(ctx, B_STATUS, 4 * ((size_t)n_ranges + 1) + 16))) return rc;
  if ((rc = ensure(ctx, B_RANGES, sizeof(LzRange) * (size_t)n_ranges + 16))) return rc;
  if ((rc = ensure(ctx, B_TILE_RANGE, 4 * ((size_t)total_tiles + 1) + 16))) return rc;
  if ((rc = ensure(ctx, B_REF_SUMS, 16 * (size_t)n_ranges + 16))) return rc;
  HIP_TRY(ctx, hipMemsetAsync(ctx->buf[B_STATUS].p, 0, 4 * ((size_t)n_ranges + 1) + 16, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(ctx->buf[B_OFFSETS].p, h_off, 8 * np1, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(ctx->buf[B_SEG_START].p, h_seg, 4 * np1, hipMemcpyHostToDevice, ctx->stream));
  record(ctx, 0);
  int32_t* d_status = dev<int32_t>(ctx, B_STATUS);
  // ---- phase 1: checksums + frame discovery of every range, one wait ---------------------------------------------
  if (do_sum && n_parts > 0) {  // the checksums of every range: one segments launch + one combine launch (TaskTail, s3s_internal.h)
    if ((rc = ensure(ctx, B_TAILS, sizeof(TaskTail) * (size_t)n_ranges + 16))) return rc;
    TaskTail* h_tails = reinterpret_cast<TaskTail*>(hs + o_tails);
    for (int32_t r = 0; r < n_ranges; r++) {
      const s3s_fetch_range& k = R[r];
      TaskTail& d = h_tails[r];
      memset(&d, 0, sizeof(d));
      d.first_pp = (int32_t)(first_part[(size_t)r] + (size_t)r);
      d.n_parts = k.num_partitions;
      d.first_part = (int32_t)first_part[(size_t)r];
      d.first_seg = (int32_t)first_seg[(size_t)r];
      d.n_segs = (int32_t)(first_seg[(size_t)r + 1] - first_seg[(size_t)r]);
      d.data = k.d_comp;
      d.data_len = k.comp_len;
    }
    HIP_TRY(ctx, hipMemcpyAsync(ctx->buf[B_TAILS].p, h_tails, sizeof(TaskTail) * (size_t)n_ranges, hipMemcpyHostToDevice, ctx->stream));
    launch_checksum_batch(checksum_algo, dev<TaskTail>(ctx, B_TAILS), n_ranges, (int32_t)first_seg[(size_t)n_ranges], (int32_t)n_parts,
                          dev<int64_t>(ctx, B_OFFSETS), dev<int32_t>(ctx, B_SEG_START), ctx->buf[B_TABLES].p,
                          dev<uint32_t>(ctx, B_PARTIAL), dev<int64_t>(ctx, B_SUMS), ctx->stream);
  }
  record(ctx, 1);
  if (codec == S3S_CODEC_LZ4) {
    // one speculate launch over the tiles of every range, one resolve + count launch with a wavefront per range
    int32_t t0 = 0;
    for (int32_t r = 0; r < n_ranges; r++) {
      const s3s_fetch_range& k = R[r];
      const int32_t nt = (k.comp_len > 0 && k.num_partitions > 0) ? lz4_tile_count(k.comp_len) : 0;
      uint8_t* ws = dev<uint8_t>(ctx, B_PART_NFRAMES) + ws_off[(size_t)r];
      LzRange& d = h_desc[r];
      memset(&d, 0, sizeof(d));
      d.comp = k.d_comp;
      d.comp_len = k.comp_len;
      d.n_tiles = nt;
      d.tile0 = t0;
      const int32_t cap_t = lz4_tile_count(k.comp_len) + 1;
      d.spec_entry = reinterpret_cast<int64_t*>(ws);
      d.spec_exit = d.spec_entry + cap_t;
      d.true_entry = d.spec_exit + cap_t;
      d.frame_base = d.true_entry + cap_t;
      d.spec_count = reinterpret_cast<int32_t*>(d.frame_base + cap_t);
      d.status = d_status + r;
      d.result = dev<int64_t>(ctx, B_REF_SUMS) + 2 * (size_t)r;
      d.dst_base = (int64_t)reinterpret_cast<uintptr_t>(k.d_dst);
      d.dst_capacity = k.dst_capacity;
      for (int32_t t = 0; t < nt; t++) h_map[t0 + t] = r;
      t0 += nt;
    }
    const int32_t used_tiles = t0;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->buf[B_RANGES].p, h_desc, sizeof(LzRange) * (size_t)n_ranges, hipMemcpyHostToDevice, ctx->stream));
    if (used_tiles > 0)
      HIP_TRY(ctx, hipMemcpyAsync(ctx->buf[B_TILE_RANGE].p, h_map, 4 * (size_t)used_tiles, hipMemcpyHostToDevice, ctx->stream));
    launch_lz4_discover_batch(dev<LzRange>(ctx, B_RANGES), n_ranges, dev<int32_t>(ctx, B_TILE_RANGE), used_tiles, ctx->stream);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(h_res, ctx->buf[B_REF_SUMS].p, 16 * (size_t)n_ranges, hipMemcpyDeviceToHost, ctx->stream));
  } else {
    for (int32_t r = 0; r < n_ranges; r++) {
      const s3s_fetch_range& k = R[r];
      h_nf[r] = 0;
      if (k.comp_len == 0 || k.num_partitions == 0) continue;
      uint8_t* ws = dev<uint8_t>(ctx, B_PART_NFRAMES) + ws_off[(size_t)r];
      const size_t pp = first_part[(size_t)r] + (size_t)r;
      uint32_t* d_cnt = reinterpret_cast<uint32_t*>(ws);
      int64_t* d_base = reinterpret_cast<int64_t*>(ws + al(4 * ((size_t)k.num_partitions + 1)));
      launch_snappy_count_frames(k.d_comp, dev<int64_t>(ctx, B_OFFSETS) + pp, k.num_partitions, d_cnt, d_status + r, ctx->stream,
                                 codec == S3S_CODEC_LZF ? kChunkLzf : kChunkSnappy);
      launch_scan_u32(d_cnt, k.num_partitions, d_base, ctx->stream);
      HIP_TRY(ctx, hipMemcpyAsync(&h_nf[r], d_base + k.num_partitions, 8, hipMemcpyDeviceToHost, ctx->stream));
    }
  }
  HIP_TRY(ctx, hipGetLastError());
  if (do_sum && n_parts > 0)
    HIP_TRY(ctx, hipMemcpyAsync(h_sums, ctx->buf[B_SUMS].p, 8 * (size_t)n_parts, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(h_st, d_status, 4 * (size_t)n_ranges, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  int64_t total_frames = 0;
  std::vector<int64_t> first_frame((size_t)n_ranges + 1);
  for (int32_t r = 0; r < n_ranges; r++) {
    s3s_fetch_range& k = R[r];
    first_frame[(size_t)r] = total_frames;
    if (codec == S3S_CODEC_LZ4) h_nf[r] = h_res[2 * r];
    if (do_sum)
      for (int32_t p = 0; p < k.num_partitions; p++)
        if (h_sums[first_part[(size_t)r] + (size_t)p] != k.ref_checksums[p]) {
          k.status = S3S_E_CHECKSUM;
          k.bad_partition = p;
          break;
        }
    if (k.status == S3S_OK && h_st[r] != 0) k.status = S3S_E_BAD_FRAME;
    if (k.status != S3S_OK) h_nf[r] = 0;  // no frames of this range take part in what follows
    total_frames += h_nf[r];
  }
  first_frame[(size_t)n_ranges] = total_frames;
  if (total_frames > 0x7fffff00ll) return fail(ctx, S3S_E_UNSUPPORTED, "too many frames in one call");
  // ---- phase 2: frame records + output offsets of every range, then ONE decode launch ----------------------------------
  const size_t nf1 = (size_t)total_frames + (size_t)n_ranges + 1;
  if ((rc = ensure(ctx, B_FRAMES, sizeof(Frame) * nf1))) return rc;
  if ((rc = ensure(ctx, B_ITEM_SIZE, 4 * nf1))) return rc;
  if ((rc = ensure(ctx, B_FRAME_OUT, 8 * nf1))) return rc;
  if ((rc = ensure(ctx, B_ITEM_OFF, 8 * nf1))) return rc;
  int32_t* d_dec_status = d_status + n_ranges;
  bool lz4_split = false;  // profile: ev_hash sits between the decode kernel and the frame-check kernel
  if (codec == S3S_CODEC_LZ4) {
    int32_t used_tiles = 0;
    for (int32_t r = 0; r < n_ranges; r++) {
      LzRange& d = h_desc[r];
      const int64_t f0 = first_frame[(size_t)r];
      d.n_frames = h_nf[r];
      d.skip = R[r].status != S3S_OK ? 1 : 0;
      d.frames = dev<Frame>(ctx, B_FRAMES) + f0;
      d.frame_orig = dev<uint32_t>(ctx, B_ITEM_SIZE) + f0 + r;
      d.frame_out = dev<int64_t>(ctx, B_FRAME_OUT) + f0 + r;
      d.out_abs = dev<int64_t>(ctx, B_ITEM_OFF) + f0;
      used_tiles += d.n_tiles;
    }
    HIP_TRY(ctx, hipMemcpyAsync(ctx->buf[B_RANGES].p, h_desc, sizeof(LzRange) * (size_t)n_ranges, hipMemcpyHostToDevice, ctx->stream));
    launch_lz4_frames_batch(dev<LzRange>(ctx, B_RANGES), n_ranges, dev<int32_t>(ctx, B_TILE_RANGE), used_tiles, ctx->stream);
    HIP_TRY(ctx, hipGetLastError());
    record(ctx, 2);
    if (total_frames > 0)
      launch_lz4_decompress(nullptr, dev<Frame>(ctx, B_FRAMES), (int32_t)total_frames, dev<int64_t>(ctx, B_ITEM_OFF), nullptr,
                            d_dec_status, ctx->lz4_decode_variant, ctx->stream, ctx->profile ? ctx->ev_hash : nullptr);
    else if (ctx->profile)
      HIP_TRY(ctx, hipEventRecord(ctx->ev_hash, ctx->stream));
    lz4_split = ctx->profile != 0;
    HIP_TRY(ctx, hipGetLastError());
    record(ctx, 3);
    HIP_TRY(ctx, hipMemcpyAsync(h_res, ctx->buf[B_REF_SUMS].p, 16 * (size_t)n_ranges, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(h_st, d_status, 4 * ((size_t)n_ranges + 1), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int32_t r = 0; r < n_ranges; r++) {
      s3s_fetch_range& k = R[r];
      if (k.status != S3S_OK) continue;
      if (h_st[r] != 0) { k.status = S3S_E_BAD_FRAME; continue; }
      k.out_len = h_res[2 * r + 1];
      if (k.out_len > k.dst_capacity) k.status = S3S_E_CAPACITY;
    }
  } else {
    for (int32_t r = 0; r < n_ranges; r++) {
      const s3s_fetch_range& k = R[r];
      h_tot[r] = 0;
      const int64_t nf = h_nf[r], f0 = first_frame[(size_t)r];
      if (nf == 0) continue;
      uint8_t* ws = dev<uint8_t>(ctx, B_PART_NFRAMES) + ws_off[(size_t)r];
      Frame* d_fr = dev<Frame>(ctx, B_FRAMES) + f0;
      uint32_t* d_orig = dev<uint32_t>(ctx, B_ITEM_SIZE) + f0 + r;
      int64_t* d_out_rel = dev<int64_t>(ctx, B_FRAME_OUT) + f0 + r;
      const size_t pp = first_part[(size_t)r] + (size_t)r;
      int64_t* d_base = reinterpret_cast<int64_t*>(ws + al(4 * ((size_t)k.num_partitions + 1)));
      launch_snappy_emit_frames(k.d_comp, dev<int64_t>(ctx, B_OFFSETS) + pp, k.num_partitions, d_base, d_fr, d_orig, d_status + r, ctx->stream,
                                codec == S3S_CODEC_LZF ? kChunkLzf : kChunkSnappy);
      launch_scan_u32(d_orig, nf, d_out_rel, ctx->stream);
      HIP_TRY(ctx, hipMemcpyAsync(&h_tot[r], d_out_rel + nf, 8, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(h_st, d_status, 4 * (size_t)n_ranges, hipMemcpyDeviceToHost, ctx->stream));
    record(ctx, 2);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int32_t r = 0; r < n_ranges; r++) {
      s3s_fetch_range& k = R[r];
      const int64_t nf = h_nf[r], f0 = first_frame[(size_t)r];
      if (k.status == S3S_OK && h_st[r] != 0) k.status = S3S_E_BAD_FRAME;
      if (k.status == S3S_OK) {
        k.out_len = h_tot[r];
        if (h_tot[r] > k.dst_capacity) k.status = S3S_E_CAPACITY;
      }
      if (nf == 0) continue;
      hipLaunchKernelGGL(rebase_frames_kernel, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, ctx->stream,
                         dev<Frame>(ctx, B_FRAMES) + f0, dev<int64_t>(ctx, B_FRAME_OUT) + f0 + r, dev<int64_t>(ctx, B_ITEM_OFF) + f0,
                         (int32_t)nf, (int64_t)reinterpret_cast<uintptr_t>(k.d_comp), (int64_t)reinterpret_cast<uintptr_t>(k.d_dst),
                         k.status != S3S_OK ? 1 : 0);
    }
    if (total_frames > 0)
      launch_snappy_decompress(nullptr, dev<Frame>(ctx, B_FRAMES), (int32_t)total_frames, dev<int64_t>(ctx, B_ITEM_OFF), nullptr,
                               d_dec_status, ctx->lz4_decode_variant, ctx->stream, codec == S3S_CODEC_LZF ? kChunkLzf : kChunkSnappy);
    HIP_TRY(ctx, hipGetLastError());
    record(ctx, 3);
    HIP_TRY(ctx, hipMemcpyAsync(&h_st[n_ranges], d_dec_status, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  }
  if (ctx->profile) {
    float ms = 0;
    hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[3]); ctx->stage_ms[S3S_STAGE_TOTAL] = ms;
    hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]); ctx->stage_ms[S3S_STAGE_CHECKSUM] = ms;
    hipEventElapsedTime(&ms, ctx->ev[1], ctx->ev[2]); ctx->stage_ms[S3S_STAGE_DISCOVER] = ms;
    hipEventElapsedTime(&ms, ctx->ev[2], ctx->ev[3]); ctx->stage_ms[S3S_STAGE_CODEC] = ms;
    if (lz4_split) {  // codec = the decode kernel alone, hash = the frame checks
      hipEventElapsedTime(&ms, ctx->ev[2], ctx->ev_hash); ctx->stage_ms[S3S_STAGE_CODEC] = ms;
      hipEventElapsedTime(&ms, ctx->ev_hash, ctx->ev[3]); ctx->stage_ms[S3S_STAGE_HASH] = ms;
    }
  }
  if (h_st[n_ranges] != 0) {
    // some frame of the batch is corrupt (or unsupported): decode the ranges one by one to say which
    for (int32_t r = 0; r < n_ranges; r++)
      if (R[r].status == S3S_OK) single(R[r]);
  }
  return verdict.finish(first_error());
}
