// codec_api.hip — the C-ABI of include/s3shuffle_codec.h: context, planning, orchestration.
//
// One s3s_ctx = one HIP stream + one growable device workspace + one pinned staging area, used
// by exactly one task thread (the reference creates one writer object per map task used by one
// task thread, S3ShuffleDataIO.scala:34-43).  Every call enqueues its kernels on the context's
// stream and synchronises once at the end to hand index / checksums back to the caller.
//
// There is NO CPU fallback anywhere in this file: without a HIP device s3s_create() fails and
// every entry point needs a context.
#include "s3s_ctx.h"
#include <atomic>

using namespace s3s;

namespace {
// S3S_DEBUG_SYNC=1: wait after every stage of the batched call and say which one finished (fault triage)
inline void dbg_sync(s3s_ctx* ctx, const char* what) {
  static const bool on = getenv("S3S_DEBUG_SYNC") != nullptr;
  if (!on) return;
  const hipError_t e = hipStreamSynchronize(ctx->stream);
  fprintf(stderr, "[s3s] %s: %s\n", what, hipGetErrorString(e));
}
}  // namespace

extern "C" {

const char* s3s_version(void) { return "s3shuffle-codec-mi355x 0.1.0 (gfx950)"; }
int s3s_abi_version(void) { return S3S_ABI_VERSION; }

int s3s_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

s3s_ctx* s3s_create(int device_ordinal, int64_t scratch_bytes) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    fail(nullptr, S3S_E_HIP, "no HIP device available (%s); this library has no CPU fallback",
         e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    return nullptr;
  }
  if (device_ordinal < 0 || device_ordinal >= n) {
    fail(nullptr, S3S_E_INVALID, "device ordinal %d out of range [0,%d)", device_ordinal, n);
    return nullptr;
  }
  s3s_ctx* ctx = new s3s_ctx();
  ctx->device = device_ordinal;
  auto bail = [&](const char* what, hipError_t err) -> s3s_ctx* {
    fail(nullptr, S3S_E_HIP, "%s failed: %s", what, hipGetErrorString(err));
    delete ctx;
    return nullptr;
  };
  if ((e = hipSetDevice(device_ordinal)) != hipSuccess) return bail("hipSetDevice", e);
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_ordinal) == hipSuccess && cus > 0) ctx->cu_count = cus;
  }
  if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess)
    return bail("hipStreamCreate", e);
  for (auto& ev : ctx->ev)
    if ((e = hipEventCreate(&ev)) != hipSuccess) return bail("hipEventCreate", e);
  if ((e = hipEventCreate(&ctx->ev_hash)) != hipSuccess) return bail("hipEventCreate", e);
  for (auto& ev : ctx->ev_auto)
    if ((e = hipEventCreate(&ev)) != hipSuccess) return bail("hipEventCreate", e);
  // constant tables for the checksum kernels
  const size_t tb = checksum_tables_bytes();
  std::vector<uint8_t> host_tabs(tb);
  checksum_tables_build(host_tabs.data());
  if (ensure(ctx, B_TABLES, tb) != S3S_OK ||
      hipMemcpy(ctx->buf[B_TABLES].p, host_tabs.data(), tb, hipMemcpyHostToDevice) != hipSuccess) {
    fail(nullptr, S3S_E_HIP, "table upload failed: %s", ctx->err);
    s3s_destroy(ctx);
    return nullptr;
  }
  if (scratch_bytes > 0) (void)ensure(ctx, B_SLOTS, (size_t)scratch_bytes);
  return ctx;
}

void s3s_destroy(s3s_ctx* ctx) {
  if (!ctx) return;
  hipSetDevice(ctx->device);
  if (ctx->stream) hipStreamSynchronize(ctx->stream);
  // the copy lanes may be the device's SHARED ones (host_batch.hip's arbiter): wait for what THIS context put on them (its
  // group events), not for other contexts' DMA — reaping a dead task thread's context must not stall behind live ones (advisor r4)
  if (ctx->hb_shared) {
    for (int i = 0; i < 2; i++) {
      if (ctx->hb_ev_in[i]) hipEventSynchronize(ctx->hb_ev_in[i]);
      if (ctx->hb_ev_out[i]) hipEventSynchronize(ctx->hb_ev_out[i]);
    }
  } else {
    if (ctx->hb_in) hipStreamSynchronize(ctx->hb_in);
    if (ctx->hb_out) hipStreamSynchronize(ctx->hb_out);
  }
  // (hipFree itself still synchronises the device — the buffers are hipMalloc'd, not pool allocations, so there is no
  // stream-ordered free for them; the per-event waits above only keep this thread from blocking on the shared lanes BEFORE it)
  for (auto& b : ctx->buf)
    if (b.p) hipFree(b.p);
  if (ctx->h_stage) hipHostFree(ctx->h_stage);
  for (auto& ev : ctx->ev)
    if (ev) hipEventDestroy(ev);
  if (ctx->ev_hash) hipEventDestroy(ctx->ev_hash);
  for (auto& ev : ctx->ev_auto)
    if (ev) hipEventDestroy(ev);
  for (auto& ev : ctx->ev_up)
    if (ev) hipEventDestroy(ev);
  if (ctx->copy_stream) hipStreamDestroy(ctx->copy_stream);
  for (int i = 0; i < 2; i++) {
    if (ctx->hb_ev_in[i]) hipEventDestroy(ctx->hb_ev_in[i]);
    if (ctx->hb_ev_out[i]) hipEventDestroy(ctx->hb_ev_out[i]);
  }
  if (ctx->hb_in && !ctx->hb_shared) hipStreamDestroy(ctx->hb_in);
  if (ctx->hb_out && !ctx->hb_shared) hipStreamDestroy(ctx->hb_out);
  if (ctx->stream) hipStreamDestroy(ctx->stream);
  delete ctx;
}

void* s3s_host_alloc(int64_t bytes) {
  if (bytes <= 0) return nullptr;
  void* p = nullptr;
  if (hipHostMalloc(&p, (size_t)bytes, hipHostMallocPortable) != hipSuccess) return nullptr;
  return p;
}

void s3s_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}

const char* s3s_last_error(const s3s_ctx* ctx) { return ctx ? ctx->err : g_create_error; }

int s3s_set_option(s3s_ctx* ctx, int key, int64_t value) {
  if (!ctx) return S3S_E_INVALID;
  switch (key) {
    case S3S_OPT_LZ4_BLOCK_SIZE:
      // LZ4BlockOutputStream accepts 64 .. 32 MiB; the map side takes what liblz4 parses with its 16-bit table (kLz4MaxBlock);
      // the reduce side decodes frames of any block size whatever this option says
      if (value < 64) return fail(ctx, S3S_E_INVALID, "lz4 blockSize must be >= 64, got %lld", (long long)value);
      if (value > kLz4MaxBlock)
        return fail(ctx, S3S_E_UNSUPPORTED, "lz4 blockSize %lld > %d not supported", (long long)value, kLz4MaxBlock);
      ctx->lz4_block = value;
      return S3S_OK;
    case S3S_OPT_SNAPPY_BLOCK_SIZE:
      if (value <= 0) return fail(ctx, S3S_E_INVALID, "snappy blockSize must be > 0");
      if (value > kMaxBlock)
        return fail(ctx, S3S_E_UNSUPPORTED, "snappy blockSize %lld > %d not supported", (long long)value, kMaxBlock);
      ctx->snappy_block = value;
      return S3S_OK;
    case S3S_OPT_PROFILE:
      ctx->profile = value != 0;
      return S3S_OK;
    case S3S_OPT_LZ4_DECODE_VARIANT:
      if (value != 3 && value != 4) return fail(ctx, S3S_E_INVALID, "decode variant must be 3 (ring decoder) or 4 (batch decoder)");
      ctx->lz4_decode_variant = (int)value;
      return S3S_OK;
    case S3S_OPT_SNAPPY_VARIANT:
      if (value < 0 || value > 1) return fail(ctx, S3S_E_INVALID, "snappy variant must be 0..1");
      ctx->snappy_variant = (int)value;
      return S3S_OK;
    case S3S_OPT_LZ4_VARIANT:
      if (value != 1 && value != 9 && value != 10) return fail(ctx, S3S_E_INVALID, "lz4 variant must be 1 (general batch), 10 (exact windows) or 9 (auto)");
      ctx->lz4_variant = (int)value;
      ctx->auto_samples[0] = ctx->auto_samples[1] = 0;
      ctx->auto_tick = 0;
      return S3S_OK;
  }
  return fail(ctx, S3S_E_INVALID, "unknown option %d", key);
}

int64_t s3s_get_option(const s3s_ctx* ctx, int key) {
  if (!ctx) return S3S_E_INVALID;
  switch (key) {
    case S3S_OPT_LZ4_BLOCK_SIZE: return ctx->lz4_block;
    case S3S_OPT_SNAPPY_BLOCK_SIZE: return ctx->snappy_block;
    case S3S_OPT_PROFILE: return ctx->profile;
    case S3S_OPT_LZ4_VARIANT: return ctx->lz4_variant;
    case S3S_OPT_LZ4_VARIANT_USED: return ctx->lz4_variant_used;
    case S3S_OPT_LZ4_DECODE_VARIANT: return ctx->lz4_decode_variant;
    case S3S_OPT_SNAPPY_VARIANT: return ctx->snappy_variant;
  }
  return S3S_E_INVALID;
}

void* s3s_stream(const s3s_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

double s3s_stage_ms(const s3s_ctx* ctx, int stage) {
  if (!ctx || stage < 0 || stage >= S3S_STAGE_COUNT) return -1.0;
  return ctx->stage_ms[stage];
}

int64_t s3s_max_compressed_size(const s3s_ctx* ctx, int codec, const int64_t* src_offsets,
                                int32_t n) {
  if (!src_offsets || n < 0) return S3S_E_INVALID;
  if (codec != S3S_CODEC_NONE && codec != S3S_CODEC_LZ4 && codec != S3S_CODEC_SNAPPY) return S3S_E_INVALID;
  const int64_t bs = effective_block(ctx, codec);
  int64_t total = 0;
  for (int32_t p = 0; p < n; p++) {
    const int64_t u = src_offsets[p + 1] - src_offsets[p];
    if (u < 0) return S3S_E_INVALID;
    total += max_partition_size(codec, bs, u);
  }
  return total;
}

// The map-side path.  Partition p is made of the segments [pfs[p], pfs[p+1]) of seg_offsets (ns segments in
// all, contiguous in d_src); every non-empty segment becomes one complete codec stream.
// Map-side calls in flight on a device (every context is one stream; the entry points return after their stream has
// drained, so this is what that GPU sees).  A call that is alone asks for the whole chip (10 wavefronts per CU is what the
// LZ4 tables allow); calls that run side by side ask for half of it each, so that two of them are resident together and
// the tail of one overlaps the body of the next (measured with four task threads: 89.4 against 87.0 GB/s; halving the
// live blocks also lifts the L2 hit rate of the candidate gathers from 84.5 to 90 %).
static std::atomic<int> g_compress_calls[64];
struct CompressCallScope {
  std::atomic<int>& n;
  explicit CompressCallScope(const s3s_ctx* ctx) : n(g_compress_calls[ctx->device & 63]) { n.fetch_add(1, std::memory_order_relaxed); }
  ~CompressCallScope() { n.fetch_sub(1, std::memory_order_relaxed); }
};
static int lz4_resident_waves(const s3s_ctx* ctx) {
  return (g_compress_calls[ctx->device & 63].load(std::memory_order_relaxed) > 1 ? 5 : 10) * ctx->cu_count;
}

static int compress_core(s3s_ctx* ctx, int codec, int checksum_algo, const uint8_t* d_src,
                         const int64_t* seg_offsets, int32_t ns, const int32_t* pfs, int32_t n,
                         uint8_t* d_dst, int64_t dst_capacity, int64_t* out_index,
                         int64_t* out_checksums, int64_t* out_total) {
  if (!ctx) return S3S_E_INVALID;
  ctx->err[0] = 0;
  const CompressCallScope in_flight(ctx);
  if (n < 0 || ns < 0 || !seg_offsets || !pfs || !out_index)
    return fail(ctx, S3S_E_INVALID, "null offsets/index or negative partition count");
  if (pfs[0] != 0 || pfs[n] != ns) return fail(ctx, S3S_E_INVALID, "part_first_seg must start at 0 and end at n_segs");
  for (int32_t p = 0; p < n; p++)
    if (pfs[p + 1] < pfs[p]) return fail(ctx, S3S_E_INVALID, "part_first_seg not monotonic at %d", p);
  if (codec == S3S_CODEC_ZSTD || codec == S3S_CODEC_LZF)
    return fail(ctx, S3S_E_UNSUPPORTED, "%s compression stays on the JVM codec (decode only: s3s_decompress_range*)", codec == S3S_CODEC_LZF ? "lzf" : "zstd");
  if (codec != S3S_CODEC_NONE && codec != S3S_CODEC_LZ4 && codec != S3S_CODEC_SNAPPY)
    return fail(ctx, S3S_E_INVALID, "unknown codec %d", codec);
  if (checksum_algo != S3S_CHECKSUM_NONE && checksum_algo != S3S_CHECKSUM_ADLER32 && checksum_algo != S3S_CHECKSUM_CRC32 &&
      checksum_algo != S3S_CHECKSUM_CRC32C)
    return fail(ctx, S3S_E_INVALID, "unknown checksum algorithm %d", checksum_algo);
  if (checksum_algo != S3S_CHECKSUM_NONE && !out_checksums)
    return fail(ctx, S3S_E_INVALID, "out_checksums is null but a checksum algorithm is selected");
  for (int32_t g = 0; g < ns; g++)
    if (seg_offsets[g + 1] < seg_offsets[g]) return fail(ctx, S3S_E_INVALID, "offsets not monotonic at %d", g);
  if (dst_capacity < 0) return fail(ctx, S3S_E_INVALID, "negative dst_capacity");
  if (codec == S3S_CODEC_SNAPPY && !snappy_compress_available())
    return fail(ctx, S3S_E_UNSUPPORTED, "snappy compression is not available in this build");
  HIP_TRY(ctx, hipSetDevice(ctx->device));

  const int64_t bs = effective_block(ctx, codec);
  const int64_t total_u = ns > 0 ? seg_offsets[ns] - seg_offsets[0] : 0;
  if ((total_u > 0 && !d_src) || (!d_dst && dst_capacity > 0)) return fail(ctx, S3S_E_INVALID, "null data pointer");
  const int level = codec == S3S_CODEC_LZ4 ? lz4_level(bs) : 0;

  // ---- plan (host): items in .data order, first item per partition, checksum segment slots ---
  int64_t n_items64 = 0, n_chunks64 = 0;
  if (codec != S3S_CODEC_NONE) {
    for (int32_t g = 0; g < ns; g++) {
      const int64_t u = seg_offsets[g + 1] - seg_offsets[g];
      if (u > 0) {
        const int64_t ch = (u + bs - 1) / bs;
        n_chunks64 += ch;
        n_items64 += ch + 1;  // + LZ4 end frame / snappy stream header
      }
    }
  }
  if (n_items64 > 0x7fffff00ll) return fail(ctx, S3S_E_UNSUPPORTED, "too many codec blocks in one call");
  const int32_t n_items = (int32_t)n_items64, n_chunks = (int32_t)n_chunks64;

  const size_t items_bytes = sizeof(Item) * (size_t)n_items;
  const size_t pf_bytes = sizeof(int32_t) * (size_t)(n + 1);
  const size_t seg_bytes = sizeof(int32_t) * (size_t)(n + 1);
  const size_t idx_bytes = sizeof(int64_t) * (size_t)(n + 1);
  const size_t sums_bytes = sizeof(int64_t) * (size_t)(n > 0 ? n : 1);
  // pinned staging layout: [items][part_first][seg_start][index out][sums out][status out]
  size_t o_items = 0, o_pf = (o_items + items_bytes + 15) & ~size_t(15),
         o_seg = (o_pf + pf_bytes + 15) & ~size_t(15), o_idx = (o_seg + seg_bytes + 15) & ~size_t(15),
         o_sums = (o_idx + idx_bytes + 15) & ~size_t(15), o_status = (o_sums + sums_bytes + 15) & ~size_t(15),
         stage_total = o_status + 16;
  int rc;
  if ((rc = ensure_stage(ctx, stage_total))) return rc;
  uint8_t* hs = static_cast<uint8_t*>(ctx->h_stage);
  Item* h_items = reinterpret_cast<Item*>(hs + o_items);
  int32_t* h_pf = reinterpret_cast<int32_t*>(hs + o_pf);
  int32_t* h_seg = reinterpret_cast<int32_t*>(hs + o_seg);
  int64_t* h_idx = reinterpret_cast<int64_t*>(hs + o_idx);
  int64_t* h_sums = reinterpret_cast<int64_t*>(hs + o_sums);
  int32_t* h_status = reinterpret_cast<int32_t*>(hs + o_status);

  {
    int32_t it = 0, ch = 0, seg = 0;
    for (int32_t p = 0; p < n; p++) {
      h_pf[p] = it;
      h_seg[p] = seg;
      int64_t worst = 0;  // of the partition's compressed bytes: one stream per non-empty segment
      for (int32_t g = pfs[p]; g < pfs[p + 1]; g++) {
        const int64_t u = seg_offsets[g + 1] - seg_offsets[g];
        worst += max_partition_size(codec, bs, u);
        if (codec == S3S_CODEC_NONE || u <= 0) continue;
        if (codec == S3S_CODEC_SNAPPY) h_items[it++] = Item{0, 0, kItemSnappyHeader, -1, p};
        for (int64_t pos = 0; pos < u; pos += bs) {
          const int32_t len = (int32_t)((u - pos) < bs ? (u - pos) : bs);
          const int32_t kind = codec == S3S_CODEC_LZ4 ? (kItemLz4Chunk | (level << 8)) : kItemSnappyChunk;
          h_items[it++] = Item{seg_offsets[g] + pos, len, kind, ch++, p};
        }
        if (codec == S3S_CODEC_LZ4) h_items[it++] = Item{0, 0, kItemLz4End | (level << 8), -1, p};
      }
      seg += worst_segs(worst);
    }
    h_pf[n] = it;
    h_seg[n] = seg;
  }

  if ((rc = ensure(ctx, B_INDEX, idx_bytes))) return rc;
  if ((rc = ensure(ctx, B_SUMS, sums_bytes))) return rc;
  if ((rc = ensure(ctx, B_STATUS, 16))) return rc;
  HIP_TRY(ctx, hipMemsetAsync(ctx->buf[B_STATUS].p, 0, 16, ctx->stream));
  record(ctx, 0);
  bool auto_timed = false;
  int auto_which = 0, lz4_variant_run = 10;
  constexpr int64_t kAutoMinBytes = 4 << 20;

  if (codec == S3S_CODEC_NONE) {
    // spark.shuffle.compress=false: the partition bytes are the stream
    if (total_u > dst_capacity) return fail(ctx, S3S_E_CAPACITY, "dst_capacity %lld < %lld", (long long)dst_capacity, (long long)total_u);
    if (total_u > 0)
      HIP_TRY(ctx, hipMemcpyAsync(d_dst, d_src + seg_offsets[0], (size_t)total_u, hipMemcpyDeviceToDevice, ctx->stream));
    for (int32_t p = 0; p <= n; p++) h_idx[p] = ns > 0 ? seg_offsets[pfs[p]] - seg_offsets[0] : 0;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->buf[B_INDEX].p, h_idx, idx_bytes, hipMemcpyHostToDevice, ctx->stream));
    record(ctx, 1);
    record(ctx, 2);
  } else {
    if ((rc = ensure(ctx, B_ITEMS, items_bytes + 16))) return rc;
    if ((rc = ensure(ctx, B_PART_FIRST, pf_bytes))) return rc;
    // a slot holds one chunk's codec output: LZ4 payloads never exceed the chunk (RAW fallback), a raw
    // snappy block can grow to MaxCompressedLength(chunk)
    const int64_t slot_stride = codec == S3S_CODEC_SNAPPY
                                    ? (int64_t)kSlotHeader + ((snappy_max_len(bs) + 15) & ~int64_t(15))
                                    : (int64_t)kSlotHeader + ((bs + 15) & ~int64_t(15));
    if ((rc = ensure(ctx, B_SLOTS, (size_t)slot_stride * (size_t)(n_chunks > 0 ? n_chunks : 1)))) return rc;
    if ((rc = ensure(ctx, B_ITEM_SIZE, sizeof(uint32_t) * (size_t)(n_items + 1)))) return rc;
    if ((rc = ensure(ctx, B_ITEM_OFF, sizeof(int64_t) * (size_t)(n_items + 1)))) return rc;
    if ((rc = ensure(ctx, B_ITEM_CHECK, sizeof(uint32_t) * (size_t)(n_items + 1)))) return rc;
    if ((rc = ensure(ctx, B_WORK, 64))) return rc;
    if (n_items > 0)
      HIP_TRY(ctx, hipMemcpyAsync(ctx->buf[B_ITEMS].p, h_items, items_bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->buf[B_PART_FIRST].p, h_pf, pf_bytes, hipMemcpyHostToDevice, ctx->stream));
    // the snappy stream header has a constant size; seed item_size for every item kind the
    // codec kernel does not write
    if (codec == S3S_CODEC_LZ4) {
      int variant = ctx->lz4_variant;
      if (variant == 9) {  // auto: run the parse that has been faster on this context's data
        auto_which = ctx->auto_choice;
        if (total_u >= kAutoMinBytes) {  // large enough for the kernel time to mean something
          auto_timed = true;
          if (ctx->auto_samples[0] < 2 || ctx->auto_samples[1] < 2)
            auto_which = ctx->auto_samples[0] <= ctx->auto_samples[1] ? 0 : 1;  // 1, 10, 1, 10
          else if (++ctx->auto_tick % 32 == 0)
            auto_which = 1 - ctx->auto_choice;  // keep the other one's figure fresh
        }
        variant = auto_which ? 10 : 1;
        if (auto_timed) HIP_TRY(ctx, hipEventRecord(ctx->ev_auto[0], ctx->stream));
      }
      ctx->lz4_variant_used = variant;
      lz4_variant_run = variant;
    }
    // One launch over all blocks — or, when the source is still on the host (up_host), one launch per uploaded
    // chunk: the copy engine brings chunk g+1 while the blocks of chunk g are compressed.
    auto launch_codec = [&](int32_t i0, int32_t i1) {
      if (i1 <= i0) return;
      if (codec == S3S_CODEC_LZ4)
        launch_lz4_compress(d_src, dev<Item>(ctx, B_ITEMS) + i0, i1 - i0, dev<uint32_t>(ctx, B_ITEM_CHECK) + i0,
                            dev<uint8_t>(ctx, B_SLOTS), (int32_t)slot_stride, dev<uint32_t>(ctx, B_ITEM_SIZE) + i0, dev<uint32_t>(ctx, B_WORK),
                            lz4_resident_waves(ctx), lz4_variant_run, ctx->stream,
                            ctx->profile && i1 == n_items ? ctx->ev_hash : nullptr);
      else
        launch_snappy_compress(d_src, dev<Item>(ctx, B_ITEMS) + i0, i1 - i0, dev<uint8_t>(ctx, B_SLOTS), slot_stride,
                               dev<uint32_t>(ctx, B_ITEM_SIZE) + i0, ctx->snappy_variant, ctx->stream);
    };
    constexpr int64_t kUpChunk = 64ll << 20;  // one launch should fill the chip (2560 resident blocks = 80 MiB): 16 MiB chunks were slower than no overlap
    if (ctx->up_host && total_u > kUpChunk + (kUpChunk >> 1)) {
      const int64_t base0 = seg_offsets[0];
      const int32_t n_up = (int32_t)((total_u + kUpChunk - 1) / kUpChunk);
      if (!ctx->copy_stream) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
      while ((int32_t)ctx->ev_up.size() < n_up + 1) {
        hipEvent_t ev = nullptr;
        HIP_TRY(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        ctx->ev_up.push_back(ev);
      }
      // the copy stream must not overtake work of this stream that still reads the buffer (previous call): order it
      HIP_TRY(ctx, hipEventRecord(ctx->ev_up[(size_t)n_up], ctx->stream));
      HIP_TRY(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->ev_up[(size_t)n_up], 0));
      int32_t i0 = 0;
      for (int32_t g = 0; g < n_up; g++) {
        const int64_t b0 = (int64_t)g * kUpChunk, b1 = b0 + kUpChunk < total_u ? b0 + kUpChunk : total_u;
        HIP_TRY(ctx, hipMemcpyAsync(const_cast<uint8_t*>(d_src) + base0 + b0, ctx->up_host + b0, (size_t)(b1 - b0),
                                    hipMemcpyHostToDevice, ctx->copy_stream));
        HIP_TRY(ctx, hipEventRecord(ctx->ev_up[(size_t)g], ctx->copy_stream));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_up[(size_t)g], 0));
        int32_t i1 = i0;  // items are in ascending source order: take those whose bytes have all arrived
        while (i1 < n_items && (h_items[i1].len == 0 || h_items[i1].src_off + h_items[i1].len <= base0 + b1)) i1++;
        if (g == n_up - 1) i1 = n_items;
        launch_codec(i0, i1);
        i0 = i1;
      }
    } else {
      if (ctx->up_host && total_u > 0)
        HIP_TRY(ctx, hipMemcpyAsync(const_cast<uint8_t*>(d_src) + seg_offsets[0], ctx->up_host, (size_t)total_u,
                                    hipMemcpyHostToDevice, ctx->stream));
      launch_codec(0, n_items);
    }
    if (codec == S3S_CODEC_LZ4 && auto_timed) HIP_TRY(ctx, hipEventRecord(ctx->ev_auto[1], ctx->stream));
    HIP_TRY(ctx, hipGetLastError());
    record(ctx, 1);
    launch_scan_items(dev<Item>(ctx, B_ITEMS), dev<uint32_t>(ctx, B_ITEM_SIZE), n_items,
                      dev<int64_t>(ctx, B_ITEM_OFF), dev<int32_t>(ctx, B_PART_FIRST), n,
                      dev<int64_t>(ctx, B_INDEX), ctx->stream);
    launch_gather_items(d_src, dev<Item>(ctx, B_ITEMS), n_items, dev<uint8_t>(ctx, B_SLOTS), slot_stride,
                        dev<uint32_t>(ctx, B_ITEM_SIZE), dev<int64_t>(ctx, B_ITEM_OFF), d_dst,
                        dst_capacity, dev<int32_t>(ctx, B_STATUS), ctx->stream);
    HIP_TRY(ctx, hipGetLastError());
    record(ctx, 2);
  }
  if (checksum_algo != S3S_CHECKSUM_NONE && n > 0) {
    if ((rc = run_checksum(ctx, checksum_algo, d_dst, dev<int64_t>(ctx, B_INDEX), n, h_seg,
                           dev<int64_t>(ctx, B_SUMS), dst_capacity)))
      return rc;
  }
  record(ctx, 3);
  HIP_TRY(ctx, hipMemcpyAsync(h_idx, ctx->buf[B_INDEX].p, idx_bytes, hipMemcpyDeviceToHost, ctx->stream));
  if (checksum_algo != S3S_CHECKSUM_NONE && n > 0)
    HIP_TRY(ctx, hipMemcpyAsync(h_sums, ctx->buf[B_SUMS].p, sizeof(int64_t) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(h_status, ctx->buf[B_STATUS].p, 4, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (auto_timed) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, ctx->ev_auto[0], ctx->ev_auto[1]) == hipSuccess && ms > 0) {
      const double rate = (double)ms / ((double)total_u / 1048576.0);
      double& r = ctx->auto_ms_per_mib[auto_which];
      // first samples: keep the best (the very first call runs on cold clocks); later: smooth
      r = ctx->auto_samples[auto_which] == 0 ? rate
          : (ctx->auto_samples[auto_which] < 2 ? (rate < r ? rate : r) : 0.5 * r + 0.5 * rate);
      ctx->auto_samples[auto_which]++;
      if (ctx->auto_samples[0] >= 2 && ctx->auto_samples[1] >= 2)
        ctx->auto_choice = ctx->auto_ms_per_mib[1] < ctx->auto_ms_per_mib[0] ? 1 : 0;
    }
  }
  if (ctx->profile) {
    float ms = 0;
    hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[3]); ctx->stage_ms[S3S_STAGE_TOTAL] = ms;
    hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]); ctx->stage_ms[S3S_STAGE_CODEC] = ms;
    ctx->stage_ms[S3S_STAGE_HASH] = 0;
    if (codec == S3S_CODEC_LZ4) {  // split the xxHash32 pre-pass off the compress kernel
      hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev_hash); ctx->stage_ms[S3S_STAGE_HASH] = ms;
      hipEventElapsedTime(&ms, ctx->ev_hash, ctx->ev[1]); ctx->stage_ms[S3S_STAGE_CODEC] = ms;
    }
    hipEventElapsedTime(&ms, ctx->ev[1], ctx->ev[2]); ctx->stage_ms[S3S_STAGE_ASSEMBLE] = ms;
    hipEventElapsedTime(&ms, ctx->ev[2], ctx->ev[3]); ctx->stage_ms[S3S_STAGE_CHECKSUM] = ms;
    ctx->stage_ms[S3S_STAGE_DISCOVER] = 0;
  }
  memcpy(out_index, h_idx, idx_bytes);
  if (out_total) *out_total = h_idx[n];
  if (*h_status != 0 || h_idx[n] > dst_capacity)
    return fail(ctx, S3S_E_CAPACITY, "dst_capacity %lld too small for %lld output bytes",
                (long long)dst_capacity, (long long)h_idx[n]);
  if (checksum_algo != S3S_CHECKSUM_NONE && n > 0) memcpy(out_checksums, h_sums, sizeof(int64_t) * (size_t)n);
  return S3S_OK;
}

int s3s_compress_map_output_device(s3s_ctx* ctx, int codec, int checksum_algo,
                                   const uint8_t* d_src, const int64_t* src_offsets, int32_t n,
                                   uint8_t* d_dst, int64_t dst_capacity, int64_t* out_index,
                                   int64_t* out_checksums, int64_t* out_total) {
  if (!ctx) return S3S_E_INVALID;
  if (n < 0 || !src_offsets) {
    ctx->err[0] = 0;
    return fail(ctx, S3S_E_INVALID, "null offsets/index or negative partition count");
  }
  std::vector<int32_t> pfs((size_t)n + 1);  // one segment per partition
  for (int32_t p = 0; p <= n; p++) pfs[(size_t)p] = p;
  return compress_core(ctx, codec, checksum_algo, d_src, src_offsets, n, pfs.data(), n, d_dst, dst_capacity,
                       out_index, out_checksums, out_total);
}

// Batched map side: ONE codec launch over the chunks of every map task (a 128 MiB task alone is 1.6
// rounds of the chip's 2 560 resident wavefronts; an 8 MiB one a tenth of a round), then scan / gather /
// checksum per task on the same stream, ONE stream synchronisation for the whole batch.
int s3s_compress_map_outputs_batch_device(s3s_ctx* ctx, int codec, int checksum_algo, s3s_map_task* tasks,
                                          int32_t n_tasks) {
  if (!ctx) return S3S_E_INVALID;
  ctx->err[0] = 0;
  const CompressCallScope in_flight(ctx);
  if (n_tasks < 0 || (n_tasks > 0 && !tasks)) return fail(ctx, S3S_E_INVALID, "null task array or negative count");
  // "stamped on entry" (s3shuffle_codec.h): BEFORE the argument checks below, so a caller with zeroed status fields never reads
  // S3S_OK out of a call that was refused (advisor r4)
  BatchVerdict<s3s_map_task> verdict(tasks, n_tasks);
  if (codec == S3S_CODEC_ZSTD || codec == S3S_CODEC_LZF)
    return fail(ctx, S3S_E_UNSUPPORTED, "%s compression stays on the JVM codec (decode only: s3s_decompress_range*)", codec == S3S_CODEC_LZF ? "lzf" : "zstd");
  if (codec != S3S_CODEC_NONE && codec != S3S_CODEC_LZ4 && codec != S3S_CODEC_SNAPPY)
    return fail(ctx, S3S_E_INVALID, "unknown codec %d", codec);
  if (checksum_algo != S3S_CHECKSUM_NONE && checksum_algo != S3S_CHECKSUM_ADLER32 && checksum_algo != S3S_CHECKSUM_CRC32 &&
      checksum_algo != S3S_CHECKSUM_CRC32C)
    return fail(ctx, S3S_E_INVALID, "unknown checksum algorithm %d", checksum_algo);
  if (codec == S3S_CODEC_SNAPPY && !snappy_compress_available())
    return fail(ctx, S3S_E_UNSUPPORTED, "snappy compression is not available in this build");
  if (n_tasks == 0) return S3S_OK;
  if (codec == S3S_CODEC_NONE) {  // nothing to batch: plain copies
    int worst = S3S_OK;
    for (int32_t t = 0; t < n_tasks; t++) {
      s3s_map_task& k = tasks[t];
      k.status = s3s_compress_map_output_device(ctx, codec, checksum_algo, k.d_src, k.src_offsets, k.num_partitions,
                                                k.d_dst, k.dst_capacity, k.out_index, k.out_checksums, &k.out_total);
      if (k.status != S3S_OK && worst == S3S_OK) worst = k.status;
    }
    return verdict.finish(worst);
  }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int64_t bs = effective_block(ctx, codec);
  const int level = codec == S3S_CODEC_LZ4 ? lz4_level(bs) : 0;
  // ---- validate, count ------------------------------------------------------------------------------
  int64_t n_items64 = 0, n_chunks64 = 0, n_parts64 = 0;
  for (int32_t t = 0; t < n_tasks; t++) {
    const s3s_map_task& k = tasks[t];
    if (k.num_partitions < 0 || !k.src_offsets || !k.out_index || k.dst_capacity < 0)
      return fail(ctx, S3S_E_INVALID, "task %d: null offsets/index, negative partition count or capacity", t);
    if (checksum_algo != S3S_CHECKSUM_NONE && !k.out_checksums)
      return fail(ctx, S3S_E_INVALID, "task %d: out_checksums is null but a checksum algorithm is selected", t);
    for (int32_t p = 0; p < k.num_partitions; p++) {
      const int64_t u = k.src_offsets[p + 1] - k.src_offsets[p];
      if (u < 0) return fail(ctx, S3S_E_INVALID, "task %d: offsets not monotonic at %d", t, p);
      if (u > 0) {
        const int64_t ch = (u + bs - 1) / bs;
        n_chunks64 += ch;
        n_items64 += ch + 1;
      }
    }
    const int64_t tu = k.num_partitions > 0 ? k.src_offsets[k.num_partitions] - k.src_offsets[0] : 0;
    if ((tu > 0 && !k.d_src) || (!k.d_dst && k.dst_capacity > 0)) return fail(ctx, S3S_E_INVALID, "task %d: null data pointer", t);
    n_parts64 += k.num_partitions;
  }
  if (n_items64 > 0x7fffff00ll) return fail(ctx, S3S_E_UNSUPPORTED, "too many codec blocks in one call");
  const int32_t n_items = (int32_t)n_items64, n_chunks = (int32_t)n_chunks64;
  const size_t np1 = (size_t)n_parts64 + (size_t)n_tasks;  // sum of (n_t + 1)
  // pinned staging: [items][part_first (per task, relative)][seg_start (per task)][index out][sums out][status out]
  const size_t items_bytes = sizeof(Item) * (size_t)n_items;
  auto al = [](size_t x) { return (x + 15) & ~size_t(15); };
  const size_t o_items = 0, o_pf = al(o_items + items_bytes), o_seg = al(o_pf + 4 * np1), o_idx = al(o_seg + 4 * np1),
               o_sums = al(o_idx + 8 * np1), o_status = al(o_sums + 8 * ((size_t)n_parts64 + 1)),
               o_tails = al(o_status + 4 * (size_t)n_tasks + 16), stage_total = o_tails + sizeof(TaskTail) * (size_t)n_tasks + 16;
  int rc;
  if ((rc = ensure_stage(ctx, stage_total))) return rc;
  uint8_t* hs = static_cast<uint8_t*>(ctx->h_stage);
  Item* h_items = reinterpret_cast<Item*>(hs + o_items);
  int32_t* h_pf = reinterpret_cast<int32_t*>(hs + o_pf);
  int32_t* h_seg = reinterpret_cast<int32_t*>(hs + o_seg);
  int64_t* h_idx = reinterpret_cast<int64_t*>(hs + o_idx);
  int64_t* h_sums = reinterpret_cast<int64_t*>(hs + o_sums);
  int32_t* h_status = reinterpret_cast<int32_t*>(hs + o_status);
  TaskTail* h_tails = reinterpret_cast<TaskTail*>(hs + o_tails);
  std::vector<int32_t> first_item((size_t)n_tasks + 1), first_part((size_t)n_tasks + 1), first_seg((size_t)n_tasks + 1);
  const uint8_t* base = tasks[0].d_src;  // item sources are offsets from one base pointer (signed 64-bit)
  {
    int32_t it = 0, ch = 0, pp = 0, seg = 0;
    for (int32_t t = 0; t < n_tasks; t++) {
      const s3s_map_task& k = tasks[t];
      first_item[(size_t)t] = it;
      first_part[(size_t)t] = pp;
      first_seg[(size_t)t] = seg;
      const int64_t delta = k.d_src ? (int64_t)(k.d_src - base) : 0;
      int32_t* pf = h_pf + pp + t;
      int32_t* sg = h_seg + pp + t;
      int32_t seg_t = 0;
      for (int32_t p = 0; p < k.num_partitions; p++) {
        pf[p] = it - first_item[(size_t)t];
        sg[p] = seg_t;
        const int64_t u = k.src_offsets[p + 1] - k.src_offsets[p];
        seg_t += worst_segs(max_partition_size(codec, bs, u));
        if (u <= 0) continue;
        if (codec == S3S_CODEC_SNAPPY) h_items[it++] = Item{0, 0, kItemSnappyHeader, -1, p};
        for (int64_t pos = 0; pos < u; pos += bs) {
          const int32_t len = (int32_t)((u - pos) < bs ? (u - pos) : bs);
          const int32_t kind = codec == S3S_CODEC_LZ4 ? (kItemLz4Chunk | (level << 8)) : kItemSnappyChunk;
          h_items[it++] = Item{delta + k.src_offsets[p] + pos, len, kind, ch++, p};
        }
        if (codec == S3S_CODEC_LZ4) h_items[it++] = Item{0, 0, kItemLz4End | (level << 8), -1, p};
      }
      pf[k.num_partitions] = it - first_item[(size_t)t];
      sg[k.num_partitions] = seg_t;
      seg += seg_t;
      pp += k.num_partitions;
    }
    first_item[(size_t)n_tasks] = it;
    first_part[(size_t)n_tasks] = pp;
    first_seg[(size_t)n_tasks] = seg;
  }
  const int32_t total_segs = first_seg[(size_t)n_tasks];
  const int64_t slot_stride = codec == S3S_CODEC_SNAPPY
                                  ? (int64_t)kSlotHeader + ((snappy_max_len(bs) + 15) & ~int64_t(15))
                                  : (int64_t)kSlotHeader + ((bs + 15) & ~int64_t(15));
  if ((rc = ensure(ctx, B_ITEMS, items_bytes + 16))) return rc;
  if ((rc = ensure(ctx, B_PART_FIRST, 4 * np1))) return rc;
  if ((rc = ensure(ctx, B_SEG_START, 4 * np1))) return rc;
  if ((rc = ensure(ctx, B_INDEX, 8 * np1))) return rc;
  if ((rc = ensure(ctx, B_SUMS, 8 * ((size_t)n_parts64 + 1)))) return rc;
  if ((rc = ensure(ctx, B_STATUS, 4 * (size_t)n_tasks + 16))) return rc;
  if ((rc = ensure(ctx, B_PARTIAL, 16 * (size_t)(total_segs > 0 ? total_segs : 1)))) return rc;
  if ((rc = ensure(ctx, B_SLOTS, (size_t)slot_stride * (size_t)(n_chunks > 0 ? n_chunks : 1)))) return rc;
  if ((rc = ensure(ctx, B_ITEM_SIZE, sizeof(uint32_t) * (size_t)(n_items + 1)))) return rc;
  if ((rc = ensure(ctx, B_ITEM_OFF, sizeof(int64_t) * ((size_t)n_items + (size_t)n_tasks + 1)))) return rc;
  if ((rc = ensure(ctx, B_ITEM_CHECK, sizeof(uint32_t) * (size_t)(n_items + 1)))) return rc;
  if ((rc = ensure(ctx, B_WORK, 64))) return rc;
  if ((rc = ensure(ctx, B_TAILS, sizeof(TaskTail) * (size_t)n_tasks + 16))) return rc;
  for (int32_t t = 0; t < n_tasks; t++) {  // what the once-per-call tail kernels need to know about each task
    const s3s_map_task& k = tasks[t];
    TaskTail& d = h_tails[t];
    d.first_item = first_item[(size_t)t];
    d.n_items = first_item[(size_t)t + 1] - first_item[(size_t)t];
    d.first_pp = first_part[(size_t)t] + t;
    d.n_parts = k.num_partitions;
    d.first_part = first_part[(size_t)t];
    d.first_seg = first_seg[(size_t)t];
    d.n_segs = first_seg[(size_t)t + 1] - first_seg[(size_t)t];
    d.pad = 0;
    d.data = k.d_dst;
    d.data_len = k.dst_capacity;
    d.dst = k.d_dst;
    d.dst_capacity = k.dst_capacity;
  }
  HIP_TRY(ctx, hipMemsetAsync(ctx->buf[B_STATUS].p, 0, 4 * (size_t)n_tasks + 16, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(ctx->buf[B_TAILS].p, h_tails, sizeof(TaskTail) * (size_t)n_tasks, hipMemcpyHostToDevice, ctx->stream));
  // one upload: items | part_first | seg_start are consecutive in the staging buffer but live in separate
  // device buffers
  if (n_items > 0)
    HIP_TRY(ctx, hipMemcpyAsync(ctx->buf[B_ITEMS].p, h_items, items_bytes, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(ctx->buf[B_PART_FIRST].p, h_pf, 4 * np1, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(ctx->buf[B_SEG_START].p, h_seg, 4 * np1, hipMemcpyHostToDevice, ctx->stream));
  record(ctx, 0);
  // ---- ONE codec launch over every task's chunks ---------------------------------------------------------
  if (codec == S3S_CODEC_LZ4) {
    const int variant = ctx->lz4_variant == 9 ? 10 : ctx->lz4_variant;
    ctx->lz4_variant_used = variant;
    launch_lz4_compress(base, dev<Item>(ctx, B_ITEMS), n_items, dev<uint32_t>(ctx, B_ITEM_CHECK),
                        dev<uint8_t>(ctx, B_SLOTS), (int32_t)slot_stride, dev<uint32_t>(ctx, B_ITEM_SIZE), dev<uint32_t>(ctx, B_WORK), lz4_resident_waves(ctx),
                        variant, ctx->stream,
                        ctx->profile ? ctx->ev_hash : nullptr);
  } else {
    launch_snappy_compress(base, dev<Item>(ctx, B_ITEMS), n_items, dev<uint8_t>(ctx, B_SLOTS), slot_stride,
                           dev<uint32_t>(ctx, B_ITEM_SIZE), ctx->snappy_variant, ctx->stream);
  }
  HIP_TRY(ctx, hipGetLastError());
  dbg_sync(ctx, "batch codec");
  record(ctx, 1);
  // ---- offsets, .data images, checksums of every task: one launch each (TaskTail, s3s_internal.h) ---------------------
  launch_scan_items_batch(dev<TaskTail>(ctx, B_TAILS), n_tasks, dev<uint32_t>(ctx, B_ITEM_SIZE), dev<int64_t>(ctx, B_ITEM_OFF),
                          dev<int32_t>(ctx, B_PART_FIRST), dev<int64_t>(ctx, B_INDEX), ctx->stream);
  dbg_sync(ctx, "batch scan");
  launch_gather_items_batch(dev<TaskTail>(ctx, B_TAILS), n_tasks, n_items, base, dev<Item>(ctx, B_ITEMS), dev<uint8_t>(ctx, B_SLOTS),
                            slot_stride, dev<uint32_t>(ctx, B_ITEM_SIZE), dev<int64_t>(ctx, B_ITEM_OFF), dev<int32_t>(ctx, B_STATUS),
                            ctx->stream);
  dbg_sync(ctx, "batch gather");
  HIP_TRY(ctx, hipGetLastError());
  record(ctx, 2);
  if (checksum_algo != S3S_CHECKSUM_NONE) {
    launch_checksum_batch(checksum_algo, dev<TaskTail>(ctx, B_TAILS), n_tasks, total_segs, (int32_t)n_parts64, dev<int64_t>(ctx, B_INDEX),
                          dev<int32_t>(ctx, B_SEG_START), ctx->buf[B_TABLES].p, dev<uint32_t>(ctx, B_PARTIAL), dev<int64_t>(ctx, B_SUMS),
                          ctx->stream);
    HIP_TRY(ctx, hipGetLastError());
  }
  record(ctx, 3);
  HIP_TRY(ctx, hipMemcpyAsync(h_idx, ctx->buf[B_INDEX].p, 8 * np1, hipMemcpyDeviceToHost, ctx->stream));
  if (checksum_algo != S3S_CHECKSUM_NONE && n_parts64 > 0)
    HIP_TRY(ctx, hipMemcpyAsync(h_sums, ctx->buf[B_SUMS].p, 8 * (size_t)n_parts64, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(h_status, ctx->buf[B_STATUS].p, 4 * (size_t)n_tasks, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->profile) {
    float ms = 0;
    hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[3]); ctx->stage_ms[S3S_STAGE_TOTAL] = ms;
    hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]); ctx->stage_ms[S3S_STAGE_CODEC] = ms;
    ctx->stage_ms[S3S_STAGE_HASH] = 0;
    if (codec == S3S_CODEC_LZ4) {
      hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev_hash); ctx->stage_ms[S3S_STAGE_HASH] = ms;
      hipEventElapsedTime(&ms, ctx->ev_hash, ctx->ev[1]); ctx->stage_ms[S3S_STAGE_CODEC] = ms;
    }
    hipEventElapsedTime(&ms, ctx->ev[1], ctx->ev[2]); ctx->stage_ms[S3S_STAGE_ASSEMBLE] = ms;
    hipEventElapsedTime(&ms, ctx->ev[2], ctx->ev[3]); ctx->stage_ms[S3S_STAGE_CHECKSUM] = ms;
    ctx->stage_ms[S3S_STAGE_DISCOVER] = 0;
  }
  int worst = S3S_OK;
  for (int32_t t = 0; t < n_tasks; t++) {
    s3s_map_task& k = tasks[t];
    const int32_t pp = first_part[(size_t)t] + t;
    memcpy(k.out_index, h_idx + pp, sizeof(int64_t) * ((size_t)k.num_partitions + 1));
    k.out_total = h_idx[pp + k.num_partitions];
    k.status = S3S_OK;
    if (h_status[t] != 0 || k.out_total > k.dst_capacity) {
      k.status = S3S_E_CAPACITY;
      if (worst == S3S_OK)
        worst = fail(ctx, S3S_E_CAPACITY, "task %d: dst_capacity %lld too small for %lld output bytes", t,
                     (long long)k.dst_capacity, (long long)k.out_total);
      continue;
    }
    if (checksum_algo != S3S_CHECKSUM_NONE && k.num_partitions > 0)
      memcpy(k.out_checksums, h_sums + first_part[(size_t)t], sizeof(int64_t) * (size_t)k.num_partitions);
  }
  return verdict.finish(worst);
}

int64_t s3s_max_compressed_size_segments(const s3s_ctx* ctx, int codec, const int64_t* seg_offsets, int32_t n_segs) {
  return s3s_max_compressed_size(ctx, codec, seg_offsets, n_segs);  // a bound per stream, summed
}

int s3s_compress_map_output_segments_device(s3s_ctx* ctx, int codec, int checksum_algo, const uint8_t* d_src,
                                            const int64_t* seg_offsets, int32_t n_segs,
                                            const int32_t* part_first_seg, int32_t n, uint8_t* d_dst,
                                            int64_t dst_capacity, int64_t* out_index, int64_t* out_checksums,
                                            int64_t* out_total) {
  return compress_core(ctx, codec, checksum_algo, d_src, seg_offsets, n_segs, part_first_seg, n, d_dst,
                       dst_capacity, out_index, out_checksums, out_total);
}

int s3s_compress_map_output_segments(s3s_ctx* ctx, int codec, int checksum_algo, const uint8_t* src,
                                     const int64_t* seg_offsets, int32_t n_segs, const int32_t* part_first_seg,
                                     int32_t n, uint8_t* dst, int64_t dst_capacity, int64_t* out_index,
                                     int64_t* out_checksums, int64_t* out_total) {
  if (!ctx) return S3S_E_INVALID;
  ctx->err[0] = 0;
  if (n < 0 || n_segs < 0 || !seg_offsets || !part_first_seg)
    return fail(ctx, S3S_E_INVALID, "null offsets or negative partition / segment count");
  for (int32_t g = 0; g < n_segs; g++)
    if (seg_offsets[g + 1] < seg_offsets[g]) return fail(ctx, S3S_E_INVALID, "offsets not monotonic at %d", g);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int64_t first = n_segs > 0 ? seg_offsets[0] : 0;
  const int64_t total_u = n_segs > 0 ? seg_offsets[n_segs] - first : 0;
  const int64_t bound = s3s_max_compressed_size(ctx, codec, seg_offsets, n_segs);
  if (bound < 0) return fail(ctx, S3S_E_INVALID, "invalid codec or offsets");
  if ((total_u > 0 && !src) || (dst_capacity > 0 && !dst) || dst_capacity < 0) return fail(ctx, S3S_E_INVALID, "null/invalid host buffer");
  const int64_t dcap = dst_capacity < bound ? dst_capacity : bound;
  int rc;
  if ((rc = ensure(ctx, B_SRC, (size_t)total_u + 64))) return rc;
  if ((rc = ensure(ctx, B_DST, (size_t)dcap + 64))) return rc;
  if (total_u > 0)
    HIP_TRY(ctx, hipMemcpyAsync(ctx->buf[B_SRC].p, src + first, (size_t)total_u, hipMemcpyHostToDevice, ctx->stream));
  std::vector<int64_t> rebased((size_t)n_segs + 1);
  for (int32_t g = 0; g <= n_segs; g++) rebased[(size_t)g] = seg_offsets[g] - first;
  int64_t total = 0;
  rc = compress_core(ctx, codec, checksum_algo, dev<uint8_t>(ctx, B_SRC), rebased.data(), n_segs, part_first_seg, n,
                     dev<uint8_t>(ctx, B_DST), dcap, out_index, out_checksums, &total);
  if (out_total) *out_total = total;
  if (rc != S3S_OK) return rc;
  if (total > 0) HIP_TRY(ctx, hipMemcpy(dst, ctx->buf[B_DST].p, (size_t)total, hipMemcpyDeviceToHost));
  return S3S_OK;
}

int s3s_compress_map_output(s3s_ctx* ctx, int codec, int checksum_algo, const uint8_t* src,
                            const int64_t* src_offsets, int32_t n, uint8_t* dst,
                            int64_t dst_capacity, int64_t* out_index, int64_t* out_checksums,
                            int64_t* out_total) {
  if (!ctx) return S3S_E_INVALID;
  ctx->err[0] = 0;
  if (n < 0 || !src_offsets) return fail(ctx, S3S_E_INVALID, "null offsets or negative partition count");
  for (int32_t p = 0; p < n; p++)
    if (src_offsets[p + 1] < src_offsets[p]) return fail(ctx, S3S_E_INVALID, "src_offsets not monotonic at %d", p);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int64_t first = n > 0 ? src_offsets[0] : 0;
  const int64_t total_u = n > 0 ? src_offsets[n] - first : 0;
  const int64_t bound = s3s_max_compressed_size(ctx, codec, src_offsets, n);
  if (bound < 0) return fail(ctx, S3S_E_INVALID, "invalid codec or offsets");
  if ((total_u > 0 && !src) || (dst_capacity > 0 && !dst) || dst_capacity < 0) return fail(ctx, S3S_E_INVALID, "null/invalid host buffer");
  const int64_t dcap = dst_capacity < bound ? dst_capacity : bound;
  int rc;
  if ((rc = ensure(ctx, B_SRC, (size_t)total_u + 64))) return rc;
  if ((rc = ensure(ctx, B_DST, (size_t)dcap + 64))) return rc;
  // rebase the offsets so partition 0 starts at device offset 0; the upload itself happens inside the device path,
  // in chunks that overlap with the codec kernels (ctx->up_host)
  std::vector<int64_t> rebased((size_t)n + 1);
  for (int32_t p = 0; p <= n; p++) rebased[(size_t)p] = src_offsets[p] - first;
  int64_t total = 0;
  if (codec == S3S_CODEC_NONE && total_u > 0)  // (no codec kernel to overlap with)
    HIP_TRY(ctx, hipMemcpyAsync(ctx->buf[B_SRC].p, src + first, (size_t)total_u, hipMemcpyHostToDevice, ctx->stream));
  {
    // up_host is set for the device call only and cleared on every way out of it: a stale pointer would make a
    // later s3s_compress_map_output_device call on this context upload over the caller's d_src
    struct UpHostScope {
      s3s_ctx* c;
      ~UpHostScope() { c->up_host = nullptr; }
    } scope{ctx};
    ctx->up_host = (total_u > 0 && codec != S3S_CODEC_NONE) ? src + first : nullptr;
    rc = s3s_compress_map_output_device(ctx, codec, checksum_algo, dev<uint8_t>(ctx, B_SRC), rebased.data(), n,
                                        dev<uint8_t>(ctx, B_DST), dcap, out_index, out_checksums, &total);
  }
  if (out_total) *out_total = total;
  if (rc != S3S_OK) return rc;
  if (total > 0) HIP_TRY(ctx, hipMemcpy(dst, ctx->buf[B_DST].p, (size_t)total, hipMemcpyDeviceToHost));
  return S3S_OK;
}

int s3s_checksum_ranges_device(s3s_ctx* ctx, int algo, const uint8_t* d_data,
                               const int64_t* offsets, int32_t n, int64_t* out) {
  if (!ctx) return S3S_E_INVALID;
  ctx->err[0] = 0;
  if (n < 0 || !offsets || (n > 0 && !out)) return fail(ctx, S3S_E_INVALID, "null offsets/out or negative count");
  if (algo != S3S_CHECKSUM_ADLER32 && algo != S3S_CHECKSUM_CRC32 && algo != S3S_CHECKSUM_CRC32C)
    return fail(ctx, S3S_E_INVALID, "Unsupported shuffle checksum algorithm: %d", algo);
  if (n == 0) return S3S_OK;
  for (int32_t p = 0; p < n; p++)
    if (offsets[p + 1] < offsets[p]) return fail(ctx, S3S_E_INVALID, "offsets not monotonic at %d", p);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const size_t off_bytes = sizeof(int64_t) * (size_t)(n + 1), seg_bytes = sizeof(int32_t) * (size_t)(n + 1);
  const size_t o_seg = (off_bytes + 15) & ~size_t(15), o_out = (o_seg + seg_bytes + 15) & ~size_t(15);
  int rc;
  if ((rc = ensure_stage(ctx, o_out + sizeof(int64_t) * (size_t)n))) return rc;
  uint8_t* hs = static_cast<uint8_t*>(ctx->h_stage);
  int64_t* h_off = reinterpret_cast<int64_t*>(hs);
  int32_t* h_seg = reinterpret_cast<int32_t*>(hs + o_seg);
  int64_t* h_out = reinterpret_cast<int64_t*>(hs + o_out);
  int64_t segs = 0;
  for (int32_t p = 0; p < n; p++) {
    h_off[p] = offsets[p];
    h_seg[p] = (int32_t)segs;
    segs += worst_segs(offsets[p + 1] - offsets[p]);
    if (segs > 0x7fffff00ll) return fail(ctx, S3S_E_UNSUPPORTED, "range too large for one call");
  }
  h_off[n] = offsets[n];
  h_seg[n] = (int32_t)segs;
  if ((rc = ensure(ctx, B_OFFSETS, off_bytes))) return rc;
  if ((rc = ensure(ctx, B_SUMS, sizeof(int64_t) * (size_t)n))) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(ctx->buf[B_OFFSETS].p, h_off, off_bytes, hipMemcpyHostToDevice, ctx->stream));
  record(ctx, 0);
  if ((rc = run_checksum(ctx, algo, d_data, dev<int64_t>(ctx, B_OFFSETS), n, h_seg, dev<int64_t>(ctx, B_SUMS), offsets[n]))) return rc;
  record(ctx, 3);
  HIP_TRY(ctx, hipMemcpyAsync(h_out, ctx->buf[B_SUMS].p, sizeof(int64_t) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->profile) {
    float ms = 0;
    hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[3]);
    for (auto& v : ctx->stage_ms) v = 0;
    ctx->stage_ms[S3S_STAGE_TOTAL] = ctx->stage_ms[S3S_STAGE_CHECKSUM] = ms;
  }
  memcpy(out, h_out, sizeof(int64_t) * (size_t)n);
  return S3S_OK;
}

int s3s_checksum_ranges(s3s_ctx* ctx, int algo, const uint8_t* data, const int64_t* offsets,
                        int32_t n, int64_t* out) {
  if (!ctx) return S3S_E_INVALID;
  ctx->err[0] = 0;
  if (n < 0 || !offsets) return fail(ctx, S3S_E_INVALID, "null offsets or negative count");
  if (n == 0) return S3S_OK;
  const int64_t first = offsets[0], total = offsets[n] - first;
  if (total < 0 || (total > 0 && !data)) return fail(ctx, S3S_E_INVALID, "invalid range");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int rc;
  if ((rc = ensure(ctx, B_SRC, (size_t)total + 64))) return rc;
  if (total > 0)
    HIP_TRY(ctx, hipMemcpyAsync(ctx->buf[B_SRC].p, data + first, (size_t)total, hipMemcpyHostToDevice, ctx->stream));
  std::vector<int64_t> rebased((size_t)n + 1);
  for (int32_t p = 0; p <= n; p++) rebased[(size_t)p] = offsets[p] - first;
  return s3s_checksum_ranges_device(ctx, algo, dev<uint8_t>(ctx, B_SRC), rebased.data(), n, out);
}

#ifdef S3S_LZ4_TIMING
extern __device__ unsigned long long g_dec_dbg[16];
int s3s_debug_read_dec(unsigned long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dec_dbg), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_dec_dbg), z, sizeof z) != hipSuccess) return -1;
  }
  return 0;
}
extern __device__ unsigned long long g_bdec_dbg[16];
int s3s_debug_read_bdec(unsigned long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bdec_dbg), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_bdec_dbg), z, sizeof z) != hipSuccess) return -1;
  }
  return 0;
}
extern __device__ unsigned long long g_lz4_dbg[32];
int s3s_debug_read(unsigned long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lz4_dbg), 32 * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[32] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_lz4_dbg), z, sizeof z) != hipSuccess) return -1;
  }
  return 0;
}
#endif

}  // extern "C"
