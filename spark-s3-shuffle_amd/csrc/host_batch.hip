// host_batch.hip — the batched entry points for HOST buffers: what a Spark executor can actually call.
//
// The reference's call sites hold heap / direct buffers (shuffle/S3ShuffleMapOutputWriter.scala:91-118, 168-202;
// storage/S3ShuffleReader.scala:98-110), so the path a JVM reaches is host -> device -> host.  One call takes the map
// outputs (or fetched ranges) of several tasks and runs them as a three-stage pipeline over groups of ~64 MiB:
//
//     hb_in  stream   H2D of group g+1        (PCIe host -> device, page-locked source = plain DMA)
//     ctx    stream   codec + assemble + checksums of group g      (s3s_*_batch_device, unchanged)
//     hb_out stream   D2H of group g-1        (PCIe device -> host)
//
// with double-buffered device staging, so both PCIe directions and the compute stream are busy at once and the call
// runs at the speed of the slowest of the three — the upload of the uncompressed bytes on the map side, the download
// of the decoded bytes on the reduce side (PCIe Gen5 x16: ~55 GB/s per direction).  Results per task / range are
// exactly those of the single-task entry points.  Buffers from s3s_host_alloc move by DMA; pageable memory works but
// goes through the runtime's bounce buffers.
#include "s3s_ctx.h"

#include <mutex>

using namespace s3s;

namespace {

// uncompressed bytes per pipeline stage (S3S_HB_GROUP_MIB overrides it for tuning runs)
const int64_t kGroupBytes = [] {
  const char* e = getenv("S3S_HB_GROUP_MIB");
  const long v = e ? atol(e) : 0;
  return (int64_t)(v > 0 ? v : 64) << 20;  // measured 32 / 64 / 128 MiB: 45.1 / 45.5 / 44.9 GB/s host to host with one task thread
}();

// ---- copy arbiter (round 4) -----------------------------------------------------------------------
// Several task threads of one executor each run their own pipeline; with a pair of copy streams per CONTEXT their uploads
// (and downloads) are in flight side by side and share the PCIe link in small slices: every pipeline's copy stage takes
// n times longer while its codec stage waits (measured r03: 49.4 / 35.1 / 45.5 GB/s with 1 / 2 / 4 task threads).  The
// arbiter gives a DEVICE one upload lane and one download lane - two streams shared by all contexts of the process - so
// that copy groups of different pipelines go over the link ONE AFTER THE OTHER, each at the full rate, in the order they
// were enqueued.  Nothing queued on a lane ever waits for an event (a group's copies are enqueued only when its staging
// buffer is free and its source is ready), so the FIFO has no head-of-line blocking.  A mutex per lane keeps the copies of
// one group contiguous.  S3S_HB_SHARED_COPY=0 gives every context its own pair again.
struct CopyLanes {
  std::mutex create;
  hipStream_t in = nullptr, out = nullptr;
  std::mutex in_mu, out_mu;
};
CopyLanes g_lanes[64];
const bool kSharedCopy = [] {
  const char* e = getenv("S3S_HB_SHARED_COPY");
  return e ? atoi(e) != 0 : true;
}();

int hb_init(s3s_ctx* ctx) {
  if (kSharedCopy && !ctx->hb_in && !ctx->hb_out) {
    CopyLanes& L = g_lanes[ctx->device & 63];
    std::lock_guard<std::mutex> g(L.create);
    // The lanes are created at the HIGHEST stream priority: the runtime keeps one pool of hardware queues per priority, so a
    // lane never shares a hardware queue with a context's (normal-priority) compute stream - streams that share a queue run
    // one after the other.  Measured in the bench process (profiles/r04d, r04e): shared lanes at normal priority 49.6 / 52.8 /
    // 43.5 GB/s with 1 / 2 / 4 task threads, at the highest priority 49.5 / 53.3 / 50.9 (three runs within 1 GB/s); own
    // streams per context (round 3) 49.7 / 47.1 / 43.5.  S3S_HB_LANE_PRIO=0: normal priority.
    // S3S_HB_LANE_PRIO=2: upload lane highest, download lane LOWEST priority (three pools: the two lanes cannot share a queue either).
    int least = 0, greatest = 0;
    static const int prio = getenv("S3S_HB_LANE_PRIO") ? atoi(getenv("S3S_HB_LANE_PRIO")) : 1;
    if (prio) (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    if (!L.in) HIP_TRY(ctx, hipStreamCreateWithPriority(&L.in, hipStreamNonBlocking, greatest));
    if (!L.out) HIP_TRY(ctx, hipStreamCreateWithPriority(&L.out, hipStreamNonBlocking, prio == 2 ? least : greatest));
    ctx->hb_in = L.in;
    ctx->hb_out = L.out;
    ctx->hb_shared = true;
  }
  if (!ctx->hb_in) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->hb_in, hipStreamNonBlocking));
  if (!ctx->hb_out) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->hb_out, hipStreamNonBlocking));
  for (int i = 0; i < 2; i++) {
    if (!ctx->hb_ev_in[i]) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->hb_ev_in[i], hipEventDisableTiming));
    if (!ctx->hb_ev_out[i]) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->hb_ev_out[i], hipEventDisableTiming));
  }
  return S3S_OK;
}

// holds the lane of one direction while a group's copies are enqueued (shared lanes only)
struct LaneLock {
  std::mutex* m;
  LaneLock(s3s_ctx* ctx, bool upload) : m(nullptr) {
    if (ctx->hb_shared) {
      CopyLanes& L = g_lanes[ctx->device & 63];
      m = upload ? &L.in_mu : &L.out_mu;
      m->lock();
    }
  }
  ~LaneLock() {
    if (m) m->unlock();
  }
};

// everything THIS context has put on the copy lanes is done (its group events; a shared lane may hold other contexts' copies)
int hb_wait_own(s3s_ctx* ctx, bool in, bool out) {
  for (int i = 0; i < 2; i++) {
    if (in && ctx->hb_ev_in[i]) HIP_TRY(ctx, hipEventSynchronize(ctx->hb_ev_in[i]));
    if (out && ctx->hb_ev_out[i]) HIP_TRY(ctx, hipEventSynchronize(ctx->hb_ev_out[i]));
  }
  return S3S_OK;
}

// the staging buffers are (re)allocated only while nothing of this context is in flight
int hb_ensure(s3s_ctx* ctx, size_t in_bytes, size_t out_bytes) {
  if (in_bytes > ctx->buf[B_HB_IN0].cap || in_bytes > ctx->buf[B_HB_IN1].cap || out_bytes > ctx->buf[B_HB_OUT0].cap ||
      out_bytes > ctx->buf[B_HB_OUT1].cap) {
    int rc0 = hb_wait_own(ctx, true, true);
    if (rc0) return rc0;
  }
  int rc;
  if ((rc = ensure(ctx, B_HB_IN0, in_bytes))) return rc;
  if ((rc = ensure(ctx, B_HB_IN1, in_bytes))) return rc;
  if ((rc = ensure(ctx, B_HB_OUT0, out_bytes))) return rc;
  if ((rc = ensure(ctx, B_HB_OUT1, out_bytes))) return rc;
  return S3S_OK;
}

struct HbDrain {
  s3s_ctx* c;
  ~HbDrain() { (void)hb_wait_own(c, true, true); }
};

inline int64_t al256(int64_t x) { return (x + 255) & ~int64_t(255); }

}  // namespace

extern "C" {

int s3s_compress_map_outputs_batch(s3s_ctx* ctx, int codec, int checksum_algo, s3s_map_task* tasks, int32_t n_tasks) {
  if (!ctx) return S3S_E_INVALID;
  ctx->err[0] = 0;
  if (n_tasks < 0 || (n_tasks > 0 && !tasks)) return fail(ctx, S3S_E_INVALID, "null task array or negative count");
  if (n_tasks == 0) return S3S_OK;
  BatchVerdict<s3s_map_task> verdict(tasks, n_tasks);  // (a call-level failure leaves NOT_RUN marks: the return code is theirs)
  for (int32_t t = 0; t < n_tasks; t++) tasks[t].out_total = 0;
  std::vector<int64_t> u((size_t)n_tasks), cap((size_t)n_tasks);
  for (int32_t t = 0; t < n_tasks; t++) {
    const s3s_map_task& k = tasks[t];
    if (k.num_partitions < 0 || !k.src_offsets || !k.out_index || k.dst_capacity < 0)
      return fail(ctx, S3S_E_INVALID, "task %d: null offsets/index, negative partition count or capacity", t);
    for (int32_t p = 0; p < k.num_partitions; p++)
      if (k.src_offsets[p + 1] < k.src_offsets[p]) return fail(ctx, S3S_E_INVALID, "task %d: offsets not monotonic at %d", t, p);
    u[(size_t)t] = k.num_partitions > 0 ? k.src_offsets[k.num_partitions] - k.src_offsets[0] : 0;
    const int64_t bound = s3s_max_compressed_size(ctx, codec, k.src_offsets, k.num_partitions);
    if (bound < 0) return fail(ctx, S3S_E_INVALID, "invalid codec or offsets");
    cap[(size_t)t] = k.dst_capacity < bound ? k.dst_capacity : bound;
    if ((u[(size_t)t] > 0 && !k.d_src) || (k.dst_capacity > 0 && !k.d_dst)) return fail(ctx, S3S_E_INVALID, "task %d: null host buffer", t);
  }
  if (n_tasks == 1) {  // nothing to pipeline across tasks: the single-task path overlaps its own upload in chunks
    s3s_map_task& k = tasks[0];
    k.status = s3s_compress_map_output(ctx, codec, checksum_algo, k.d_src, k.src_offsets, k.num_partitions, k.d_dst,
                                       k.dst_capacity, k.out_index, k.out_checksums, &k.out_total);
    return verdict.finish(k.status);
  }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int rc;
  if ((rc = hb_init(ctx))) return rc;
  HbDrain drain{ctx};  // no DMA of this call survives it, whichever way it returns (the buffers are the caller's)
  // ---- groups of consecutive tasks, ~kGroupBytes of source each ----------------------------------------
  std::vector<int32_t> g0;  // first task of every group (+ end)
  size_t max_in = 64, max_out = 64;
  {
    int64_t in = 0, out = 0;
    for (int32_t t = 0; t < n_tasks; t++) {
      if (t == 0 || in + u[(size_t)t] > kGroupBytes) {
        g0.push_back(t);
        in = out = 0;
      }
      in += al256(u[(size_t)t]);
      out += al256(cap[(size_t)t]);
      if ((size_t)in + 64 > max_in) max_in = (size_t)in + 64;
      if ((size_t)out + 64 > max_out) max_out = (size_t)out + 64;
    }
    g0.push_back(n_tasks);
  }
  const int32_t n_groups = (int32_t)g0.size() - 1;
  if ((rc = hb_ensure(ctx, max_in, max_out))) return rc;
  uint8_t* d_in[2] = {dev<uint8_t>(ctx, B_HB_IN0), dev<uint8_t>(ctx, B_HB_IN1)};
  uint8_t* d_out[2] = {dev<uint8_t>(ctx, B_HB_OUT0), dev<uint8_t>(ctx, B_HB_OUT1)};
  // hb_in must not overwrite staging a previous call's kernels may still read: those calls synchronised ctx->stream
  auto upload = [&](int32_t g) -> int {
    const LaneLock lane(ctx, true);
    int64_t off = 0;
    for (int32_t t = g0[(size_t)g]; t < g0[(size_t)g + 1]; t++) {
      const s3s_map_task& k = tasks[t];
      if (u[(size_t)t] > 0)
        HIP_TRY(ctx, hipMemcpyAsync(d_in[g & 1] + off, k.d_src + k.src_offsets[0], (size_t)u[(size_t)t], hipMemcpyHostToDevice, ctx->hb_in));
      off += al256(u[(size_t)t]);
    }
    HIP_TRY(ctx, hipEventRecord(ctx->hb_ev_in[g & 1], ctx->hb_in));
    return S3S_OK;
  };
  int worst = S3S_OK;
  std::vector<s3s_map_task> dt;
  std::vector<std::vector<int64_t>> rebased;
  double stage_acc[S3S_STAGE_COUNT] = {};
  if ((rc = upload(0))) return rc;
  for (int32_t g = 0; g < n_groups; g++) {
    if (g + 1 < n_groups && (rc = upload(g + 1))) return rc;  // (its staging buffer was last read by group g-1: finished)
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->hb_ev_in[g & 1], 0));
    if (g >= 2) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->hb_ev_out[g & 1], 0));  // group g-2 still leaves through this buffer
    const int32_t t0 = g0[(size_t)g], t1 = g0[(size_t)g + 1];
    dt.assign((size_t)(t1 - t0), s3s_map_task{});
    rebased.assign((size_t)(t1 - t0), {});
    int64_t in_off = 0, out_off = 0;
    for (int32_t t = t0; t < t1; t++) {
      const s3s_map_task& k = tasks[t];
      std::vector<int64_t>& ro = rebased[(size_t)(t - t0)];
      ro.resize((size_t)k.num_partitions + 1);
      for (int32_t p = 0; p <= k.num_partitions; p++) ro[(size_t)p] = k.src_offsets[p] - k.src_offsets[0];
      s3s_map_task& d = dt[(size_t)(t - t0)];
      d.d_src = d_in[g & 1] + in_off;
      d.src_offsets = ro.data();
      d.num_partitions = k.num_partitions;
      d.d_dst = d_out[g & 1] + out_off;
      d.dst_capacity = cap[(size_t)t];
      d.out_index = k.out_index;
      d.out_checksums = k.out_checksums;
      in_off += al256(u[(size_t)t]);
      out_off += al256(cap[(size_t)t]);
    }
    rc = s3s_compress_map_outputs_batch_device(ctx, codec, checksum_algo, dt.data(), t1 - t0);  // (synchronises ctx->stream)
    if (rc != S3S_OK && rc != S3S_E_CAPACITY) return rc;
    if (ctx->profile)
      for (int s = 0; s < S3S_STAGE_COUNT; s++) stage_acc[s] += ctx->stage_ms[s];
    const LaneLock lane(ctx, false);
    for (int32_t t = t0; t < t1; t++) {
      s3s_map_task& k = tasks[t];
      const s3s_map_task& d = dt[(size_t)(t - t0)];
      k.out_total = d.out_total;
      k.status = d.status;
      if (k.status == S3S_OK && k.out_total > k.dst_capacity) k.status = S3S_E_CAPACITY;
      if (k.status != S3S_OK) {
        if (worst == S3S_OK) worst = k.status;
        continue;
      }
      if (k.out_total > 0)
        HIP_TRY(ctx, hipMemcpyAsync(k.d_dst, d.d_dst, (size_t)k.out_total, hipMemcpyDeviceToHost, ctx->hb_out));
    }
    HIP_TRY(ctx, hipEventRecord(ctx->hb_ev_out[g & 1], ctx->hb_out));
  }
  if ((rc = hb_wait_own(ctx, false, true))) return rc;
  if (ctx->profile)
    for (int s = 0; s < S3S_STAGE_COUNT; s++) ctx->stage_ms[s] = stage_acc[s];  // summed over the groups
  return verdict.finish(worst);
}

int s3s_decompress_ranges_batch(s3s_ctx* ctx, int codec, int checksum_algo, s3s_fetch_range* ranges, int32_t n_ranges) {
  if (!ctx) return S3S_E_INVALID;
  ctx->err[0] = 0;
  if (n_ranges < 0 || (n_ranges > 0 && !ranges)) return fail(ctx, S3S_E_INVALID, "null range array or negative count");
  if (n_ranges == 0) return S3S_OK;
  BatchVerdict<s3s_fetch_range> verdict(ranges, n_ranges);
  for (int32_t r = 0; r < n_ranges; r++) {
    ranges[r].out_len = 0;
    ranges[r].bad_partition = -1;
  }
  for (int32_t r = 0; r < n_ranges; r++) {
    const s3s_fetch_range& k = ranges[r];
    if (k.comp_len < 0 || k.dst_capacity < 0 || k.num_partitions < 0 || !k.part_offsets)
      return fail(ctx, S3S_E_INVALID, "range %d: negative length / capacity / partition count or null offsets", r);
    if ((k.comp_len > 0 && !k.d_comp) || (k.dst_capacity > 0 && !k.d_dst)) return fail(ctx, S3S_E_INVALID, "range %d: null host buffer", r);
  }
  if (n_ranges == 1) {
    s3s_fetch_range& k = ranges[0];
    k.bad_partition = -1;
    k.status = s3s_decompress_range(ctx, codec, checksum_algo, k.d_comp, k.comp_len, k.part_offsets, k.ref_checksums,
                                    k.num_partitions, k.d_dst, k.dst_capacity, &k.out_len, &k.bad_partition);
    return verdict.finish(k.status);
  }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int rc;
  if ((rc = hb_init(ctx))) return rc;
  HbDrain drain{ctx};
  // groups by DECODED capacity (the larger side): ~2 x kGroupBytes of output per stage
  std::vector<int32_t> g0;
  size_t max_in = 64, max_out = 64;
  {
    int64_t in = 0, out = 0;
    for (int32_t r = 0; r < n_ranges; r++) {
      if (r == 0 || out + ranges[r].dst_capacity > 2 * kGroupBytes) {
        g0.push_back(r);
        in = out = 0;
      }
      in += al256(ranges[r].comp_len);
      out += al256(ranges[r].dst_capacity);
      if ((size_t)in + 64 > max_in) max_in = (size_t)in + 64;
      if ((size_t)out + 64 > max_out) max_out = (size_t)out + 64;
    }
    g0.push_back(n_ranges);
  }
  const int32_t n_groups = (int32_t)g0.size() - 1;
  if ((rc = hb_ensure(ctx, max_in, max_out))) return rc;
  uint8_t* d_in[2] = {dev<uint8_t>(ctx, B_HB_IN0), dev<uint8_t>(ctx, B_HB_IN1)};
  uint8_t* d_out[2] = {dev<uint8_t>(ctx, B_HB_OUT0), dev<uint8_t>(ctx, B_HB_OUT1)};
  auto upload = [&](int32_t g) -> int {
    const LaneLock lane(ctx, true);
    int64_t off = 0;
    for (int32_t r = g0[(size_t)g]; r < g0[(size_t)g + 1]; r++) {
      const s3s_fetch_range& k = ranges[r];
      if (k.comp_len > 0)
        HIP_TRY(ctx, hipMemcpyAsync(d_in[g & 1] + off, k.d_comp, (size_t)k.comp_len, hipMemcpyHostToDevice, ctx->hb_in));
      off += al256(k.comp_len);
    }
    HIP_TRY(ctx, hipEventRecord(ctx->hb_ev_in[g & 1], ctx->hb_in));
    return S3S_OK;
  };
  int worst = S3S_OK;
  std::vector<s3s_fetch_range> dr;
  double stage_acc[S3S_STAGE_COUNT] = {};
  if ((rc = upload(0))) return rc;
  for (int32_t g = 0; g < n_groups; g++) {
    if (g + 1 < n_groups && (rc = upload(g + 1))) return rc;
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->hb_ev_in[g & 1], 0));
    if (g >= 2) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->hb_ev_out[g & 1], 0));
    const int32_t r0 = g0[(size_t)g], r1 = g0[(size_t)g + 1];
    dr.assign((size_t)(r1 - r0), s3s_fetch_range{});
    int64_t in_off = 0, out_off = 0;
    for (int32_t r = r0; r < r1; r++) {
      const s3s_fetch_range& k = ranges[r];
      s3s_fetch_range& d = dr[(size_t)(r - r0)];
      d = k;
      d.d_comp = d_in[g & 1] + in_off;
      d.d_dst = d_out[g & 1] + out_off;
      in_off += al256(k.comp_len);
      out_off += al256(k.dst_capacity);
    }
    rc = s3s_decompress_ranges_batch_device(ctx, codec, checksum_algo, dr.data(), r1 - r0);  // (synchronises ctx->stream)
    if (rc == S3S_E_INVALID || rc == S3S_E_HIP || rc == S3S_E_NOMEM) return rc;
    if (ctx->profile)
      for (int s = 0; s < S3S_STAGE_COUNT; s++) stage_acc[s] += ctx->stage_ms[s];
    const LaneLock lane(ctx, false);
    for (int32_t r = r0; r < r1; r++) {
      s3s_fetch_range& k = ranges[r];
      const s3s_fetch_range& d = dr[(size_t)(r - r0)];
      k.out_len = d.out_len;
      k.bad_partition = d.bad_partition;
      k.status = d.status;
      if (k.status != S3S_OK) {
        if (worst == S3S_OK) worst = k.status;
        continue;
      }
      if (k.out_len > 0)
        HIP_TRY(ctx, hipMemcpyAsync(k.d_dst, d.d_dst, (size_t)k.out_len, hipMemcpyDeviceToHost, ctx->hb_out));
    }
    HIP_TRY(ctx, hipEventRecord(ctx->hb_ev_out[g & 1], ctx->hb_out));
  }
  if ((rc = hb_wait_own(ctx, false, true))) return rc;
  if (ctx->profile)
    for (int s = 0; s < S3S_STAGE_COUNT; s++) ctx->stage_ms[s] = stage_acc[s];
  return verdict.finish(worst);
}

}  // extern "C"
