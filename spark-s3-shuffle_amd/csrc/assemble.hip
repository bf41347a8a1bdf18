// assemble.hip — turn per-chunk codec output into the exact .data byte image.
//
// The reference writes partitions strictly in ascending order into ONE object and records
// each partition's byte count (S3ShuffleMapOutputWriter.scala:67-83, :197-201); the index is
// the running sum with a leading zero (S3ShuffleHelper.scala:44-47).  Here every frame of
// every partition was produced independently into a fixed-stride slot, so the layout step is
// an exclusive scan over item sizes followed by a gather:
//   scan_items    item_off[i] = sum_{j<i} size_j ; index[p] = item_off[first item of p]
//   gather_items  dst[item_off[i] ...] = header | payload   (payload from the slot, or from the
//                 uncompressed source for LZ4 frames stored RAW)
// Both are pure HBM streaming: (compressed bytes) read + written once.
#include "s3s_internal.h"

namespace s3s {
namespace {

// Exclusive scan of the item sizes + partition index.  ONE wavefront and NO LDS on purpose: the codec
// kernels of the other task threads book all 160 KiB of every CU's LDS (10 x 16 KiB tables), and a
// workgroup that needs even a few bytes of it waits for one of their ~1 ms chunks to finish (the
// 1024-thread LDS version of this kernel took 9 us alone and 234 us on average next to a second
// stream, profiles/r01d).  256 items per step (4 per lane), carried in a register.
__device__ __forceinline__ void scan_items_body(
    const uint32_t* __restrict__ item_size, int32_t n_items, int64_t* item_off,
    const int32_t* __restrict__ part_first, int32_t n_parts, int64_t* __restrict__ index) {
  const int lane = threadIdx.x;
  int64_t carry = 0;
  for (int32_t tile = 0; tile < n_items; tile += 4 * kWave) {
    const int32_t i0 = tile + 4 * lane;
    int64_t x[4];
#pragma unroll
    for (int k = 0; k < 4; k++) x[k] = i0 + k < n_items ? (int64_t)(item_size[i0 + k] & ~kRawFlag) : 0;
    const int64_t mine = x[0] + x[1] + x[2] + x[3];
    int64_t inc = mine;  // inclusive scan of the lane sums
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
      const int64_t y = __shfl_up(inc, d);
      if (lane >= d) inc += y;
    }
    int64_t off = carry + inc - mine;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (i0 + k < n_items) item_off[i0 + k] = off;
      off += x[k];
    }
    carry += __shfl(inc, kWave - 1);
  }
  if (lane == 0) item_off[n_items] = carry;
  __threadfence();  // the index pass below reads what this wavefront just wrote
  // partition index: offset of the partition's first item (== total for trailing empties)
  for (int32_t p = lane; p <= n_parts; p += kWave)
    index[p] = __builtin_nontemporal_load(&item_off[part_first[p]]);
}

__global__ __launch_bounds__(kWave) void scan_items_kernel(
    const uint32_t* __restrict__ item_size, int32_t n_items, int64_t* item_off,
    const int32_t* __restrict__ part_first, int32_t n_parts, int64_t* __restrict__ index) {
  scan_items_body(item_size, n_items, item_off, part_first, n_parts, index);
}

// the task that owns global index i of a packed array: the last t with first(t) <= i  (wave-uniform: scalar loads)
template <typename F>
__device__ __forceinline__ int owner_task(int32_t n_tasks, int32_t i, F first) {
  int lo = 0, hi = n_tasks;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (first(mid) <= i) lo = mid; else hi = mid;
  }
  return lo;
}

// one wavefront per TASK of a batched call (item_off holds one extra entry per task: + t)
__global__ __launch_bounds__(kWave) void scan_items_batch_kernel(
    const TaskTail* __restrict__ tails, int32_t n_tasks, const uint32_t* __restrict__ item_size, int64_t* item_off,
    const int32_t* __restrict__ part_first, int64_t* __restrict__ index) {
  const int t = blockIdx.x;
  if (t >= n_tasks) return;
  const TaskTail k = tails[t];
  scan_items_body(item_size + k.first_item, k.n_items, item_off + k.first_item + t, part_first + k.first_pp, k.n_parts,
                  index + k.first_pp);
}

constexpr int kGatherThreads = 256;

// n bytes global -> global; 16-byte stores on the destination's alignment.
__device__ __forceinline__ void copy_bytes(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src,
                                           int n, int tid) {
  int head = (int)((16u - (uint32_t)(uintptr_t)dst) & 15u);
  head = head < n ? head : n;
  if (tid < head) dst[tid] = src[tid];
  const int nvec = (n - head) >> 4;
  uint4* d16 = reinterpret_cast<uint4*>(dst + head);
  const uint8_t* s = src + head;
  for (int v = tid; v < nvec; v += kGatherThreads) {
    uint4 x;
    __builtin_memcpy(&x, s + 16 * v, 16);  // source alignment is arbitrary
    d16[v] = x;
  }
  const int done = head + 16 * nvec;
  if (tid < n - done) dst[done + tid] = src[done + tid];
}

__device__ __forceinline__ void gather_item_body(
    const uint8_t* __restrict__ src, const Item* __restrict__ items, int32_t n_items,
    const uint8_t* __restrict__ slots, int64_t slot_stride, const uint32_t* __restrict__ item_size,
    const int64_t* __restrict__ item_off, uint8_t* __restrict__ dst, int64_t dst_capacity,
    int32_t* __restrict__ status, int it) {
  if (it >= n_items) return;
  const Item item = items[it];
  const uint32_t sz = item_size[it];
  const int n = (int)(sz & ~kRawFlag);
  const int64_t off = item_off[it];
  const int tid = threadIdx.x;
  if (off + n > dst_capacity) {
    if (tid == 0) atomicExch(status, S3S_E_CAPACITY);
    return;
  }
  uint8_t* d = dst + off;
  const int kind = item.kind & 0xff;
  if (kind == kItemLz4End) {
    // LZ4BlockOutputStream.finish(): magic | RAW|level | 0 | 0 | 0
    if (tid < kLz4FrameHeader) {
      const uint64_t magic = 0x6b636f6c42345a4cull;
      uint32_t b = 0;
      if (tid < 8) b = (uint32_t)(magic >> (8 * tid));
      else if (tid == 8) b = 0x10u | ((uint32_t)(item.kind >> 8) & 0x0Fu);
      d[tid] = (uint8_t)b;
    }
    return;
  }
  if (kind == kItemSnappyHeader) {
    if (tid < kSnappyStreamHeader) {
      const uint64_t lo = 0x00595050414e5382ull;  // 0x82 'S' 'N' 'A' 'P' 'P' 'Y' 0x00
      uint32_t b;
      if (tid < 8) b = (uint32_t)(lo >> (8 * tid));
      else b = (tid == 11 || tid == 15) ? 1u : 0u;  // version = 1, compatible version = 1 (BE)
      d[tid] = (uint8_t)b;
    }
    return;
  }
  const uint8_t* slot = slots + (size_t)item.chunk * (size_t)slot_stride;
  if (kind == kItemLz4Chunk) {
    if (sz & kRawFlag) {
      copy_bytes(d, slot + (kSlotHeader - kLz4FrameHeader), kLz4FrameHeader, tid);
      copy_bytes(d + kLz4FrameHeader, src + item.src_off, n - kLz4FrameHeader, tid);
    } else {
      copy_bytes(d, slot + (kSlotHeader - kLz4FrameHeader), n, tid);
    }
  } else {  // kItemSnappyChunk: i32 BE length right-aligned in the slot header, then payload
    copy_bytes(d, slot + (kSlotHeader - 4), n, tid);
  }
}

__global__ __launch_bounds__(kGatherThreads) void gather_items_kernel(
    const uint8_t* __restrict__ src, const Item* __restrict__ items, int32_t n_items,
    const uint8_t* __restrict__ slots, int64_t slot_stride, const uint32_t* __restrict__ item_size,
    const int64_t* __restrict__ item_off, uint8_t* __restrict__ dst, int64_t dst_capacity,
    int32_t* __restrict__ status) {
  gather_item_body(src, items, n_items, slots, slot_stride, item_size, item_off, dst, dst_capacity, status, (int)blockIdx.x);
}

// one workgroup per item of EVERY task of a batched call
__global__ __launch_bounds__(kGatherThreads) void gather_items_batch_kernel(
    const TaskTail* __restrict__ tails, int32_t n_tasks, int32_t n_items_total, const uint8_t* __restrict__ src,
    const Item* __restrict__ items, const uint8_t* __restrict__ slots, int64_t slot_stride,
    const uint32_t* __restrict__ item_size, const int64_t* __restrict__ item_off, int32_t* __restrict__ status) {
  const int g = blockIdx.x;
  if (g >= n_items_total) return;
  const int t = owner_task(n_tasks, g, [&](int m) { return tails[m].first_item; });
  const TaskTail k = tails[t];
  gather_item_body(src, items + k.first_item, k.n_items, slots, slot_stride, item_size + k.first_item,
                   item_off + k.first_item + t, k.dst, k.dst_capacity, status + t, g - k.first_item);
}

}  // namespace

void launch_scan_items_batch(const TaskTail* d_tails, int32_t n_tasks, const uint32_t* d_item_size, int64_t* d_item_off,
                             const int32_t* d_part_first, int64_t* d_index, hipStream_t st) {
  if (n_tasks <= 0) return;
  hipLaunchKernelGGL(scan_items_batch_kernel, dim3((unsigned)n_tasks), dim3(kWave), 0, st, d_tails, n_tasks, d_item_size, d_item_off,
                     d_part_first, d_index);
}

void launch_gather_items_batch(const TaskTail* d_tails, int32_t n_tasks, int32_t n_items_total, const uint8_t* d_src, const Item* d_items,
                               const uint8_t* d_slots, int64_t slot_stride, const uint32_t* d_item_size, const int64_t* d_item_off,
                               int32_t* d_status, hipStream_t st) {
  if (n_items_total <= 0) return;
  hipLaunchKernelGGL(gather_items_batch_kernel, dim3((unsigned)n_items_total), dim3(kGatherThreads), 0, st, d_tails, n_tasks,
                     n_items_total, d_src, d_items, d_slots, slot_stride, d_item_size, d_item_off, d_status);
}

void launch_scan_items(const Item*, const uint32_t* d_item_size, int32_t n_items,
                       int64_t* d_item_off, const int32_t* d_part_first, int32_t n_parts,
                       int64_t* d_index, hipStream_t st) {
  hipLaunchKernelGGL(scan_items_kernel, dim3(1), dim3(kWave), 0, st, d_item_size, n_items,
                     d_item_off, d_part_first, n_parts, d_index);
}

void launch_gather_items(const uint8_t* d_src, const Item* d_items, int32_t n_items,
                         const uint8_t* d_slots, int64_t slot_stride, const uint32_t* d_item_size,
                         const int64_t* d_item_off, uint8_t* d_dst, int64_t dst_capacity,
                         int32_t* d_status, hipStream_t st) {
  if (n_items <= 0) return;
  hipLaunchKernelGGL(gather_items_kernel, dim3((unsigned)n_items), dim3(kGatherThreads), 0, st,
                     d_src, d_items, n_items, d_slots, slot_stride, d_item_size, d_item_off, d_dst,
                     dst_capacity, d_status);
}

}  // namespace s3s
