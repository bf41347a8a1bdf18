// s3s_internal.h — shared declarations of the MI355X shuffle-block codec library.
//
// Layout of the device workspace for one compress call (all in HBM, owned by s3s_ctx):
//   items[n_items]        16 B plan records (host-built, uploaded once per call)
//   part_first[N+1]       first item of every partition
//   slots[n_chunks]       one SLOT per codec chunk: frame header right-aligned in the first
//                         32 B, payload from +32 (16 B aligned)  -> written by the codec kernel
//   item_size[n_items]    bytes each item contributes to the .data image
//   item_off[n_items+1]   exclusive scan of item_size                 -> scan kernel
//   d_index[N+1], d_sums[N]                                           -> D2H at the end
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/s3shuffle_codec.h"

namespace s3s {

constexpr int kWave = 64;
constexpr int kMaxBlock = 32768;          // largest codec chunk the LDS-resident kernels take (Snappy; LZ4's default)
constexpr int kLz4MaxBlock = 65536;       // largest LZ4 chunk of the map side (round 4): liblz4 parses inputs below 65 547 bytes
                                          // with the SAME 8192 x u16 table (byU16) - positions just need all 16 bits - so the
                                          // window engine takes them as it is; from 64 KiB + 11 on liblz4 switches to its
                                          // 4096 x u32 table with a 5-byte hash: another parse, not built
constexpr int kBatchMaxBlock = 1 << 25;   // largest LZ4Block frame the batch decoder takes (lz4-java's MAX_BLOCK_SIZE)
constexpr int kSlotHeader = 32;           // bytes reserved in front of a slot's payload
constexpr int kLz4FrameHeader = 21;       // "LZ4Block" + token + 3 x i32
constexpr int kSnappyStreamHeader = 16;
constexpr uint32_t kLz4BlockSeed = 0x9747b28cu;

// plan record kinds
enum : int32_t {
  kItemLz4Chunk = 0,     // LZ4Block data frame (header + payload, payload may be RAW)
  kItemLz4End = 1,       // 21-byte end-of-stream frame
  kItemSnappyHeader = 2, // 16-byte SnappyOutputStream header
  kItemSnappyChunk = 3   // i32 BE length + raw snappy
};

struct Item {
  int64_t src_off;  // offset of the chunk in the uncompressed source (chunks only)
  int32_t len;      // uncompressed chunk length (chunks only)
  int32_t kind;     // low 8 bits: kind; bits 8..15: LZ4Block level nibble; bits 16..: unused
  int32_t chunk;    // slot index for chunk items, -1 otherwise
  int32_t part;     // owning partition
};

// item_size bit 31 marks an LZ4 chunk stored RAW (payload copied from the source)
constexpr uint32_t kRawFlag = 0x80000000u;

// ---- kernel launchers (each defined next to its kernel) ---------------------------------
// LZ4: compress every kItemLz4Chunk item into its slot, write the frame header, item_size.
//   d_item_check: per-item xxHash32 workspace (written by the xxh32 pre-pass)
//   variant 0: chunk staged in LDS (3 wavefronts/CU); 1: chunk read through L1/L2 (10/CU);
//   2: as 1 plus the window-speculative parse (default)
//   slot_stride: bytes between slots (kSlotHeader + the block size rounded up to 16)
void launch_lz4_compress(const uint8_t* d_src, const Item* d_items, int32_t n_items,
                         uint32_t* d_item_check, uint8_t* d_slots, int32_t slot_stride, uint32_t* d_item_size, uint32_t* d_work, int resident_waves,
                         int variant, hipStream_t st, hipEvent_t after_hash = nullptr);
// Snappy: same for kItemSnappyChunk.
bool snappy_compress_available();
//   slot_stride: bytes between slots (a raw snappy block can be larger than its chunk)
void launch_snappy_compress(const uint8_t* d_src, const Item* d_items, int32_t n_items,
                            uint8_t* d_slots, int64_t slot_stride, uint32_t* d_item_size,
                            int variant, hipStream_t st);  // variant 0: batch only, 1: exact windows first
// exclusive scan of item sizes + partition index extraction
void launch_scan_items(const Item* d_items, const uint32_t* d_item_size, int32_t n_items,
                       int64_t* d_item_off, const int32_t* d_part_first, int32_t n_parts,
                       int64_t* d_index, hipStream_t st);
// copy every item to its place in the .data image; sets *d_status != 0 on capacity overflow
void launch_gather_items(const uint8_t* d_src, const Item* d_items, int32_t n_items,
                         const uint8_t* d_slots, int64_t slot_stride, const uint32_t* d_item_size,
                         const int64_t* d_item_off, uint8_t* d_dst, int64_t dst_capacity,
                         int32_t* d_status, hipStream_t st);
// per-range Adler32 / CRC32: out[i] over data[offsets[i], offsets[i+1])
//   d_seg_start[n+1]: prefix of worst-case 16 KiB segment counts (host-built from upper bounds)
//   d_tables: constant tables (checksum_tables_build), d_partial: 4 x uint32 per segment slot
size_t checksum_tables_bytes();
void checksum_tables_build(void* host_buf);
//   max_segs_per_range / d_partial2: a range of more than kChecksumFoldFrom segments has groups of kChecksumFoldGroup
//   segments folded first (checksum_fold_kernel); d_partial2 then holds 4 x uint32 per (range, group):
//   n x checksum_fold_groups(n, max_segs_per_range) entries.  nullptr / 0: no folding (one wavefront folds a whole range).
void launch_checksum_with_tables(int algo, const uint8_t* d_data, const int64_t* d_offsets,
                                 int32_t n, const int32_t* d_seg_start, int32_t total_segs,
                                 const void* d_tables, uint32_t* d_partial, int64_t* d_out,
                                 int64_t data_len /* bytes readable at d_data */, hipStream_t st,
                                 int32_t max_segs_per_range = 0, uint32_t* d_partial2 = nullptr);
// ---- batched tails (round 6) --------------------------------------------------------------------------------------------
// A batched call used to launch its small kernels PER TASK (scan, gather, memset + segments + combine of the checksums: six
// launches per map task, 2 - 5 us each plus the gap between them): 32 blocks of 8 MiB in one call spent 2.2 of its 5.6 ms
// there (profiles/r06e_*).  One descriptor per task / fetched range, uploaded with the plan; every tail kernel is launched
// ONCE per call and finds its task by a binary search over the descriptors (a handful of scalar loads).
struct TaskTail {
  int32_t first_item, n_items;   // compress: the task's plan records (scan / gather)
  int32_t first_pp, n_parts;     // its partitions in the packed arrays that hold n + 1 entries per task (index, part_first, seg_start)
  int32_t first_part;            // ... and in the arrays that hold n entries per task (checksums out)
  int32_t first_seg;             // its first checksum segment slot in the call's partial array (seg_start is task-relative)
  int32_t n_segs, pad;
  const uint8_t* data;           // what the checksums run over: the task's .data image / the fetched range
  int64_t data_len;              // bytes readable there
  uint8_t* dst;                  // compress: the .data image the gather writes
  int64_t dst_capacity;
};
void launch_scan_items_batch(const TaskTail* d_tails, int32_t n_tasks, const uint32_t* d_item_size, int64_t* d_item_off,
                             const int32_t* d_part_first, int64_t* d_index, hipStream_t st);
void launch_gather_items_batch(const TaskTail* d_tails, int32_t n_tasks, int32_t n_items_total, const uint8_t* d_src, const Item* d_items,
                               const uint8_t* d_slots, int64_t slot_stride, const uint32_t* d_item_size, const int64_t* d_item_off,
                               int32_t* d_status, hipStream_t st);
// checksums of every task's ranges: offsets / seg_start packed (n + 1 per task), partial zeroed here, out packed (n per task)
void launch_checksum_batch(int algo, const TaskTail* d_tails, int32_t n_tasks, int32_t total_segs, int32_t total_parts,
                           const int64_t* d_offsets, const int32_t* d_seg_start, const void* d_tables, uint32_t* d_partial,
                           int64_t* d_out, hipStream_t st);
constexpr int kChecksumSegBytes = 16384;
constexpr int kChecksumFoldGroup = 256;   // segments per folded group (4 MiB)
constexpr int kChecksumFoldFrom = 2048;   // a range with more segments than this (32 MiB) is folded in two levels
// groups per range of the folded layout, 0 = do not fold (no range is large, or n x groups would not be small)
inline int32_t checksum_fold_groups(int32_t n, int32_t max_segs_per_range) {
  if (max_segs_per_range <= kChecksumFoldFrom) return 0;
  const int64_t g = ((int64_t)max_segs_per_range + kChecksumFoldGroup - 1) / kChecksumFoldGroup;
  return (int64_t)n * g <= (1 << 20) ? (int32_t)g : 0;
}

// reduce side ------------------------------------------------------------------------------
struct Frame {        // one discovered codec frame
  int64_t comp_off;   // payload offset in the compressed range
  int32_t comp_len;   // payload bytes
  int32_t orig_len;   // decoded bytes
  uint32_t check;     // LZ4Block: xxh32 & 0x0FFFFFFF; snappy: unused
  int32_t method;     // 0x10 raw / 0x20 lz4 (LZ4Block); 1 = snappy chunk; LZF: 0x10 stored chunk / 2 compressed chunk
};
// LZ4Block frame discovery over the whole range (see lz4_decompress.hip): 64 KiB tiles are
// walked speculatively, resolved into the true chain, then emitted in stream order.
//   d_spec_count[n_tiles] (as uint32) -> d_frame_base[n_tiles+1] (exclusive scan; last = n_frames)
int32_t lz4_tile_count(int64_t comp_len);
// one fetched range of a batched reduce-side call (s3s_decompress_ranges_batch_device): the discovery kernels of ALL
// ranges run as one launch each (block -> range through a host-built tile map)
struct LzRange {
  const uint8_t* comp;
  int64_t comp_len;
  int32_t n_tiles, tile0;  // tiles of this range, index of its first tile in the batch-wide tile list
  int64_t *spec_entry, *spec_exit, *true_entry, *frame_base;
  int32_t* spec_count;
  int32_t* status;         // per-range status word
  int64_t* result;         // [0] = frames of the range (phase 1), [1] = decoded bytes (phase 2)
  // phase 2, filled by the host once the frame counts are known
  Frame* frames;
  uint32_t* frame_orig;
  int64_t* frame_out;      // n_frames + 1 offsets relative to the range's destination
  int64_t* out_abs;        // absolute output address per frame (the one decode launch runs with dst = nullptr)
  int64_t n_frames;
  int64_t dst_base, dst_capacity;
  int32_t skip, pad;       // the range failed an earlier check: its frames become empty
};
void launch_lz4_discover_batch(const LzRange* d_ranges, int32_t n_ranges, const int32_t* d_tile_range,
                               int32_t total_tiles, hipStream_t st);
void launch_lz4_frames_batch(LzRange* d_ranges, int32_t n_ranges, const int32_t* d_tile_range, int32_t total_tiles,
                             hipStream_t st);
void launch_lz4_discover(const uint8_t* d_comp, int64_t comp_len, int32_t n_tiles,
                         int64_t* d_spec_entry, int64_t* d_spec_exit, int32_t* d_spec_count,
                         int64_t* d_true_entry, int64_t* d_frame_base, int32_t* d_status,
                         hipStream_t st);
//   writes d_frames[n_frames], d_frame_orig[n_frames] and d_frame_out[n_frames+1] (scan of
//   decoded sizes; last = total decoded bytes)
void launch_lz4_emit_frames(const uint8_t* d_comp, int64_t comp_len, int32_t n_tiles,
                            const int64_t* d_true_entry, const int64_t* d_frame_base,
                            Frame* d_frames, uint32_t* d_frame_orig, int64_t n_frames,
                            int64_t* d_frame_out, int32_t* d_status, hipStream_t st);
//   variant 0: frame staged in LDS (3 frames per CU); 1: straight to global memory (no LDS)
//   after_decode (optional): recorded between the decode kernel and the frame-check kernel of variant 4
void launch_lz4_decompress(const uint8_t* d_comp, const Frame* d_frames, int32_t n_frames,
                           const int64_t* d_frame_out, uint8_t* d_dst, int32_t* d_status,
                           int variant, hipStream_t st, hipEvent_t after_decode = nullptr);
//   variant 4: batches of sequences, one lane per sequence (lz4_decode_batch.hip)
void launch_lz4_decompress_batch(const uint8_t* d_comp, const Frame* d_frames, int32_t n_frames,
                                 const int64_t* d_frame_out, uint8_t* d_dst, int32_t* d_status,
                                 hipStream_t st, hipEvent_t after_decode = nullptr);
void launch_snappy_decompress_batch(const uint8_t* d_comp, const Frame* d_frames, int32_t n_frames,
                                    const int64_t* d_frame_out, uint8_t* d_dst, int32_t* d_status,
                                    hipStream_t st);
void launch_lzf_decompress_batch(const uint8_t* d_comp, const Frame* d_frames, int32_t n_frames,
                                 const int64_t* d_frame_out, uint8_t* d_dst, int32_t* d_status, hipStream_t st);
// Snappy (SnappyInputStream framing): chunks are chained by their i32 BE length only, so the walk
// is one lane per partition (each non-empty partition is one or more complete streams, each
// starting with the 16-byte header).  Pass 1 counts, pass 2 (after a scan of the counts) writes
// d_frames / d_frame_orig at d_frame_base[partition].
// chunk_format: kChunkSnappy, or kChunkLzf (round 4: compress-lzf chunks 'Z' 'V' type | len BE [| ulen BE]; stored chunks
// become frames with method 0x10, compressed ones method 2) - the same two passes, the same batch decoder behind them.
enum { kChunkSnappy = 0, kChunkLzf = 1 };
void launch_snappy_count_frames(const uint8_t* d_comp, const int64_t* d_part_off, int32_t n_parts,
                                uint32_t* d_part_nframes, int32_t* d_status, hipStream_t st, int chunk_format = kChunkSnappy);
void launch_snappy_emit_frames(const uint8_t* d_comp, const int64_t* d_part_off, int32_t n_parts,
                               const int64_t* d_frame_base, Frame* d_frames, uint32_t* d_frame_orig,
                               int32_t* d_status, hipStream_t st, int chunk_format = kChunkSnappy);
//   variant 0: block staged in LDS; otherwise the VALU ring decoder
void launch_snappy_decompress(const uint8_t* d_comp, const Frame* d_frames, int32_t n_frames,
                              const int64_t* d_frame_out, uint8_t* d_dst, int32_t* d_status,
                              int variant, hipStream_t st, int chunk_format = kChunkSnappy);
void launch_scan_u32(const uint32_t* d_in, int64_t n, int64_t* d_out, hipStream_t st);

// ---- device helpers shared by several kernels --------------------------------------------
__device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) {
  return __builtin_amdgcn_alignbit(x, x, 32 - r);
}

// unaligned 32-bit read from LDS: two aligned dwords + v_alignbyte
__device__ __forceinline__ uint32_t lds_rd32(const uint8_t* base, int pos) {
  const uint32_t* p = reinterpret_cast<const uint32_t*>(base + (pos & ~3));
  const uint32_t lo = p[0], hi = p[1];
  return __builtin_amdgcn_alignbyte(hi, lo, (uint32_t)pos & 3u);
}

}  // namespace s3s
