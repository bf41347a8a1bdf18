/*
 * s3shuffle_codec.h — C-ABI of the MI355X-native shuffle-block codec path.
 *
 * This is the drop-in boundary for ONE hot path of IBM/spark-s3-shuffle: the map-side
 * compress + checksum of a map task's shuffle partitions and the reduce-side verify +
 * decompress of a fetched block range.  Everything above it (ShuffleManager /
 * ShuffleDataIO SPI, object-store I/O, lifecycle) stays on the JVM unchanged; a thin JNI
 * shim binds exactly these entry points.  Plain pointers and sizes only.
 *
 * Reference interfaces each entry point replaces (paths relative to the reference repo,
 * src/main/scala/org/apache/spark/...):
 *
 *   s3s_compress_map_output        the [EXT] LZ4BlockOutputStream / SnappyOutputStream +
 *                                  MutableCheckedOutputStream stage that feeds
 *                                  shuffle/S3ShuffleMapOutputWriter.scala:168-202
 *                                  (S3ShuffleOutputStream.write) and whose per-partition
 *                                  lengths / checksums are persisted by
 *                                  shuffle/S3ShuffleMapOutputWriter.scala:91-118
 *                                  (commitAllPartitions) through
 *                                  shuffle/helper/S3ShuffleHelper.scala:44-59
 *                                  (writePartitionLengths / writeChecksum); also the
 *                                  pre-built spill file + lengths + checksums handed to
 *                                  shuffle/S3SingleSpillShuffleMapOutputWriter.scala:24-64.
 *   s3s_checksum_ranges            shuffle/helper/S3ShuffleHelper.scala:94-103
 *                                  (createChecksumAlgorithm) as used by
 *                                  storage/S3ChecksumValidationStream.scala:54-66.
 *   s3s_decompress_range           storage/S3ChecksumValidationStream.scala:17-92 (verify)
 *                                  + the [EXT] serializerManager.wrapStream decompression
 *                                  at storage/S3ShuffleReader.scala:98-110, over the byte
 *                                  range storage/S3ShuffleBlockIterator.scala:37-42 /
 *                                  storage/S3ShuffleBlockStream.scala:36-40 define.
 *   s3s_decompressed_size          (no reference counterpart: sizing helper for the shim;
 *                                  the JVM path grows its buffers while streaming)
 *   s3s_max_compressed_size        (sizing helper; LZ4BlockOutputStream allocates
 *                                  HEADER_LENGTH + maxCompressedLength(blockSize) per block)
 *   s3s_create / s3s_destroy       shuffle/helper/S3ShuffleDispatcher.scala:240-255 (the
 *                                  process-wide lazily built state); the device is chosen
 *                                  by the caller as mapId % nGPU, mirroring
 *                                  mapId % folderPrefixes at S3ShuffleDispatcher.scala:142.
 *
 * Byte formats produced / consumed (bit-exact with the JVM path):
 *   LZ4    one lz4-java LZ4BlockOutputStream per non-empty partition: frames of
 *          "LZ4Block" | token | compressedLen LE | originalLen LE | xxh32&0x0FFFFFFF LE |
 *          payload (LZ4_compress_default of a block_size chunk, or the raw chunk when that
 *          is not smaller), then a 21-byte end frame.  Empty partition = 0 bytes.
 *   SNAPPY one snappy-java SnappyOutputStream per non-empty partition: 16-byte header,
 *          then compressedLen BE | raw snappy per block_size chunk.
 *   NONE   the partition bytes unchanged (spark.shuffle.compress=false).
 *   out_index     cumulative offsets [0, L0, L0+L1, ...] (host-endian int64; the caller
 *                 serialises big-endian exactly like S3ShuffleHelper.writeArrayAsBlock).
 *   out_checksums java.util.zip.Adler32 / CRC32 getValue() over each partition's
 *                 COMPRESSED bytes, i.e. over data[index[p], index[p+1]).
 *
 * Threading: an s3s_ctx is NOT thread-safe; use one per task thread.  Calls on distinct
 * contexts are fully concurrent (each owns its HIP stream and device workspace).
 * Ownership: the caller owns every buffer passed in; the library owns only the ctx.
 * Errors: 0 on success, negative S3S_E_* otherwise; never aborts; s3s_last_error() gives
 * a message.  There is NO CPU fallback: without a usable HIP device s3s_create() fails.
 */
#ifndef S3SHUFFLE_CODEC_H
#define S3SHUFFLE_CODEC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define S3S_ABI_VERSION 8 /* 2: + segments entry points, page-locked staging, tuning options 6, 7;
                             3: + s3s_compress_map_outputs_batch_device;
                             4: + s3s_decompress_ranges_batch_device; decode variants {3, 4}, LZ4 parses {1, 9, 10};
                             5: + s3s_compress_map_outputs_batch / s3s_decompress_ranges_batch (host buffers),
                                S3S_CODEC_ZSTD on the reduce side;
                             6: S3S_STATUS_NOT_RUN in the per-entry status of the batched calls (a call-level failure is told
                                apart from an entry's own verdict);
                             7: + S3S_CODEC_LZF on the reduce side; LZ4Block frames above 32 KiB through the batch decoder;
                             8: + S3S_CHECKSUM_CRC32C */

/* spark.io.compression.codec (only when spark.shuffle.compress=true) */
enum { S3S_CODEC_NONE = 0, S3S_CODEC_LZ4 = 1, S3S_CODEC_SNAPPY = 2,
       S3S_CODEC_LZF = 4, /* reduce side only (ABI 7): LZFCompressionCodec streams (compress-lzf chunks 'Z' 'V' type | len ...
                             around liblzf blocks, up to 65 535 bytes each) through the batch decoder; compression stays on
                             the JVM - compress-lzf's output is not a function of the partition's bytes */
       S3S_CODEC_ZSTD = 3 /* reduce side only (s3s_decompress_range*, s3s_decompressed_size): Zstandard frames as
                             ZStdCompressionCodec / zstd-jni write them, one per non-empty partition; the compress
                             entry points answer S3S_E_UNSUPPORTED (the codec stays on the JVM) */ };
/* spark.shuffle.checksum.algorithm (NONE when spark.shuffle.checksum.enabled=false) */
enum { S3S_CHECKSUM_NONE = 0, S3S_CHECKSUM_ADLER32 = 1, S3S_CHECKSUM_CRC32 = 2,
       S3S_CHECKSUM_CRC32C = 3 /* ABI 8: java.util.zip.CRC32C (Castagnoli), Spark 4's third spark.shuffle.checksum.algorithm */ };

enum {
  S3S_OK = 0,
  S3S_E_INVALID = -1,     /* bad argument (RuntimeException("Precondition: ...") on the JVM) */
  S3S_E_CAPACITY = -2,    /* dst_capacity too small */
  S3S_E_BAD_FRAME = -3,   /* IOException("Stream is corrupted") */
  S3S_E_CHECKSUM = -4,    /* SparkException("Invalid checksum detected for ...") */
  S3S_E_HIP = -5,         /* HIP runtime / device failure */
  S3S_E_UNSUPPORTED = -6, /* e.g. block size outside the supported range */
  S3S_E_NOMEM = -7,
  S3S_STATUS_NOT_RUN = -100 /* only ever in s3s_map_task.status / s3s_fetch_range.status: every entry of a batch is stamped
                               with it on entry to the call, so that after a CALL-level failure (bad argument, HIP error
                               before or between the tasks) the entries the library never finished are told apart from the
                               ones that have their own verdict; the call's return code applies to exactly these */
};

/* option keys for s3s_set_option / s3s_get_option */
enum {
  S3S_OPT_LZ4_BLOCK_SIZE = 1,    /* spark.io.compression.lz4.blockSize, default 32768;
                                    map side: 64..65536 (liblz4's 16-bit-table parse; larger blocks are its 32-bit-table
                                    parse: S3S_E_UNSUPPORTED, the JVM codec writes them); the reduce side decodes frames
                                    of ANY block size whatever this option says */
  S3S_OPT_SNAPPY_BLOCK_SIZE = 2, /* spark.io.compression.snappy.blockSize, default 32768;
                                    supported 1024..32768 (snappy-java raises smaller values
                                    to 1024) */
  S3S_OPT_PROFILE = 3,           /* 1: record per-stage HIP-event timings (s3s_stage_ms) */
  S3S_OPT_LZ4_VARIANT = 4,       /* tuning, identical output: 1 = general batch only (64 probes of the greedy
                                    parse per step), 10 (default) = lean exact 64-byte windows in front of
                                    it (one candidate gather per window, several sequences per round trip),
                                    9 = auto: the context times 1 and 10 on its first large map outputs
                                    (1, 10, 1, 10), keeps the faster and re-measures the other every 32nd
                                    call.  Other values are refused. */
  S3S_OPT_LZ4_VARIANT_USED = 7,  /* read-only: the parse (1 or 10) the last LZ4 compress call ran */
  S3S_OPT_SNAPPY_VARIANT = 6,    /* tuning, identical output: 0 = general batch only, 1 (default) = exact
                                    64-byte windows in front of it (several copies per round trip) */
  S3S_OPT_LZ4_DECODE_VARIANT = 5 /* tuning, identical output, LZ4 and Snappy: 4 (default) = batch decoder
                                    (one sequence per lane, dependency rounds, sliding LDS output window),
                                    3 = ring decoder (one sequence per step, parse on the vector ALU) */
};

/* stages reported by s3s_stage_ms (valid after a call made with S3S_OPT_PROFILE=1) */
enum {
  S3S_STAGE_TOTAL = 0,      /* first kernel start -> last kernel end of the last call */
  S3S_STAGE_CODEC = 1,      /* the block compress (or decompress) kernel ALONE — the dominant one */
  S3S_STAGE_ASSEMBLE = 2,   /* offset scan + frame gather */
  S3S_STAGE_CHECKSUM = 3,   /* per-partition Adler32 / CRC32 */
  S3S_STAGE_DISCOVER = 4,   /* reduce side: frame discovery */
  S3S_STAGE_HASH = 5,       /* map side: per-chunk xxHash32 pre-pass (LZ4Block frame check) */
  S3S_STAGE_COUNT = 6
};

typedef struct s3s_ctx s3s_ctx;

/* ---- lifecycle ------------------------------------------------------------------------ */
const char* s3s_version(void);
int s3s_abi_version(void);
/* Number of visible HIP devices (0 when there is none; never fails). */
int s3s_device_count(void);
/* Creates a context bound to `device_ordinal`; `scratch_bytes` pre-sizes the device
 * workspace (0 = grow on demand).  Returns NULL on failure (s3s_last_error(NULL) tells why). */
s3s_ctx* s3s_create(int device_ordinal, int64_t scratch_bytes);
void s3s_destroy(s3s_ctx* ctx);
const char* s3s_last_error(const s3s_ctx* ctx);
int s3s_set_option(s3s_ctx* ctx, int key, int64_t value);
int64_t s3s_get_option(const s3s_ctx* ctx, int key);
/* The HIP stream (hipStream_t) the context launches on, for callers that time with events. */
void* s3s_stream(const s3s_ctx* ctx);
double s3s_stage_ms(const s3s_ctx* ctx, int stage);

/* ---- sizing ---------------------------------------------------------------------------- */
/* Upper bound of the .data image for these partition ranges. ctx may be NULL (defaults). */
int64_t s3s_max_compressed_size(const s3s_ctx* ctx, int codec, const int64_t* src_offsets,
                                int32_t num_partitions);

/* ---- map side: compress + checksum one map task's output ------------------------------- */
/* Partition p's serialized (uncompressed) bytes are src[src_offsets[p], src_offsets[p+1]).
 * Writes the exact .data byte image into dst, out_index[num_partitions+1],
 * out_checksums[num_partitions] (may be NULL iff checksum_algo == NONE), *out_total.
 * Host-buffer variant: src/dst are host memory (H2D/D2H inside the call).               */
int s3s_compress_map_output(s3s_ctx* ctx, int codec, int checksum_algo, const uint8_t* src,
                            const int64_t* src_offsets, int32_t num_partitions, uint8_t* dst,
                            int64_t dst_capacity, int64_t* out_index, int64_t* out_checksums,
                            int64_t* out_total);
/* Device-buffer variant: src/dst are device memory on ctx's device (offsets/index/checksums
 * stay host arrays).  This is the form the roofline metric is measured on. */
int s3s_compress_map_output_device(s3s_ctx* ctx, int codec, int checksum_algo,
                                   const uint8_t* d_src, const int64_t* src_offsets,
                                   int32_t num_partitions, uint8_t* d_dst, int64_t dst_capacity,
                                   int64_t* out_index, int64_t* out_checksums,
                                   int64_t* out_total);

/* Batched form of s3s_compress_map_output_device: several map tasks of one executor in ONE call.
 * No reference counterpart as an interface — on the JVM every task thread drives its own
 * LZ4BlockOutputStream (shuffle/S3ShuffleMapOutputWriter.scala:168-202); the shim collects the
 * spill buffers of the tasks that commit together (shuffle/S3ShuffleMapOutputWriter.scala:91-118)
 * and hands them over at once, so that the chip sees enough 32 KiB chunks to fill its 2 560
 * resident wavefronts (an 8 MiB map output — the reference's default write buffer,
 * shuffle/helper/S3ShuffleDispatcher.scala:55 — is 256 chunks).  All chunks of all tasks go
 * through one codec launch; offsets, gather and checksums run per task behind it; one stream
 * synchronisation for the batch.  Every task gets exactly the bytes, index and checksums
 * s3s_compress_map_output_device would have produced for it alone.  One codec stream per
 * non-empty partition (no multi-spill pieces in the batched form).
 * Returns S3S_OK, or the first failing task's error; each task's own result is in .status
 * (S3S_STATUS_NOT_RUN = the call failed before this task had a result: the return code is its error). */
typedef struct s3s_map_task {
  const uint8_t* d_src;         /* in: device memory holding this task's partitions */
  const int64_t* src_offsets;   /* in: host array [num_partitions + 1], offsets into d_src */
  int32_t num_partitions;       /* in */
  uint8_t* d_dst;               /* in: device buffer for the .data image */
  int64_t dst_capacity;         /* in */
  int64_t* out_index;           /* out: host array [num_partitions + 1] */
  int64_t* out_checksums;       /* out: host array [num_partitions] (may be NULL iff checksum NONE) */
  int64_t out_total;            /* out: bytes of the .data image */
  int32_t status;               /* out: S3S_OK / S3S_E_CAPACITY for this task; S3S_STATUS_NOT_RUN: see the enum */
} s3s_map_task;
int s3s_compress_map_outputs_batch_device(s3s_ctx* ctx, int codec, int checksum_algo,
                                          s3s_map_task* tasks, int32_t n_tasks);

/* The same batch for HOST buffers — the form a Spark executor reaches through the JNI shim (the reference's
 * call sites hold heap / direct buffers: shuffle/S3ShuffleMapOutputWriter.scala:91-118, 168-202).  In this
 * variant `d_src` / `d_dst` of every task are HOST addresses (page-locked memory from s3s_host_alloc moves by
 * plain DMA; pageable memory works through the runtime's bounce buffers).  The tasks run as a pipeline over
 * groups of ~64 MiB: the upload of group g+1 and the download of group g-1 are in flight while group g is
 * compressed, so one call is bound by the slower of PCIe host->device (uncompressed bytes) and the codec.
 * Results per task are exactly those of s3s_compress_map_output.  A single task takes that entry point's
 * own chunked upload overlap. */
int s3s_compress_map_outputs_batch(s3s_ctx* ctx, int codec, int checksum_algo, s3s_map_task* tasks,
                                   int32_t n_tasks);

/* ---- checksum only ---------------------------------------------------------------------- */
/* out[i] = checksum(data[offsets[i], offsets[i+1])) for i in [0, n). */
int s3s_checksum_ranges(s3s_ctx* ctx, int checksum_algo, const uint8_t* data,
                        const int64_t* offsets, int32_t n, int64_t* out);
int s3s_checksum_ranges_device(s3s_ctx* ctx, int checksum_algo, const uint8_t* d_data,
                               const int64_t* offsets, int32_t n, int64_t* out);

/* ---- reduce side: verify + decompress one fetched block range --------------------------- */
/* comp[0, comp_len) holds partitions r0..r1-1 of one map output (a ShuffleBlockId or a
 * ShuffleBlockBatchId range); part_offsets[nparts+1] are the .index entries relative to the
 * range start (part_offsets[0] == 0, part_offsets[nparts] == comp_len).  When
 * checksum_algo != NONE every partition is validated against ref_checksums first
 * (S3S_E_CHECKSUM, *out_bad_partition = first failing one); then the concatenated codec
 * streams are decoded into dst (frame hashes verified; S3S_E_BAD_FRAME on corruption). */
int s3s_decompress_range(s3s_ctx* ctx, int codec, int checksum_algo, const uint8_t* comp,
                         int64_t comp_len, const int64_t* part_offsets,
                         const int64_t* ref_checksums, int32_t nparts, uint8_t* dst,
                         int64_t dst_capacity, int64_t* out_len, int32_t* out_bad_partition);
int s3s_decompress_range_device(s3s_ctx* ctx, int codec, int checksum_algo,
                                const uint8_t* d_comp, int64_t comp_len,
                                const int64_t* part_offsets, const int64_t* ref_checksums,
                                int32_t nparts, uint8_t* d_dst, int64_t dst_capacity,
                                int64_t* out_len, int32_t* out_bad_partition);
/* Batched reduce side: many fetched ranges (the blocks a reduce task's prefetcher has staged — reference
 * buffers of 8 MiB..128 MiB, S3ShuffleDispatcher.scala:55-57, S3BufferedPrefetchIterator.scala:102-153) in ONE
 * call: checksums and frame discovery of all ranges are queued back to back, ONE decode launch covers the
 * frames of every range, and the host waits three times per batch instead of three times per range.  Every
 * range gets exactly what s3s_decompress_range_device would have reported for it alone (status, out_len,
 * bad_partition).  Returns S3S_OK or the first failing range's error. */
typedef struct s3s_fetch_range {
  const uint8_t* d_comp;         /* in: device memory, comp_len bytes */
  int64_t comp_len;              /* in */
  const int64_t* part_offsets;   /* in: host array [num_partitions + 1], relative to the range start */
  const int64_t* ref_checksums;  /* in: host array [num_partitions] (may be NULL iff checksum NONE) */
  int32_t num_partitions;        /* in */
  uint8_t* d_dst;                /* in: device buffer for the decoded bytes */
  int64_t dst_capacity;          /* in */
  int64_t out_len;               /* out: decoded bytes */
  int32_t bad_partition;         /* out: first partition with a wrong checksum, or -1 */
  int32_t status;                /* out: S3S_OK / S3S_E_CHECKSUM / S3S_E_BAD_FRAME / S3S_E_CAPACITY / ...; S3S_STATUS_NOT_RUN: see the enum */
} s3s_fetch_range;
int s3s_decompress_ranges_batch_device(s3s_ctx* ctx, int codec, int checksum_algo,
                                       s3s_fetch_range* ranges, int32_t n_ranges);

/* Host-buffer form (storage/S3ShuffleReader.scala:98-110 holds the prefetcher's heap buffers): `d_comp` /
 * `d_dst` of every range are HOST addresses; upload of the next group, decode of this one and download of the
 * previous one overlap (groups of ~128 MiB decoded), so one call is bound by the slower of PCIe
 * device->host (decoded bytes) and the decoder.  Results per range are those of s3s_decompress_range. */
int s3s_decompress_ranges_batch(s3s_ctx* ctx, int codec, int checksum_algo, s3s_fetch_range* ranges,
                                int32_t n_ranges);

/* Multi-spill map tasks.  When a map task spilled N times, Spark's merge hands every partition to the
 * partition writer as the concatenation of N independently written pieces, and on the JVM each piece is
 * one COMPLETE codec stream (LZ4Block end frame / Snappy stream header included) — see the writers that
 * feed shuffle/S3ShuffleMapOutputWriter.scala:67-83 and the pre-merged spill file of
 * shuffle/S3SingleSpillShuffleMapOutputWriter.scala:24-64.  To keep the `.data` object byte-identical
 * the shim passes the piece boundaries: seg_offsets[0..n_segs] delimit the pieces in src (ascending,
 * contiguous), part_first_seg[0..n] (part_first_seg[0] = 0, part_first_seg[n] = n_segs) assigns them to
 * the n partitions.  Every non-empty piece becomes one stream; index and checksums stay per partition.
 * With one piece per partition this is exactly s3s_compress_map_output.                               */
int64_t s3s_max_compressed_size_segments(const s3s_ctx* ctx, int codec, const int64_t* seg_offsets,
                                         int32_t n_segs);
int s3s_compress_map_output_segments(s3s_ctx* ctx, int codec, int checksum_algo, const uint8_t* src,
                                     const int64_t* seg_offsets, int32_t n_segs,
                                     const int32_t* part_first_seg, int32_t n, uint8_t* dst,
                                     int64_t dst_capacity, int64_t* out_index, int64_t* out_checksums,
                                     int64_t* out_total);
int s3s_compress_map_output_segments_device(s3s_ctx* ctx, int codec, int checksum_algo,
                                            const uint8_t* d_src, const int64_t* seg_offsets,
                                            int32_t n_segs, const int32_t* part_first_seg, int32_t n,
                                            uint8_t* d_dst, int64_t dst_capacity, int64_t* out_index,
                                            int64_t* out_checksums, int64_t* out_total);

/* Page-locked host staging memory for the host-buffer entry points (no reference counterpart:
 * it replaces the heap byte[] of storage/S3BufferedInputStreamAdaptor.scala:13-19 — one
 * BufferedInputStream of min(maxBufferSizeTask, block length) bytes per prefetched block — and of
 * the BufferedOutputStream in shuffle/S3ShuffleMapOutputWriter.scala:43-49).  The JVM shim wraps
 * it with NewDirectByteBuffer and lets the S3 client / the serializer fill it in place; the
 * library then moves it with plain DMA (~55 GB/s per direction on PCIe Gen5 x16) instead of the
 * bounce-buffer copy that pageable memory costs.  Usable from any thread and with any context;
 * returns NULL when bytes <= 0 or the allocation fails.                                        */
void* s3s_host_alloc(int64_t bytes);
void s3s_host_free(void* p);

/* Decoded size of the codec streams in comp[0, comp_len) (host memory). */
int s3s_decompressed_size(s3s_ctx* ctx, int codec, const uint8_t* comp, int64_t comp_len,
                          int64_t* out_len);

#ifdef __cplusplus
}
#endif
#endif
