// s3shuffle_host.h — C++ host-side mirror of the reference plugin's data-plane interface for the
// ONE hot path this repo replaces, built on the C-ABI of
// include/s3shuffle_codec.h.  The reference is Scala on the JVM and there is no JVM toolchain in
// this image, so the classes below restate — same names, argument meaning and error behaviour —
// what a Scala shim inside the unchanged plugin would do:
//
//   S3ShuffleDispatcher      shuffle/helper/S3ShuffleDispatcher.scala:39-70 (config), :120-144
//                            (getPath), :146-172 (listShuffleIndices), :190-198/:235-237 (open /
//                            create), :174-183 (removeShuffle).  Storage here is the local file
//                            system only (rootDir = "file:///..." or a plain path): object-store
//                            I/O is out of scope.
//   S3ShuffleHelper          shuffle/helper/S3ShuffleHelper.scala:44-59 (write index / checksum as
//                            big-endian longs), :67-92,105-121 (read them back, length % 8 check)
//   S3ShuffleMapOutputWriter shuffle/S3ShuffleMapOutputWriter.scala:27-244 — getPartitionWriter
//                            (strictly increasing ids), partition streams, commitAllPartitions
//                            (.data, then .index iff sum > 0 || alwaysCreateIndex, then .checksum
//                            iff checksumEnabled), abort.  With the GPU path on, partition
//                            writers receive UNCOMPRESSED serialized bytes and commit runs
//                            s3s_compress_map_output on device mapId % nGPU.
//   S3SingleSpillShuffleMapOutputWriter  shuffle/S3SingleSpillShuffleMapOutputWriter.scala:18-64
//   S3ShuffleReader          storage/S3ShuffleReader.scala:77-158 + S3ShuffleBlockIterator.scala:26-56
//                            + S3ShuffleBlockStream.scala:36-40,73-93 + S3ChecksumValidationStream
//                            .scala:54-86 — block (or batch) -> byte range -> verify + decode
//                            through s3s_decompress_range.
//   S3BufferedPrefetchIterator storage/S3BufferedPrefetchIterator.scala:16-200 + S3BufferedInputStreamAdaptor
//                            .scala:7-60 — the staging step in front of the decode
//                            kernel: up to maxConcurrencyTask fetch threads fill per-block buffers under the
//                            maxBufferSizeTask budget, the consumer takes blocks as they complete and gives the
//                            memory back when it closes the stream.  Here the buffers are PAGE-LOCKED
//                            (s3s_host_alloc, pooled), fetch -> H2D -> verify+decode -> D2H of different blocks
//                            overlap on separate contexts (HIP streams), and what the consumer receives is the
//                            DECODED block in pinned memory (s3shuffle_prefetch.cpp).
#pragma once
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/s3shuffle_codec.h"

namespace s3shuffle {

struct SparkException : std::runtime_error {
  using std::runtime_error::runtime_error;
};
struct IOException : std::runtime_error {
  using std::runtime_error::runtime_error;
};

// ---- block ids (org.apache.spark.storage.BlockId names) ---------------------------------------
struct BlockId {
  enum Kind { SHUFFLE, SHUFFLE_BATCH, SHUFFLE_DATA, SHUFFLE_INDEX, SHUFFLE_CHECKSUM } kind;
  int shuffleId;
  int64_t mapId;
  int reduceId;     // startReduceId for a batch
  int endReduceId;  // batch only
  std::string name() const;
  static BlockId ShuffleBlockId(int s, int64_t m, int r) { return {SHUFFLE, s, m, r, r + 1}; }
  static BlockId ShuffleBlockBatchId(int s, int64_t m, int r0, int r1) { return {SHUFFLE_BATCH, s, m, r0, r1}; }
  static BlockId ShuffleDataBlockId(int s, int64_t m) { return {SHUFFLE_DATA, s, m, 0, 1}; }
  static BlockId ShuffleIndexBlockId(int s, int64_t m) { return {SHUFFLE_INDEX, s, m, 0, 1}; }
  static BlockId ShuffleChecksumBlockId(int s, int64_t m) { return {SHUFFLE_CHECKSUM, s, m, 0, 1}; }
};

// ---- configuration: the spark.shuffle.s3.* / spark.shuffle.checksum.* / spark.io.compression.* keys
struct Conf {
  std::string rootDir = "sparkS3shuffle/";   // spark.shuffle.s3.rootDir
  std::string appId = "app";                 // spark.app.id
  int folderPrefixes = 10;                   // spark.shuffle.s3.folderPrefixes
  bool alwaysCreateIndex = false;            // spark.shuffle.s3.alwaysCreateIndex
  bool useSparkShuffleFetch = false;         // spark.shuffle.s3.useSparkShuffleFetch: the fallback-storage object layout
  bool checksumEnabled = true;               // spark.shuffle.checksum.enabled
  std::string checksumAlgorithm = "ADLER32";  // spark.shuffle.checksum.algorithm
  bool compress = true;                      // spark.shuffle.compress
  std::string codec = "lz4";                 // spark.io.compression.codec
  int blockSize = 32768;                     // spark.io.compression.{lz4,snappy}.blockSize
  int numGpus = 0;                           // spark.shuffle.s3.gpu.devices (0 = all visible)
  int64_t maxBufferSizeTask = 128ll << 20;   // spark.shuffle.s3.maxBufferSizeTask: prefetch budget (compressed bytes)
  int maxConcurrencyTask = 10;               // spark.shuffle.s3.maxConcurrencyTask: fetch threads per task
  int gpuDecodeThreads = 2;                  // spark.shuffle.s3.gpu.decodeThreads: contexts (streams) decoding per task
  int64_t gpuMaxDecodedBufferSizeTask = 512ll << 20;  // spark.shuffle.s3.gpu.maxDecodedBufferSizeTask (pinned, decoded)
  // spark.shuffle.s3.gpu.fetchThreadPredictor: let the reference's ThreadPredictor choose how many of the
  // maxConcurrencyTask fetch threads run (default off: all of them do — against a local / tmpfs store the consumer
  // never waits and the predictor would stay at one thread)
  bool fetchThreadPredictor = false;
};

// The fetch-thread tuner of the reference's prefetcher (storage/S3BufferedPrefetchIterator.scala:32-69), restated: it is
// fed how long the consumer had to wait for a block; every 20 measurements (+ one per running thread) it records their sum
// for the current number of threads and moves one thread down or up if that neighbour's recorded sum is smaller.
class ThreadPredictor {
 public:
  explicit ThreadPredictor(int maxThreads);
  int addMeasurementAndPredict(int64_t latencyNs);  // latencyNs < 0: no measurement, just the current answer
  int current() const;

 private:
  int predict();
  mutable std::mutex mu_;
  int currentThreads_ = 1;
  std::vector<int64_t> latencies_;      // [0] and [maxThreads + 1] are walls (Long.MaxValue)
  int numMeasurements_ = 0;
  int64_t measurementsNs_[20] = {0};
};

class S3ShuffleDispatcher {
 public:
  explicit S3ShuffleDispatcher(const Conf& conf);
  const Conf& conf() const { return conf_; }
  std::string getPath(const BlockId& id) const;  // local path (scheme stripped)
  std::vector<BlockId> listShuffleIndices(int shuffleId) const;
  void createBlock(const BlockId& id, const void* data, size_t n) const;
  std::vector<uint8_t> readBlock(const BlockId& id) const;
  std::vector<uint8_t> readBlockRange(const BlockId& id, int64_t pos, int64_t n) const;
  void readBlockRangeInto(const BlockId& id, int64_t pos, int64_t n, uint8_t* dst) const;  // positioned readFully
  int64_t blockLength(const BlockId& id) const;  // -1 if missing
  void removeShuffle(int shuffleId) const;
  void removeRoot() const;
  int codecId() const;      // S3S_CODEC_*
  int checksumId() const;   // S3S_CHECKSUM_*; throws UnsupportedOperationException-like for unknown names
  int deviceForMap(int64_t mapId) const;
  void setUseSparkShuffleFetch(bool on) { conf_.useSparkShuffleFetch = on; }
  void setFetchThreadPredictor(bool on) { conf_.fetchThreadPredictor = on; }
  void setPrefetch(int64_t maxBufferSizeTask, int maxConcurrencyTask, int gpuDecodeThreads, int64_t gpuMaxDecodedBufferSizeTask) {
    if (maxBufferSizeTask > 0) conf_.maxBufferSizeTask = maxBufferSizeTask;
    if (maxConcurrencyTask > 0) conf_.maxConcurrencyTask = maxConcurrencyTask;
    if (gpuDecodeThreads > 0) conf_.gpuDecodeThreads = gpuDecodeThreads;
    if (gpuMaxDecodedBufferSizeTask > 0) conf_.gpuMaxDecodedBufferSizeTask = gpuMaxDecodedBufferSizeTask;
  }

 private:
  Conf conf_;
  std::string root_;  // local directory, ends with '/'
  int ngpu_;
};

namespace S3ShuffleHelper {
void writePartitionLengths(const S3ShuffleDispatcher& d, int shuffleId, int64_t mapId,
                           const std::vector<int64_t>& partitionLengths);
void writeChecksum(const S3ShuffleDispatcher& d, int shuffleId, int64_t mapId, const std::vector<int64_t>& checksums);
void writeArrayAsBlock(const S3ShuffleDispatcher& d, const BlockId& id, const std::vector<int64_t>& array);
std::vector<int64_t> getPartitionLengths(const S3ShuffleDispatcher& d, int shuffleId, int64_t mapId);
std::vector<int64_t> getChecksums(const S3ShuffleDispatcher& d, int shuffleId, int64_t mapId);
std::vector<int64_t> readBlockAsArray(const S3ShuffleDispatcher& d, const BlockId& id);
}  // namespace S3ShuffleHelper

// One per map task, used by exactly one task thread.
class S3ShuffleMapOutputWriter {
 public:
  S3ShuffleMapOutputWriter(const S3ShuffleDispatcher& d, int shuffleId, int64_t mapId, int numPartitions);
  ~S3ShuffleMapOutputWriter();
  // ShuffleMapOutputWriter.getPartitionWriter: ids must be strictly increasing
  void getPartitionWriter(int reducePartitionId);
  // ShufflePartitionWriter.openStream().write(b, off, len) on the current partition writer
  void write(const void* bytes, size_t len);
  int64_t getNumBytesWritten() const;  // of the current partition (uncompressed: what the task wrote)
  // closes the current partition stream; commitAllPartitions closes the last one implicitly
  void closePartition();
  // multi-spill merge: the bytes written to the current partition so far are one spill's piece, the next
  // write starts another.  On the JVM every piece is a complete codec stream (the merge copies the spill
  // files' partition ranges verbatim), so a merge that knows its spill boundaries calls this between
  // pieces and the stored object stays byte-identical (s3s_compress_map_output_segments).  Optional: without
  // it a partition is one stream, which every reader decodes to the same records.
  void markSegment();
  // returns partitionLengths (compressed bytes per partition), like MapOutputCommitMessage.of(...)
  std::vector<int64_t> commitAllPartitions();
  void abort();

 private:
  const S3ShuffleDispatcher& d_;
  int shuffleId_;
  int64_t mapId_;
  int numPartitions_;
  int lastPartitionWriterId_ = -1;
  bool streamClosed_ = true, committed_ = false;
  uint8_t* stage_ = nullptr;  // page-locked, from PinnedPool::process()
  int64_t stageCap_ = 0, stageLen_ = 0;
  std::vector<int64_t> srcOffsets_;  // numPartitions + 1
  std::vector<std::vector<int64_t>> cuts_;  // per partition: piece boundaries inside it (absolute staging offsets)
  s3s_ctx* ctx_ = nullptr;
};

// shuffle/S3SingleSpillShuffleMapOutputWriter.scala:18-64: the map task produced exactly one spill file that
// already has the final partition order; the reference moves / copies it into the store as the .data object
// and writes .checksum (iff enabled) and then .index from the arrays Spark hands over.  With the GPU path on,
// Spark-level compression is off, so the spill holds UNCOMPRESSED partitions: the file is read into
// page-locked staging, compressed + checksummed on the GPU (offsets = running sum of partitionLengths) and
// stored; .index / .checksum describe the COMPRESSED object (Spark's own arrays describe the spill and are
// only used for the offsets).  Like the reference it always writes the index and consumes the spill file.
class S3SingleSpillShuffleMapOutputWriter {
 public:
  S3SingleSpillShuffleMapOutputWriter(const S3ShuffleDispatcher& d, int shuffleId, int64_t mapId)
      : d_(d), shuffleId_(shuffleId), mapId_(mapId) {}
  // returns the compressed partition lengths (what .index holds)
  std::vector<int64_t> transferMapSpillFile(const std::string& mapSpillFile, const std::vector<int64_t>& partitionLengths,
                                            const std::vector<int64_t>& checksums);

 private:
  const S3ShuffleDispatcher& d_;
  int shuffleId_;
  int64_t mapId_;
};

struct FetchedBlock {
  BlockId id;
  std::vector<uint8_t> bytes;  // decoded (decompressed) serialized records of the block
};

// ---- page-locked staging memory, pooled (hipHostMalloc costs milliseconds: never per block) ------------
class PinnedPool {
 public:
  // blocking: acquire() waits for the budget (prefetch pipeline); otherwise the budget only bounds the cache
  explicit PinnedPool(int64_t budgetBytes, bool blocking = true);
  ~PinnedPool();
  PinnedPool(const PinnedPool&) = delete;
  // blocks until `bytes` fit under the budget; a request larger than the whole budget is served once
  // nothing else is in use (the reference clamps its buffer to maxBufferSize instead and streams the
  // rest, S3BufferedPrefetchIterator.scala:127 — a GPU block has to be resident as a whole)
  uint8_t* acquire(int64_t bytes);
  void release(uint8_t* p);
  void cancel();  // waiting and future acquire() calls throw IOException (pipeline teardown)
  int64_t inUse() const;
  int64_t highWater() const;
  static PinnedPool& process();  // map-side staging: unbounded, non-blocking view of the process-wide cache

 private:
  mutable std::mutex mu_;
  std::condition_variable cv_;
  int64_t budget_, inUse_ = 0, high_ = 0;
  bool cancelled_ = false, blocking_ = true;
  std::map<uint8_t*, int64_t> used_, reserved_;  // buffer -> its real capacity / what it counts for
};

// codec contexts (stream + device workspace) recycled across tasks, keyed by device and codec settings;
// acquire never blocks (a new context is created when none is idle)
s3s_ctx* acquireContext(const S3ShuffleDispatcher& d, int device);
void releaseContext(const S3ShuffleDispatcher& d, int device, s3s_ctx* ctx);
void releaseContextCache();  // destroys the idle contexts (executor shutdown)
void releasePinnedCache();   // frees the idle page-locked buffers

// one S3ShuffleBlockStream: the byte range of a block (or batch) inside a map output's .data object
// (S3ShuffleBlockIterator.scala:37-42) plus what the decode needs
struct BlockRequest {
  BlockId id;
  BlockId dataBlock;
  int64_t startPosition = 0, maxBytes = 0;
  std::vector<int64_t> rel;   // .index slice relative to startPosition, r1 - r0 + 1 entries
  std::vector<int64_t> sums;  // reference checksums of the range (empty = validation off)
  int device = 0;
};

struct PrefetchedBlock {
  BlockId id;
  const uint8_t* data = nullptr;  // decoded bytes in page-locked memory, valid until release()
  int64_t size = 0;
};

class S3BufferedPrefetchIterator {
 public:
  S3BufferedPrefetchIterator(const S3ShuffleDispatcher& d, std::vector<BlockRequest> requests);
  ~S3BufferedPrefetchIterator();
  bool hasNext();
  // the next completed block, in completion order; throws what the block's fetch / verify / decode
  // raised (SparkException "Invalid checksum detected for ...", IOException "Stream is corrupted")
  PrefetchedBlock next();
  void release(PrefetchedBlock& b);  // ≙ closing the stream: returns the memory (onCloseStream, :96-100)
  struct Stats {
    int64_t blocks = 0, compressedBytes = 0, decodedBytes = 0;
    double secondsWaiting = 0;  // consumer blocked in next()
    int fetchThreads = 0;       // fetch threads allowed to run right now (the predictor's answer, or all of them)
    int64_t compHighWater = 0, decodedHighWater = 0;
  };
  Stats stats() const;

 private:
  struct Fetched {
    size_t req;
    uint8_t* comp;
  };
  struct Done {
    size_t req;
    uint8_t* out;
    int64_t len;
    int errKind;  // 0 ok, 1 SparkException, 2 IOException
    std::string err;
  };
  void fetchLoop(int id);  // id 1..nFetch: with the predictor on, thread id runs while id <= the predicted count
  void decodeLoop();
  const S3ShuffleDispatcher& d_;
  std::vector<BlockRequest> reqs_;
  PinnedPool comp_, dec_;
  mutable std::mutex mu_;
  std::condition_variable cvFetched_, cvDone_;
  std::deque<Fetched> fetched_;
  std::deque<Done> done_;
  size_t nextReq_ = 0, fetchersLeft_ = 0, delivered_ = 0;
  bool stop_ = false;
  std::vector<std::thread> threads_;
  Stats stats_;
  ThreadPredictor predictor_;
  int desiredFetchers_ = 1;  // (guarded by mu_)
  std::condition_variable cvPark_;
};

class S3ShuffleReader {
 public:
  // reads partitions [startPartition, endPartition) of every map output of the shuffle; with
  // doBatchFetch the range is one ShuffleBlockBatchId per map output, else one ShuffleBlockId each
  S3ShuffleReader(const S3ShuffleDispatcher& d, int shuffleId, int startPartition, int endPartition,
                  bool doBatchFetch);
  ~S3ShuffleReader();
  // all blocks, decoded, ordered by (mapId, reduceId); runs on the prefetch pipeline
  std::vector<FetchedBlock> read();
  // the same blocks fetched and decoded one after the other on one context with pageable buffers
  // (the round-1 path; kept as the baseline the pipeline is measured against)
  std::vector<FetchedBlock> readSequential();
  // the S3ShuffleBlockStream list of this reader (computeShuffleBlocks + S3ShuffleBlockIterator)
  std::vector<BlockRequest> blockRequests() const;

 private:
  const S3ShuffleDispatcher& d_;
  int shuffleId_, start_, end_;
  bool batch_;
  s3s_ctx* ctx_ = nullptr;
  int ctxDevice_ = 0;
};

}  // namespace s3shuffle
