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
//   S3ShuffleReader          storage/S3ShuffleReader.scala:77-158 + S3ShuffleBlockIterator.scala:26-56
//                            + S3ShuffleBlockStream.scala:36-40,73-93 + S3ChecksumValidationStream
//                            .scala:54-86 — block (or batch) -> byte range -> verify + decode
//                            through s3s_decompress_range.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
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
  bool checksumEnabled = true;               // spark.shuffle.checksum.enabled
  std::string checksumAlgorithm = "ADLER32";  // spark.shuffle.checksum.algorithm
  bool compress = true;                      // spark.shuffle.compress
  std::string codec = "lz4";                 // spark.io.compression.codec
  int blockSize = 32768;                     // spark.io.compression.{lz4,snappy}.blockSize
  int numGpus = 0;                           // spark.shuffle.s3.gpu.devices (0 = all visible)
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
  int64_t blockLength(const BlockId& id) const;  // -1 if missing
  void removeShuffle(int shuffleId) const;
  void removeRoot() const;
  int codecId() const;      // S3S_CODEC_*
  int checksumId() const;   // S3S_CHECKSUM_*; throws UnsupportedOperationException-like for unknown names
  int deviceForMap(int64_t mapId) const;

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
  std::vector<uint8_t> staging_;
  std::vector<int64_t> srcOffsets_;  // numPartitions + 1
  s3s_ctx* ctx_ = nullptr;
};

struct FetchedBlock {
  BlockId id;
  std::vector<uint8_t> bytes;  // decoded (decompressed) serialized records of the block
};

class S3ShuffleReader {
 public:
  // reads partitions [startPartition, endPartition) of every map output of the shuffle; with
  // doBatchFetch the range is one ShuffleBlockBatchId per map output, else one ShuffleBlockId each
  S3ShuffleReader(const S3ShuffleDispatcher& d, int shuffleId, int startPartition, int endPartition,
                  bool doBatchFetch);
  ~S3ShuffleReader();
  std::vector<FetchedBlock> read();

 private:
  const S3ShuffleDispatcher& d_;
  int shuffleId_, start_, end_;
  bool batch_;
  s3s_ctx* ctx_ = nullptr;
};

}  // namespace s3shuffle
