// s3shuffle_host.cpp — see s3shuffle_host.h.  Plain C++17 + the codec C-ABI; no torch, no oracle.
#include "s3shuffle_host.h"

#include <dirent.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace s3shuffle {
namespace {

std::atomic<int64_t> g_lastStaged{0};  // bytes the latest committed map task had staged

bool ends_with(const std::string& s, const std::string& suf) {
  return s.size() >= suf.size() && s.compare(s.size() - suf.size(), suf.size(), suf) == 0;
}

void mkdirs(const std::string& path) {
  for (size_t i = 1; i <= path.size(); i++)
    if (i == path.size() || path[i] == '/') {
      const std::string sub = path.substr(0, i);
      if (!sub.empty()) ::mkdir(sub.c_str(), 0777);
    }
}

void rm_rf(const std::string& path) {
  DIR* d = ::opendir(path.c_str());
  if (!d) {
    ::unlink(path.c_str());
    return;
  }
  while (dirent* e = ::readdir(d)) {
    const std::string n = e->d_name;
    if (n == "." || n == "..") continue;
    rm_rf(path + "/" + n);
  }
  ::closedir(d);
  ::rmdir(path.c_str());
}

}  // namespace

std::string BlockId::name() const {
  char buf[96];
  switch (kind) {
    case SHUFFLE: snprintf(buf, sizeof buf, "shuffle_%d_%lld_%d", shuffleId, (long long)mapId, reduceId); break;
    case SHUFFLE_BATCH:
      snprintf(buf, sizeof buf, "shuffle_%d_%lld_%d_%d", shuffleId, (long long)mapId, reduceId, endReduceId);
      break;
    case SHUFFLE_DATA: snprintf(buf, sizeof buf, "shuffle_%d_%lld_%d.data", shuffleId, (long long)mapId, reduceId); break;
    case SHUFFLE_INDEX: snprintf(buf, sizeof buf, "shuffle_%d_%lld_%d.index", shuffleId, (long long)mapId, reduceId); break;
    // the reference writes the checksum block WITHOUT an algorithm suffix (S3ShuffleHelper.scala:49-51)
    case SHUFFLE_CHECKSUM: snprintf(buf, sizeof buf, "shuffle_%d_%lld_%d.checksum", shuffleId, (long long)mapId, reduceId); break;
  }
  return buf;
}

// ---- dispatcher ---------------------------------------------------------------------------------
S3ShuffleDispatcher::S3ShuffleDispatcher(const Conf& conf) : conf_(conf) {
  std::string r = conf.rootDir;
  if (!ends_with(r, "/")) r += "/";  // S3ShuffleDispatcher.scala:51
  conf_.rootDir = r;
  if (r.rfind("file://", 0) == 0) r = r.substr(7);
  else if (r.find("://") != std::string::npos)
    throw IOException("only file:// (or plain path) roots are supported by the host mirror: " + conf.rootDir);
  root_ = r;
  const int visible = s3s_device_count();
  ngpu_ = conf.numGpus > 0 ? std::min(conf.numGpus, std::max(visible, 1)) : std::max(visible, 1);
}

// JavaUtils.nonNegativeHash(String): String.hashCode (s[0]*31^(n-1) + ... in 32-bit wrap-around), abs, MIN_VALUE -> 0
static int32_t java_non_negative_hash(const std::string& s) {
  uint32_t h = 0;
  for (unsigned char c : s) h = 31u * h + c;
  const int32_t v = (int32_t)h;
  return v == INT32_MIN ? 0 : (v < 0 ? -v : v);
}

std::string S3ShuffleDispatcher::getPath(const BlockId& id) const {
  if (conf_.useSparkShuffleFetch) {
    // ${rootDir}${appId}/${shuffleId}/${nonNegativeHash(name)}/${name}: Spark's FallbackStorage layout (:132-141); only the
    // three stored block kinds have a place there
    if (id.kind != BlockId::SHUFFLE_DATA && id.kind != BlockId::SHUFFLE_INDEX && id.kind != BlockId::SHUFFLE_CHECKSUM)
      throw SparkException("Unsupported block id type: " + id.name());
    const std::string name = id.name();
    return root_ + conf_.appId + "/" + std::to_string(id.shuffleId) + "/" + std::to_string(java_non_negative_hash(name)) + "/" + name;
  }
  // ${rootDir}${mapId % folderPrefixes}/${appId}/${shuffleId}/${blockId.name}   (:142-143)
  const long long idx = (long long)(id.mapId % conf_.folderPrefixes);
  return root_ + std::to_string(idx) + "/" + conf_.appId + "/" + std::to_string(id.shuffleId) + "/" + id.name();
}

std::vector<BlockId> S3ShuffleDispatcher::listShuffleIndices(int shuffleId) const {
  if (conf_.useSparkShuffleFetch) throw SparkException("Not supported.");  // (:147-149)
  std::vector<BlockId> out;
  for (int idx = 0; idx < conf_.folderPrefixes; idx++) {
    const std::string dir = root_ + std::to_string(idx) + "/" + conf_.appId + "/" + std::to_string(shuffleId) + "/";
    DIR* d = ::opendir(dir.c_str());
    if (!d) continue;  // IOException -> empty (:166-168)
    while (dirent* e = ::readdir(d)) {
      const std::string n = e->d_name;
      if (!ends_with(n, ".index")) continue;
      int s = 0, r = 0;
      long long m = 0;
      if (sscanf(n.c_str(), "shuffle_%d_%lld_%d.index", &s, &m, &r) == 3) out.push_back(BlockId::ShuffleIndexBlockId(s, m));
    }
    ::closedir(d);
  }
  std::sort(out.begin(), out.end(), [](const BlockId& a, const BlockId& b) { return a.mapId < b.mapId; });
  return out;
}

void S3ShuffleDispatcher::createBlock(const BlockId& id, const void* data, size_t n) const {
  const std::string path = getPath(id);
  mkdirs(path.substr(0, path.rfind('/')));
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) throw IOException("cannot create " + path);
  if (n && fwrite(data, 1, n, f) != n) {
    fclose(f);
    throw IOException("short write to " + path);
  }
  fclose(f);
}

int64_t S3ShuffleDispatcher::blockLength(const BlockId& id) const {
  struct stat st;
  if (::stat(getPath(id).c_str(), &st) != 0) return -1;
  return (int64_t)st.st_size;
}

std::vector<uint8_t> S3ShuffleDispatcher::readBlockRange(const BlockId& id, int64_t pos, int64_t n) const {
  const std::string path = getPath(id);
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) throw IOException("cannot open " + path);
  std::vector<uint8_t> out((size_t)n);
  // S3ShuffleBlockStream.read: positioned readFully(startPosition + numBytes, ...)   (:73-93)
  if (fseek(f, (long)pos, SEEK_SET) != 0 || (n && fread(out.data(), 1, (size_t)n, f) != (size_t)n)) {
    fclose(f);
    throw IOException("short read from " + path);
  }
  fclose(f);
  return out;
}

std::vector<uint8_t> S3ShuffleDispatcher::readBlock(const BlockId& id) const {
  const int64_t n = blockLength(id);
  if (n < 0) throw IOException("missing block " + id.name());
  return readBlockRange(id, 0, n);
}

void S3ShuffleDispatcher::removeShuffle(int shuffleId) const {
  for (int idx = 0; idx < conf_.folderPrefixes; idx++)
    rm_rf(root_ + std::to_string(idx) + "/" + conf_.appId + "/" + std::to_string(shuffleId));
}
void S3ShuffleDispatcher::removeRoot() const {
  for (int idx = 0; idx < conf_.folderPrefixes; idx++) rm_rf(root_ + std::to_string(idx) + "/" + conf_.appId);
}

int S3ShuffleDispatcher::codecId() const {
  if (!conf_.compress) return S3S_CODEC_NONE;
  if (conf_.codec == "lz4") return S3S_CODEC_LZ4;
  if (conf_.codec == "snappy") return S3S_CODEC_SNAPPY;
  throw SparkException("Codec " + conf_.codec + " is not supported by the GPU shuffle codec path");
}
int S3ShuffleDispatcher::checksumId() const {
  if (!conf_.checksumEnabled) return S3S_CHECKSUM_NONE;
  if (conf_.checksumAlgorithm == "ADLER32") return S3S_CHECKSUM_ADLER32;
  if (conf_.checksumAlgorithm == "CRC32") return S3S_CHECKSUM_CRC32;
  if (conf_.checksumAlgorithm == "CRC32C") return S3S_CHECKSUM_CRC32C;  // (Spark 4's third algorithm; the library computes it)
  // S3ShuffleHelper.createChecksumAlgorithm (:94-103) knows the first two
  throw std::invalid_argument("Unsupported shuffle checksum algorithm: " + conf_.checksumAlgorithm + ".");
}
int S3ShuffleDispatcher::deviceForMap(int64_t mapId) const { return (int)(mapId % ngpu_); }

// ---- helper: index / checksum arrays as big-endian longs ---------------------------------------------
namespace S3ShuffleHelper {

void writeArrayAsBlock(const S3ShuffleDispatcher& d, const BlockId& id, const std::vector<int64_t>& array) {
  std::vector<uint8_t> be(array.size() * 8);
  for (size_t i = 0; i < array.size(); i++)
    for (int b = 0; b < 8; b++) be[8 * i + b] = (uint8_t)((uint64_t)array[i] >> (8 * (7 - b)));  // DataOutputStream.writeLong
  d.createBlock(id, be.data(), be.size());
}

void writePartitionLengths(const S3ShuffleDispatcher& d, int shuffleId, int64_t mapId,
                           const std::vector<int64_t>& partitionLengths) {
  if (partitionLengths.empty()) throw std::out_of_range("head of empty array");  // .head on an empty Array (:45)
  std::vector<int64_t> acc(partitionLengths.size() + 1, 0);
  for (size_t i = 0; i < partitionLengths.size(); i++) acc[i + 1] = acc[i] + partitionLengths[i];
  writeArrayAsBlock(d, BlockId::ShuffleIndexBlockId(shuffleId, mapId), acc);
}

void writeChecksum(const S3ShuffleDispatcher& d, int shuffleId, int64_t mapId, const std::vector<int64_t>& checksums) {
  writeArrayAsBlock(d, BlockId::ShuffleChecksumBlockId(shuffleId, mapId), checksums);
}

std::vector<int64_t> readBlockAsArray(const S3ShuffleDispatcher& d, const BlockId& id) {
  const std::vector<uint8_t> raw = d.readBlock(id);
  if (raw.size() % 8 != 0) throw SparkException("Unexpected file length when reading " + id.name());  // :113-115
  std::vector<int64_t> out(raw.size() / 8);
  for (size_t i = 0; i < out.size(); i++) {
    uint64_t v = 0;
    for (int b = 0; b < 8; b++) v = (v << 8) | raw[8 * i + b];
    out[i] = (int64_t)v;
  }
  return out;
}

std::vector<int64_t> getPartitionLengths(const S3ShuffleDispatcher& d, int shuffleId, int64_t mapId) {
  return readBlockAsArray(d, BlockId::ShuffleIndexBlockId(shuffleId, mapId));
}
std::vector<int64_t> getChecksums(const S3ShuffleDispatcher& d, int shuffleId, int64_t mapId) {
  return readBlockAsArray(d, BlockId::ShuffleChecksumBlockId(shuffleId, mapId));
}

}  // namespace S3ShuffleHelper

// ---- map side ------------------------------------------------------------------------------------------
S3ShuffleMapOutputWriter::S3ShuffleMapOutputWriter(const S3ShuffleDispatcher& d, int shuffleId, int64_t mapId,
                                                   int numPartitions)
    : d_(d), shuffleId_(shuffleId), mapId_(mapId), numPartitions_(numPartitions), srcOffsets_((size_t)numPartitions + 1, 0) {}

S3ShuffleMapOutputWriter::~S3ShuffleMapOutputWriter() {
  PinnedPool::process().release(stage_);
  releaseContext(d_, d_.deviceForMap(mapId_), ctx_);
}

void S3ShuffleMapOutputWriter::getPartitionWriter(int reducePartitionId) {
  if (reducePartitionId <= lastPartitionWriterId_)
    throw std::runtime_error("Precondition: Expect a monotonically increasing reducePartitionId.");  // :68-70
  if (reducePartitionId >= numPartitions_)
    throw std::runtime_error("Precondition: Invalid partition id.");  // :71-73
  closePartition();
  // partitions skipped over stay empty
  for (int p = lastPartitionWriterId_ + 1; p <= reducePartitionId; p++) srcOffsets_[(size_t)p] = stageLen_;
  lastPartitionWriterId_ = reducePartitionId;
  streamClosed_ = false;
}

void S3ShuffleMapOutputWriter::write(const void* bytes, size_t len) {
  if (streamClosed_ || lastPartitionWriterId_ < 0) throw IOException("Partition writer stream is closed.");  // :183-184
  // the task's serialized bytes go straight into page-locked staging (what the JVM shim exposes as a
  // direct ByteBuffer); it grows by doubling inside the process-wide pool
  if (stageLen_ + (int64_t)len > stageCap_) {
    // first buffer: what the previous map task of this process staged (tasks of a stage are alike), so the
    // steady state never re-grows; otherwise double
    int64_t cap = std::max<int64_t>(stageCap_ * 2, std::max<int64_t>(4ll << 20, g_lastStaged.load(std::memory_order_relaxed)));
    while (cap < stageLen_ + (int64_t)len) cap *= 2;
    uint8_t* bigger = PinnedPool::process().acquire(cap);
    if (stageLen_) memcpy(bigger, stage_, (size_t)stageLen_);
    PinnedPool::process().release(stage_);
    stage_ = bigger;
    stageCap_ = cap;
  }
  if (len == 0) return;  // (write(b, off, 0) is legal and `bytes` may then be null)
  memcpy(stage_ + stageLen_, bytes, len);
  stageLen_ += (int64_t)len;
}

int64_t S3ShuffleMapOutputWriter::getNumBytesWritten() const {
  if (lastPartitionWriterId_ < 0) return 0;
  return stageLen_ - srcOffsets_[(size_t)lastPartitionWriterId_];
}

void S3ShuffleMapOutputWriter::closePartition() { streamClosed_ = true; }

void S3ShuffleMapOutputWriter::markSegment() {
  if (streamClosed_ || lastPartitionWriterId_ < 0) throw IOException("Partition writer stream is closed.");
  if (cuts_.empty()) cuts_.resize((size_t)numPartitions_);
  std::vector<int64_t>& c = cuts_[(size_t)lastPartitionWriterId_];
  if (stageLen_ > srcOffsets_[(size_t)lastPartitionWriterId_] && (c.empty() || c.back() != stageLen_)) c.push_back(stageLen_);
}

std::vector<int64_t> S3ShuffleMapOutputWriter::commitAllPartitions() {
  if (committed_) throw std::runtime_error("commitAllPartitions called twice");
  const bool timing = getenv("S3SH_TIMING") != nullptr;
  auto now = [] { return std::chrono::steady_clock::now(); };
  const auto t0 = now();
  closePartition();
  for (int p = lastPartitionWriterId_ + 1; p <= numPartitions_; p++) srcOffsets_[(size_t)p] = stageLen_;
  const int codec = d_.codecId(), algo = d_.checksumId();
  if (!ctx_) ctx_ = acquireContext(d_, d_.deviceForMap(mapId_));
  // multi-spill merge (markSegment): one stream per piece
  std::vector<int64_t> segOffsets;
  std::vector<int32_t> partFirstSeg;
  if (!cuts_.empty()) {
    partFirstSeg.assign((size_t)numPartitions_ + 1, 0);
    for (int p = 0; p < numPartitions_; p++) {
      partFirstSeg[(size_t)p] = (int32_t)segOffsets.size();
      segOffsets.push_back(srcOffsets_[(size_t)p]);
      for (int64_t c : cuts_[(size_t)p])
        if (c > segOffsets.back() && c < srcOffsets_[(size_t)p + 1]) segOffsets.push_back(c);
    }
    partFirstSeg[(size_t)numPartitions_] = (int32_t)segOffsets.size();
    segOffsets.push_back(srcOffsets_[(size_t)numPartitions_]);
  }
  const int64_t cap = cuts_.empty() ? s3s_max_compressed_size(ctx_, codec, srcOffsets_.data(), numPartitions_)
                                    : s3s_max_compressed_size_segments(ctx_, codec, segOffsets.data(), (int32_t)segOffsets.size() - 1);
  if (cap < 0) throw std::runtime_error("Precondition: invalid partition offsets");
  struct PinnedOut {  // the .data image, page-locked: D2H is plain DMA, then one write to the store
    uint8_t* p;
    ~PinnedOut() { PinnedPool::process().release(p); }
  } data{PinnedPool::process().acquire(cap + 1)};
  std::vector<int64_t> index((size_t)numPartitions_ + 1, 0), sums((size_t)std::max(numPartitions_, 1), 0);
  int64_t total = 0;
  int rc;
  if (cuts_.empty()) {
    rc = s3s_compress_map_output(ctx_, codec, algo, stage_, srcOffsets_.data(), numPartitions_, data.p, cap,
                                 index.data(), algo == S3S_CHECKSUM_NONE ? nullptr : sums.data(), &total);
  } else {
    rc = s3s_compress_map_output_segments(ctx_, codec, algo, stage_, segOffsets.data(), (int32_t)segOffsets.size() - 1,
                                          partFirstSeg.data(), numPartitions_, data.p, cap, index.data(),
                                          algo == S3S_CHECKSUM_NONE ? nullptr : sums.data(), &total);
  }
  if (rc != S3S_OK) throw IOException(std::string("s3s_compress_map_output: ") + s3s_last_error(ctx_));
  const auto t1 = now();
  std::vector<int64_t> partitionLengths((size_t)numPartitions_);
  for (int p = 0; p < numPartitions_; p++) partitionLengths[(size_t)p] = index[(size_t)p + 1] - index[(size_t)p];
  // the .data object: one block per map task, partitions in ascending order (:58).  The reference opens it lazily
  // on the first write (initStream, :43-49), so a map task that wrote nothing leaves no .data object — also with
  // alwaysCreateIndex, which only adds the .index / .checksum pair
  if (total > 0) d_.createBlock(BlockId::ShuffleDataBlockId(shuffleId_, mapId_), data.p, (size_t)total);
  // index and checksum (:111-116)
  if (total > 0 || d_.conf().alwaysCreateIndex) {
    // (the reference's writePartitionLengths calls .head on the lengths, S3ShuffleHelper.scala:45: zero partitions throw)
    if (numPartitions_ == 0) throw std::runtime_error("Precondition: writePartitionLengths needs at least one partition");
    S3ShuffleHelper::writePartitionLengths(d_, shuffleId_, mapId_, partitionLengths);
    if (d_.conf().checksumEnabled) {
      sums.resize((size_t)numPartitions_);
      S3ShuffleHelper::writeChecksum(d_, shuffleId_, mapId_, sums);
    }
  }
  committed_ = true;
  g_lastStaged.store(stageLen_, std::memory_order_relaxed);
  if (timing)
    fprintf(stderr, "[s3sh] map %lld: %lld B staged, compress call %.2f ms, store writes %.2f ms\n", (long long)mapId_,
            (long long)stageLen_, std::chrono::duration<double, std::milli>(t1 - t0).count(),
            std::chrono::duration<double, std::milli>(now() - t1).count());
  PinnedPool::process().release(stage_);
  stage_ = nullptr;
  stageCap_ = stageLen_ = 0;
  return partitionLengths;
}

void S3ShuffleMapOutputWriter::abort() {
  // the reference closes its streams (:120-134); nothing has reached the store before commit here
  PinnedPool::process().release(stage_);
  stage_ = nullptr;
  stageCap_ = stageLen_ = 0;
  streamClosed_ = true;
  committed_ = true;
}

std::vector<int64_t> S3SingleSpillShuffleMapOutputWriter::transferMapSpillFile(const std::string& mapSpillFile,
                                                                                const std::vector<int64_t>& partitionLengths,
                                                                                const std::vector<int64_t>& /*checksums*/) {
  const int n = (int)partitionLengths.size();
  std::vector<int64_t> offs((size_t)n + 1, 0);
  for (int p = 0; p < n; p++) {
    if (partitionLengths[(size_t)p] < 0) throw std::runtime_error("Precondition: negative partition length");
    offs[(size_t)p + 1] = offs[(size_t)p] + partitionLengths[(size_t)p];
  }
  const int64_t bytes = offs[(size_t)n];
  struct Pinned {
    uint8_t* p;
    ~Pinned() { PinnedPool::process().release(p); }
  };
  Pinned in{PinnedPool::process().acquire(bytes + 1)};
  {
    FILE* f = fopen(mapSpillFile.c_str(), "rb");
    if (!f) throw IOException("cannot open " + mapSpillFile);
    const size_t got = bytes ? fread(in.p, 1, (size_t)bytes, f) : 0;
    const bool more = fgetc(f) != EOF;
    fclose(f);
    if ((int64_t)got != bytes || more) throw IOException("spill file length does not match partitionLengths: " + mapSpillFile);
  }
  const int codec = d_.codecId(), algo = d_.checksumId();
  const int device = d_.deviceForMap(mapId_);
  s3s_ctx* ctx = acquireContext(d_, device);
  struct CtxGuard {
    const S3ShuffleDispatcher& d;
    int device;
    s3s_ctx* c;
    ~CtxGuard() { releaseContext(d, device, c); }
  } guard{d_, device, ctx};
  const int64_t cap = s3s_max_compressed_size(ctx, codec, offs.data(), n);
  if (cap < 0) throw std::runtime_error("Precondition: invalid partition lengths");
  Pinned out{PinnedPool::process().acquire(cap + 1)};
  std::vector<int64_t> index((size_t)n + 1, 0), sums((size_t)std::max(n, 1), 0);
  int64_t total = 0;
  const int rc = s3s_compress_map_output(ctx, codec, algo, in.p, offs.data(), n, out.p, cap, index.data(),
                                         algo == S3S_CHECKSUM_NONE ? nullptr : sums.data(), &total);
  if (rc != S3S_OK) throw IOException(std::string("s3s_compress_map_output: ") + s3s_last_error(ctx));
  d_.createBlock(BlockId::ShuffleDataBlockId(shuffleId_, mapId_), out.p, (size_t)total);
  std::vector<int64_t> lengths((size_t)n);
  for (int p = 0; p < n; p++) lengths[(size_t)p] = index[(size_t)p + 1] - index[(size_t)p];
  if (d_.conf().checksumEnabled) {  // :59-61
    sums.resize((size_t)n);
    S3ShuffleHelper::writeChecksum(d_, shuffleId_, mapId_, sums);
  }
  S3ShuffleHelper::writePartitionLengths(d_, shuffleId_, mapId_, lengths);  // :62, unconditional
  ::remove(mapSpillFile.c_str());  // the reference moves the file (:41) or streams it out of the local dir
  return lengths;
}

// ---- reduce side ---------------------------------------------------------------------------------------
S3ShuffleReader::S3ShuffleReader(const S3ShuffleDispatcher& d, int shuffleId, int startPartition, int endPartition,
                                 bool doBatchFetch)
    : d_(d), shuffleId_(shuffleId), start_(startPartition), end_(endPartition), batch_(doBatchFetch) {}

S3ShuffleReader::~S3ShuffleReader() { releaseContext(d_, ctxDevice_, ctx_); }

std::vector<BlockRequest> S3ShuffleReader::blockRequests() const {
  std::vector<BlockRequest> reqs;
  const int algo = d_.checksumId();
  // computeShuffleBlocks with useBlockManager=false: list the .index objects (S3ShuffleReader.scala:182-195)
  for (const BlockId& idx : d_.listShuffleIndices(shuffleId_)) {
    const std::vector<int64_t> lengths = S3ShuffleHelper::getPartitionLengths(d_, shuffleId_, idx.mapId);
    if ((int)lengths.size() < end_ + 1) throw SparkException("Unexpected file length when reading " + idx.name());
    std::vector<int64_t> sums;
    if (algo != S3S_CHECKSUM_NONE) sums = S3ShuffleHelper::getChecksums(d_, shuffleId_, idx.mapId);
    std::vector<std::pair<int, int>> ranges;
    if (batch_) ranges.emplace_back(start_, end_);
    else
      for (int r = start_; r < end_; r++) ranges.emplace_back(r, r + 1);
    for (const auto& rg : ranges) {
      const int r0 = rg.first, r1 = rg.second;
      BlockRequest rq;
      rq.startPosition = lengths[(size_t)r0];
      rq.maxBytes = lengths[(size_t)r1] - lengths[(size_t)r0];
      if (rq.maxBytes == 0) continue;  // S3ShuffleReader.scala:91-93 filters empty blocks
      rq.id = batch_ && r1 - r0 > 1 ? BlockId::ShuffleBlockBatchId(shuffleId_, idx.mapId, r0, r1)
                                    : BlockId::ShuffleBlockId(shuffleId_, idx.mapId, r0);
      rq.dataBlock = BlockId::ShuffleDataBlockId(shuffleId_, idx.mapId);
      rq.rel.resize((size_t)(r1 - r0) + 1);
      for (int r = r0; r <= r1; r++) rq.rel[(size_t)(r - r0)] = lengths[(size_t)r] - rq.startPosition;
      if (algo != S3S_CHECKSUM_NONE) rq.sums.assign(sums.begin() + r0, sums.begin() + r1);
      rq.device = d_.deviceForMap(idx.mapId);
      reqs.push_back(std::move(rq));
    }
  }
  return reqs;
}

std::vector<FetchedBlock> S3ShuffleReader::read() {
  std::vector<FetchedBlock> out;
  S3BufferedPrefetchIterator it(d_, blockRequests());
  while (it.hasNext()) {
    PrefetchedBlock b = it.next();
    out.push_back(FetchedBlock{b.id, std::vector<uint8_t>(b.data, b.data + b.size)});
    it.release(b);
  }
  // completion order is not deterministic (neither is the reference's); callers of read() get map order
  std::sort(out.begin(), out.end(), [](const FetchedBlock& a, const FetchedBlock& b) {
    return a.id.mapId != b.id.mapId ? a.id.mapId < b.id.mapId : a.id.reduceId < b.id.reduceId;
  });
  return out;
}

std::vector<FetchedBlock> S3ShuffleReader::readSequential() {
  std::vector<FetchedBlock> out;
  const int codec = d_.codecId(), algo = d_.checksumId();
  for (const BlockRequest& rq : blockRequests()) {
    FetchedBlock fb{rq.id, {}};
    const std::vector<uint8_t> comp = d_.readBlockRange(rq.dataBlock, rq.startPosition, rq.maxBytes);
    if (ctx_ && ctxDevice_ != rq.device) {
      releaseContext(d_, ctxDevice_, ctx_);
      ctx_ = nullptr;
    }
    if (!ctx_) ctx_ = acquireContext(d_, ctxDevice_ = rq.device);
    int64_t decoded = 0;
    if (s3s_decompressed_size(ctx_, codec, comp.data(), (int64_t)comp.size(), &decoded) != S3S_OK)
      throw IOException("Stream is corrupted");
    fb.bytes.resize((size_t)decoded);
    int64_t out_len = 0;
    int32_t bad = -1;
    const int rc = s3s_decompress_range(ctx_, codec, algo, comp.data(), (int64_t)comp.size(), rq.rel.data(),
                                        (algo == S3S_CHECKSUM_NONE || rq.sums.empty()) ? nullptr : rq.sums.data(),
                                        (int32_t)rq.rel.size() - 1, fb.bytes.data(), decoded, &out_len, &bad);
    if (rc == S3S_E_CHECKSUM) throw SparkException("Invalid checksum detected for " + fb.id.name());  // S3ChecksumValidationStream.scala:72-74
    if (rc == S3S_E_BAD_FRAME) throw IOException("Stream is corrupted");
    if (rc != S3S_OK) throw IOException(std::string("s3s_decompress_range: ") + s3s_last_error(ctx_));
    fb.bytes.resize((size_t)out_len);
    out.push_back(std::move(fb));
  }
  return out;
}

}  // namespace s3shuffle

// ---- flat C layer for the Python tests (tests/test_host_mirror.py) ------------------------------------------
using namespace s3shuffle;
namespace {
thread_local std::string g_err;
template <typename F>
int guarded(F&& f) {
  try {
    f();
    g_err.clear();
    return 0;
  } catch (const SparkException& e) {
    g_err = std::string("SparkException: ") + e.what();
    return -2;
  } catch (const IOException& e) {
    g_err = std::string("IOException: ") + e.what();
    return -3;
  } catch (const std::exception& e) {
    g_err = std::string("RuntimeException: ") + e.what();
    return -1;
  }
}
}  // namespace

extern "C" {
const char* s3sh_last_error() { return g_err.c_str(); }

void* s3sh_dispatcher_create(const char* rootDir, const char* appId, int folderPrefixes, int alwaysCreateIndex,
                             int checksumEnabled, const char* checksumAlgorithm, int compress, const char* codec,
                             int blockSize, int numGpus) {
  void* r = nullptr;
  guarded([&] {
    Conf c;
    c.rootDir = rootDir;
    c.appId = appId;
    c.folderPrefixes = folderPrefixes;
    c.alwaysCreateIndex = alwaysCreateIndex != 0;
    c.checksumEnabled = checksumEnabled != 0;
    c.checksumAlgorithm = checksumAlgorithm;
    c.compress = compress != 0;
    c.codec = codec;
    c.blockSize = blockSize;
    c.numGpus = numGpus;
    r = new S3ShuffleDispatcher(c);
  });
  return r;
}
void s3sh_dispatcher_destroy(void* d) { delete static_cast<S3ShuffleDispatcher*>(d); }
int s3sh_dispatcher_set_use_spark_shuffle_fetch(void* d, int on) {
  return guarded([&] { static_cast<S3ShuffleDispatcher*>(d)->setUseSparkShuffleFetch(on != 0); });
}
int s3sh_dispatcher_set_fetch_thread_predictor(void* d, int on) {
  return guarded([&] { static_cast<S3ShuffleDispatcher*>(d)->setFetchThreadPredictor(on != 0); });
}
int s3sh_get_path(void* d, int kind, int shuffleId, long long mapId, int r0, int r1, char* out, int cap) {
  return guarded([&] {
    BlockId id{(BlockId::Kind)kind, shuffleId, mapId, r0, r1};
    snprintf(out, (size_t)cap, "%s", static_cast<S3ShuffleDispatcher*>(d)->getPath(id).c_str());
  });
}
int s3sh_device_for_map(void* d, long long mapId) { return static_cast<S3ShuffleDispatcher*>(d)->deviceForMap(mapId); }
int s3sh_write_partition_lengths(void* d, int shuffleId, long long mapId, const long long* lens, int n) {
  return guarded([&] {
    S3ShuffleHelper::writePartitionLengths(*static_cast<S3ShuffleDispatcher*>(d), shuffleId, mapId,
                                           std::vector<int64_t>(lens, lens + n));
  });
}
int s3sh_read_block_as_array(void* d, int kind, int shuffleId, long long mapId, long long* out, int cap, int* n) {
  return guarded([&] {
    BlockId id{(BlockId::Kind)kind, shuffleId, mapId, 0, 1};
    const auto v = S3ShuffleHelper::readBlockAsArray(*static_cast<S3ShuffleDispatcher*>(d), id);
    *n = (int)v.size();
    for (int i = 0; i < *n && i < cap; i++) out[i] = v[(size_t)i];
  });
}
int s3sh_remove_shuffle(void* d, int shuffleId) {
  return guarded([&] { static_cast<S3ShuffleDispatcher*>(d)->removeShuffle(shuffleId); });
}
int s3sh_remove_root(void* d) {
  return guarded([&] { static_cast<S3ShuffleDispatcher*>(d)->removeRoot(); });
}

void* s3sh_writer_create(void* d, int shuffleId, long long mapId, int numPartitions) {
  void* r = nullptr;
  guarded([&] { r = new S3ShuffleMapOutputWriter(*static_cast<S3ShuffleDispatcher*>(d), shuffleId, mapId, numPartitions); });
  return r;
}
void s3sh_writer_destroy(void* w) { delete static_cast<S3ShuffleMapOutputWriter*>(w); }
int s3sh_writer_get_partition_writer(void* w, int reducePartitionId) {
  return guarded([&] { static_cast<S3ShuffleMapOutputWriter*>(w)->getPartitionWriter(reducePartitionId); });
}
int s3sh_writer_write(void* w, const void* bytes, long long len) {
  return guarded([&] { static_cast<S3ShuffleMapOutputWriter*>(w)->write(bytes, (size_t)len); });
}
int s3sh_writer_mark_segment(void* w) {
  return guarded([&] { static_cast<S3ShuffleMapOutputWriter*>(w)->markSegment(); });
}
int s3sh_writer_close_partition(void* w) {
  return guarded([&] { static_cast<S3ShuffleMapOutputWriter*>(w)->closePartition(); });
}
long long s3sh_writer_num_bytes_written(void* w) { return static_cast<S3ShuffleMapOutputWriter*>(w)->getNumBytesWritten(); }
int s3sh_writer_commit(void* w, long long* outLengths) {
  return guarded([&] {
    const auto v = static_cast<S3ShuffleMapOutputWriter*>(w)->commitAllPartitions();
    for (size_t i = 0; i < v.size(); i++) outLengths[i] = v[i];
  });
}
int s3sh_single_spill_transfer(void* d, int shuffleId, long long mapId, const char* spillFile, const long long* partitionLengths,
                               int numPartitions, long long* outLengths) {
  return guarded([&] {
    S3SingleSpillShuffleMapOutputWriter w(*static_cast<S3ShuffleDispatcher*>(d), shuffleId, mapId);
    std::vector<int64_t> pl(partitionLengths, partitionLengths + numPartitions);
    const std::vector<int64_t> l = w.transferMapSpillFile(spillFile, pl, {});
    for (size_t i = 0; i < l.size(); i++) outLengths[i] = l[i];
  });
}
int s3sh_writer_abort(void* w) {
  return guarded([&] { static_cast<S3ShuffleMapOutputWriter*>(w)->abort(); });
}

// reader: fills a result object; accessors copy block by block
struct s3sh_read_result {
  std::vector<FetchedBlock> blocks;
};
void* s3sh_reader_read(void* d, int shuffleId, int startPartition, int endPartition, int doBatchFetch) {
  s3sh_read_result* r = nullptr;
  const int rc = guarded([&] {
    S3ShuffleReader rd(*static_cast<S3ShuffleDispatcher*>(d), shuffleId, startPartition, endPartition, doBatchFetch != 0);
    r = new s3sh_read_result{rd.read()};
  });
  return rc == 0 ? r : nullptr;
}
void* s3sh_reader_read_sequential(void* d, int shuffleId, int startPartition, int endPartition, int doBatchFetch) {
  s3sh_read_result* r = nullptr;
  const int rc = guarded([&] {
    S3ShuffleReader rd(*static_cast<S3ShuffleDispatcher*>(d), shuffleId, startPartition, endPartition, doBatchFetch != 0);
    r = new s3sh_read_result{rd.readSequential()};
  });
  return rc == 0 ? r : nullptr;
}
// streams every block of the range through the prefetch pipeline WITHOUT copying it out (the consumer
// touches one byte per page and releases): out[0] blocks, [1] compressed bytes, [2] decoded bytes,
// [3] pinned high-water (compressed), [4] pinned high-water (decoded), [5] microseconds waiting in next()
int s3sh_reader_consume_prefetched(void* d, int shuffleId, int startPartition, int endPartition, int doBatchFetch,
                                   long long* out) {
  return guarded([&] {
    const S3ShuffleDispatcher& disp = *static_cast<S3ShuffleDispatcher*>(d);
    S3ShuffleReader rd(disp, shuffleId, startPartition, endPartition, doBatchFetch != 0);
    S3BufferedPrefetchIterator it(disp, rd.blockRequests());
    unsigned acc = 0;
    while (it.hasNext()) {
      PrefetchedBlock b = it.next();
      for (int64_t i = 0; i < b.size; i += 4096) acc += b.data[i];
      it.release(b);
    }
    const S3BufferedPrefetchIterator::Stats st = it.stats();
    out[0] = st.blocks;
    out[1] = st.compressedBytes;
    out[2] = st.decodedBytes;
    out[3] = st.compHighWater;
    out[4] = st.decodedHighWater;
    out[5] = (long long)(st.secondsWaiting * 1e6) + (acc & 0);
  });
}
// the same without the pipeline: block after block on one context, pageable buffers (out[0] blocks, out[2] decoded bytes)
int s3sh_reader_consume_sequential(void* d, int shuffleId, int startPartition, int endPartition, int doBatchFetch,
                                   long long* out) {
  return guarded([&] {
    S3ShuffleReader rd(*static_cast<S3ShuffleDispatcher*>(d), shuffleId, startPartition, endPartition, doBatchFetch != 0);
    const std::vector<FetchedBlock> blocks = rd.readSequential();
    out[0] = (long long)blocks.size();
    out[2] = 0;
    for (const FetchedBlock& b : blocks) out[2] += (long long)b.bytes.size();
  });
}
void s3sh_release_caches() {
  releaseContextCache();
  releasePinnedCache();
}
// prefetch / staging knobs of an existing dispatcher (spark.shuffle.s3.maxBufferSizeTask, .maxConcurrencyTask,
// .gpu.decodeThreads, .gpu.maxDecodedBufferSizeTask); values <= 0 keep the current setting
void s3sh_dispatcher_set_prefetch(void* d, long long maxBufferSizeTask, int maxConcurrencyTask, int gpuDecodeThreads,
                                  long long gpuMaxDecodedBufferSizeTask) {
  static_cast<S3ShuffleDispatcher*>(d)->setPrefetch(maxBufferSizeTask, maxConcurrencyTask, gpuDecodeThreads,
                                                     gpuMaxDecodedBufferSizeTask);
}
int s3sh_result_count(void* r) { return (int)static_cast<s3sh_read_result*>(r)->blocks.size(); }
long long s3sh_result_block_len(void* r, int i) { return (long long)static_cast<s3sh_read_result*>(r)->blocks[(size_t)i].bytes.size(); }
void s3sh_result_block_info(void* r, int i, long long* mapId, int* r0, int* r1, char* name, int cap) {
  const FetchedBlock& b = static_cast<s3sh_read_result*>(r)->blocks[(size_t)i];
  *mapId = b.id.mapId;
  *r0 = b.id.reduceId;
  *r1 = b.id.endReduceId;
  snprintf(name, (size_t)cap, "%s", b.id.name().c_str());
}
void s3sh_result_block_copy(void* r, int i, void* dst) {
  const FetchedBlock& b = static_cast<s3sh_read_result*>(r)->blocks[(size_t)i];
  memcpy(dst, b.bytes.data(), b.bytes.size());
}
void s3sh_result_destroy(void* r) { delete static_cast<s3sh_read_result*>(r); }
}
