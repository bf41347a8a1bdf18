// s3shuffle_prefetch.cpp — page-locked staging pool and the prefetch -> H2D -> verify+decode -> D2H
// pipeline in front of the consumer (see s3shuffle_host.h).  The reference
// (storage/S3BufferedPrefetchIterator.scala:16-200) prefetches COMPRESSED blocks into heap buffers and
// leaves verify + decompress to the consuming task thread; on MI355X the decode is ~140 GB/s in HBM and
// the path is bounded by PCIe and by the store, so the stages must overlap and every host buffer the
// DMA engines touch must be page-locked:
//
//   fetch threads (<= maxConcurrencyTask)   positioned read of the block's byte range straight into a
//                                           pinned buffer (budget maxBufferSizeTask, like the reference)
//   decode threads (gpuDecodeThreads)       one s3s_ctx (HIP stream) each: DMA up, per-partition checksum
//                                           validation, frame discovery, decode, frame hashes, DMA down into a
//                                           pinned buffer (budget gpuMaxDecodedBufferSizeTask)
//   consumer                                next() in completion order, release() gives the memory back
//
// The reference's ThreadPredictor (latency-driven thread count, :32-69) is restated below and can be switched on
// (Conf::fetchThreadPredictor); by default every one of the min(maxConcurrencyTask, number of blocks) fetch threads runs.
#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>

#include "s3shuffle_host.h"

namespace s3shuffle {

// ---- process-wide caches: page-locked buffers and codec contexts --------------------------------------
// hipHostMalloc pins pages (hundreds of milliseconds per GiB) and a fresh s3s_ctx allocates its device
// workspace on first use (hipMalloc / hipFree synchronise the device): neither may happen per block or
// per task.  Both are therefore recycled across tasks — what a JVM shim does with one direct-buffer
// arena and one S3SCodec per task thread.
namespace {
int64_t round_capacity(int64_t bytes) {
  int64_t cap = 1 << 20;  // 1 MiB granularity, powers of two: buffers get reused across blocks
  while (cap < bytes) cap <<= 1;
  return cap;
}

class PinnedCache {
 public:
  static PinnedCache& get() {
    static PinnedCache* c = new PinnedCache();  // never destroyed: outlives every HIP teardown order
    return *c;
  }
  uint8_t* take(int64_t cap, int64_t* got) {
    {
      std::lock_guard<std::mutex> lk(mu_);
      int best = -1;
      for (int i = 0; i < (int)idle_.size(); i++)
        if (idle_[(size_t)i].second >= cap && (best < 0 || idle_[(size_t)i].second < idle_[(size_t)best].second)) best = i;
      if (best >= 0) {
        auto b = idle_[(size_t)best];
        idle_.erase(idle_.begin() + best);
        idleBytes_ -= b.second;
        *got = b.second;
        return b.first;
      }
    }
    void* p = nullptr;
    if (pinned_) {
      p = s3s_host_alloc(cap);
      if (!p) {  // make room and retry once
        trim(0);
        p = s3s_host_alloc(cap);
      }
      // with a GPU present a buffer that cannot be page-locked is an error, never a silent 5x slowdown
      if (!p) throw IOException("s3s_host_alloc(" + std::to_string(cap) + ") failed");
    } else {
      p = ::malloc((size_t)cap);
      if (!p) throw IOException("out of memory");
    }
    *got = cap;
    return static_cast<uint8_t*>(p);
  }
  void give(uint8_t* p, int64_t cap) {
    {
      std::lock_guard<std::mutex> lk(mu_);
      idle_.emplace_back(p, cap);
      idleBytes_ += cap;
    }
    trim(limit_);
  }
  void trim(int64_t keep) {
    std::vector<uint8_t*> drop;
    {
      std::lock_guard<std::mutex> lk(mu_);
      while (!idle_.empty() && idleBytes_ > keep) {
        auto big = std::max_element(idle_.begin(), idle_.end(),
                                    [](const std::pair<uint8_t*, int64_t>& a, const std::pair<uint8_t*, int64_t>& b) { return a.second < b.second; });
        drop.push_back(big->first);
        idleBytes_ -= big->second;
        idle_.erase(big);
      }
    }
    for (uint8_t* p : drop) freeOne(p);
  }

 private:
  void freeOne(uint8_t* p) {
    if (pinned_) s3s_host_free(p);
    else ::free(p);
  }
  // no HIP device at all (a CPU-only box running the host-logic tests): nothing can be compressed there
  // anyway — s3s_create fails loudly — so staging is plain memory instead of an error before the error
  const bool pinned_ = s3s_device_count() > 0;
  std::mutex mu_;
  std::vector<std::pair<uint8_t*, int64_t>> idle_;
  int64_t idleBytes_ = 0;
  int64_t limit_ = 4ll << 30;  // page-locked memory kept warm between tasks
};
}  // namespace

void releasePinnedCache() { PinnedCache::get().trim(0); }

// ---- codec contexts ----------------------------------------------------------------------------------------
namespace {
struct CtxKey {
  int device, codecKey;
  int64_t blockSize;
  bool operator<(const CtxKey& o) const {
    return device != o.device ? device < o.device : (codecKey != o.codecKey ? codecKey < o.codecKey : blockSize < o.blockSize);
  }
};
std::mutex g_ctxMu;
std::multimap<CtxKey, s3s_ctx*> g_ctxIdle;
}  // namespace

s3s_ctx* acquireContext(const S3ShuffleDispatcher& d, int device) {
  const Conf& cf = d.conf();
  const CtxKey key{device, !cf.compress ? 0 : (cf.codec == "snappy" ? S3S_OPT_SNAPPY_BLOCK_SIZE : S3S_OPT_LZ4_BLOCK_SIZE),
                   cf.compress ? (int64_t)cf.blockSize : 0};
  {
    std::lock_guard<std::mutex> lk(g_ctxMu);
    auto it = g_ctxIdle.find(key);
    if (it != g_ctxIdle.end()) {
      s3s_ctx* c = it->second;
      g_ctxIdle.erase(it);
      return c;
    }
  }
  s3s_ctx* c = s3s_create(device, 0);
  if (!c) throw IOException(std::string("s3s_create failed: ") + s3s_last_error(nullptr));
  if (key.codecKey && s3s_set_option(c, key.codecKey, key.blockSize) != S3S_OK) {
    const std::string msg = s3s_last_error(c);
    s3s_destroy(c);
    throw IOException(msg);
  }
  return c;
}

void releaseContext(const S3ShuffleDispatcher& d, int device, s3s_ctx* c) {
  if (!c) return;
  const Conf& cf = d.conf();
  const CtxKey key{device, !cf.compress ? 0 : (cf.codec == "snappy" ? S3S_OPT_SNAPPY_BLOCK_SIZE : S3S_OPT_LZ4_BLOCK_SIZE),
                   cf.compress ? (int64_t)cf.blockSize : 0};
  std::lock_guard<std::mutex> lk(g_ctxMu);
  g_ctxIdle.emplace(key, c);
}

void releaseContextCache() {
  std::lock_guard<std::mutex> lk(g_ctxMu);
  for (auto& kv : g_ctxIdle) s3s_destroy(kv.second);
  g_ctxIdle.clear();
}

// ---- PinnedPool: a budget over the process-wide cache ------------------------------------------------
PinnedPool::PinnedPool(int64_t budgetBytes, bool blocking)
    : budget_(std::max<int64_t>(budgetBytes, 1 << 20)), blocking_(blocking) {}

PinnedPool::~PinnedPool() {
  for (auto& u : used_) PinnedCache::get().give(u.first, u.second);
}

uint8_t* PinnedPool::acquire(int64_t bytes) {
  const int64_t cap = round_capacity(std::max<int64_t>(bytes, 1));
  {
    std::unique_lock<std::mutex> lk(mu_);
    cv_.wait(lk, [&] { return !blocking_ || cancelled_ || inUse_ == 0 || inUse_ + cap <= budget_; });
    if (cancelled_) throw IOException("pinned pool cancelled");
    inUse_ += cap;  // reserved while the buffer is fetched from the cache
    high_ = std::max(high_, inUse_);
  }
  int64_t got = 0;
  uint8_t* p = nullptr;
  try {
    p = PinnedCache::get().take(cap, &got);
  } catch (...) {
    std::lock_guard<std::mutex> lk(mu_);
    inUse_ -= cap;
    cv_.notify_all();
    throw;
  }
  std::lock_guard<std::mutex> lk(mu_);
  used_[p] = got;
  reserved_[p] = cap;
  return p;
}

void PinnedPool::release(uint8_t* p) {
  if (!p) return;
  int64_t got = 0;
  {
    std::lock_guard<std::mutex> lk(mu_);
    auto it = used_.find(p);
    if (it == used_.end()) return;
    got = it->second;
    used_.erase(it);
    inUse_ -= reserved_[p];
    reserved_.erase(p);
    cv_.notify_all();
  }
  PinnedCache::get().give(p, got);
}

void PinnedPool::cancel() {
  std::lock_guard<std::mutex> lk(mu_);
  cancelled_ = true;
  cv_.notify_all();
}

int64_t PinnedPool::inUse() const {
  std::lock_guard<std::mutex> lk(mu_);
  return inUse_;
}
int64_t PinnedPool::highWater() const {
  std::lock_guard<std::mutex> lk(mu_);
  return high_;
}

PinnedPool& PinnedPool::process() {
  // map-side staging is the task's own data: never wait for it (a task holds its old staging buffer while
  // it grows into a bigger one — waiting there would be hold-and-wait).  Never destroyed.
  static PinnedPool* pool = new PinnedPool(1ll << 40, /*blocking=*/false);
  return *pool;
}

// ---- positioned read into caller memory ----------------------------------------------------------------
void S3ShuffleDispatcher::readBlockRangeInto(const BlockId& id, int64_t pos, int64_t n, uint8_t* dst) const {
  const std::string path = getPath(id);
  const int fd = ::open(path.c_str(), O_RDONLY);
  if (fd < 0) throw IOException("cannot open " + path);
  int64_t got = 0;
  while (got < n) {  // readFully (S3ShuffleBlockStream.scala:73-93)
    const ssize_t r = ::pread(fd, dst + got, (size_t)(n - got), (off_t)(pos + got));
    if (r <= 0) {
      ::close(fd);
      throw IOException("short read from " + path);
    }
    got += r;
  }
  ::close(fd);
}

// ---- ThreadPredictor (S3BufferedPrefetchIterator.scala:32-69) ----------------------------------------------
ThreadPredictor::ThreadPredictor(int maxThreads) : latencies_((size_t)std::max(maxThreads, 1) + 2, 0) {
  latencies_.front() = INT64_MAX;
  latencies_.back() = INT64_MAX;
}

int ThreadPredictor::predict() {  // (:42-61)
  if (numMeasurements_ < 20 + currentThreads_) return currentThreads_;
  int64_t current = 0;
  for (int64_t v : measurementsNs_) current += v;
  if (current < 500) return currentThreads_;  // "less than 25 ns latency for each request"
  latencies_[(size_t)currentThreads_] = current;
  const int64_t prevValue = latencies_[(size_t)currentThreads_ - 1], nextValue = latencies_[(size_t)currentThreads_ + 1];
  numMeasurements_ = 0;
  if (prevValue < current) currentThreads_ -= 1;
  else if (nextValue < current) currentThreads_ += 1;
  return currentThreads_;
}

int ThreadPredictor::addMeasurementAndPredict(int64_t latencyNs) {  // (:63-69)
  std::lock_guard<std::mutex> lk(mu_);
  if (latencyNs >= 0) {
    measurementsNs_[numMeasurements_ % 20] = latencyNs;
    numMeasurements_++;
  }
  return predict();
}

int ThreadPredictor::current() const {
  std::lock_guard<std::mutex> lk(mu_);
  return currentThreads_;
}

// ---- the pipeline --------------------------------------------------------------------------------------
S3BufferedPrefetchIterator::S3BufferedPrefetchIterator(const S3ShuffleDispatcher& d, std::vector<BlockRequest> requests)
    : d_(d),
      reqs_(std::move(requests)),
      comp_(d.conf().maxBufferSizeTask),
      dec_(d.conf().gpuMaxDecodedBufferSizeTask),
      predictor_(std::max(d.conf().maxConcurrencyTask, 1)) {
  const size_t n = reqs_.size();
  if (n == 0) return;
  const size_t nFetch = std::max<size_t>(1, std::min<size_t>((size_t)std::max(d.conf().maxConcurrencyTask, 1), n));
  const size_t nDecode = std::max<size_t>(1, std::min<size_t>((size_t)std::max(d.conf().gpuDecodeThreads, 1), n));
  fetchersLeft_ = nFetch;
  // "make sure that there's at least a single thread running" (:93): the predictor starts at one; without it all run
  desiredFetchers_ = d.conf().fetchThreadPredictor ? predictor_.addMeasurementAndPredict(-1) : (int)nFetch;
  for (size_t i = 0; i < nFetch; i++) threads_.emplace_back([this, i] { fetchLoop((int)i + 1); });
  for (size_t i = 0; i < nDecode; i++) threads_.emplace_back([this] { decodeLoop(); });
}

S3BufferedPrefetchIterator::~S3BufferedPrefetchIterator() {
  {
    std::lock_guard<std::mutex> lk(mu_);
    stop_ = true;
  }
  // threads blocked on a budget leave with an exception, threads blocked on a queue see stop_
  comp_.cancel();
  dec_.cancel();
  cvFetched_.notify_all();
  cvDone_.notify_all();
  cvPark_.notify_all();
  for (auto& t : threads_) t.join();
  for (auto& f : fetched_) comp_.release(f.comp);
  for (auto& dn : done_) dec_.release(dn.out);
}

void S3BufferedPrefetchIterator::fetchLoop(int id) {
  for (;;) {
    size_t i;
    {
      std::unique_lock<std::mutex> lk(mu_);
      // a thread above the predicted count parks until the count rises or the work is gone ("if (threadId >
      // desiredActiveThreads.get()) return", :116-119 — here the thread is kept instead of being started again later)
      cvPark_.wait(lk, [&] { return stop_ || nextReq_ >= reqs_.size() || id <= desiredFetchers_; });
      if (stop_ || nextReq_ >= reqs_.size()) break;
      i = nextReq_++;
      if (nextReq_ >= reqs_.size()) cvPark_.notify_all();
    }
    const BlockRequest& rq = reqs_[i];
    uint8_t* buf = nullptr;
    Done err{i, nullptr, 0, 0, ""};
    try {
      buf = comp_.acquire(rq.maxBytes);
      d_.readBlockRangeInto(rq.dataBlock, rq.startPosition, rq.maxBytes, buf);
    } catch (const std::exception& e) {
      err.errKind = 2;
      err.err = e.what();
    }
    std::lock_guard<std::mutex> lk(mu_);
    if (stop_) {
      comp_.release(buf);
      break;
    }
    if (err.errKind) {
      comp_.release(buf);
      done_.push_back(std::move(err));
      cvDone_.notify_all();
    } else {
      fetched_.push_back(Fetched{i, buf});
      cvFetched_.notify_one();
    }
  }
  std::lock_guard<std::mutex> lk(mu_);
  if (--fetchersLeft_ == 0) cvFetched_.notify_all();
}

void S3BufferedPrefetchIterator::decodeLoop() {
  std::map<int, s3s_ctx*> ctxs;  // one context per device this thread meets
  const int codec = d_.codecId(), algo = d_.checksumId();
  for (;;) {
    Fetched f{0, nullptr};
    {
      std::unique_lock<std::mutex> lk(mu_);
      cvFetched_.wait(lk, [&] { return stop_ || !fetched_.empty() || fetchersLeft_ == 0; });
      if (stop_ || fetched_.empty()) break;  // (empty and no fetcher left: everything has been decoded)
      f = fetched_.front();
      fetched_.pop_front();
    }
    const BlockRequest& rq = reqs_[f.req];
    Done dn{f.req, nullptr, 0, 0, ""};
    try {
      s3s_ctx*& ctx = ctxs[rq.device];
      if (!ctx) ctx = acquireContext(d_, rq.device);
      int64_t decoded = 0;
      if (s3s_decompressed_size(ctx, codec, f.comp, rq.maxBytes, &decoded) != S3S_OK) throw IOException("Stream is corrupted");
      dn.out = dec_.acquire(decoded);
      int32_t bad = -1;
      const int rc = s3s_decompress_range(ctx, codec, algo, f.comp, rq.maxBytes, rq.rel.data(),
                                          (algo == S3S_CHECKSUM_NONE || rq.sums.empty()) ? nullptr : rq.sums.data(),
                                          (int32_t)rq.rel.size() - 1, dn.out, decoded, &dn.len, &bad);
      if (rc == S3S_E_CHECKSUM) throw SparkException("Invalid checksum detected for " + rq.id.name());  // S3ChecksumValidationStream.scala:72-74
      if (rc == S3S_E_BAD_FRAME) throw IOException("Stream is corrupted");
      if (rc != S3S_OK) throw IOException(std::string("s3s_decompress_range: ") + s3s_last_error(ctx));
    } catch (const SparkException& e) {
      dn.errKind = 1;
      dn.err = e.what();
    } catch (const std::exception& e) {
      dn.errKind = 2;
      dn.err = e.what();
    }
    comp_.release(f.comp);
    if (dn.errKind) {
      dec_.release(dn.out);
      dn.out = nullptr;
    }
    std::lock_guard<std::mutex> lk(mu_);
    if (stop_) {
      dec_.release(dn.out);
      break;
    }
    stats_.compressedBytes += rq.maxBytes;
    stats_.decodedBytes += dn.len;
    done_.push_back(std::move(dn));
    cvDone_.notify_all();
  }
  for (auto& c : ctxs) releaseContext(d_, c.first, c.second);
}

bool S3BufferedPrefetchIterator::hasNext() {
  std::lock_guard<std::mutex> lk(mu_);
  return delivered_ < reqs_.size();
}

PrefetchedBlock S3BufferedPrefetchIterator::next() {
  const auto t0 = std::chrono::steady_clock::now();
  std::unique_lock<std::mutex> lk(mu_);
  if (delivered_ >= reqs_.size()) throw std::out_of_range("next on empty iterator");
  cvDone_.wait(lk, [&] { return !done_.empty(); });
  Done dn = std::move(done_.front());
  done_.pop_front();
  delivered_++;
  stats_.blocks++;
  const auto waited = std::chrono::steady_clock::now() - t0;
  stats_.secondsWaiting += std::chrono::duration<double>(waited).count();
  if (d_.conf().fetchThreadPredictor) {  // configureThreads(latency) (:77-91)
    const int want = predictor_.addMeasurementAndPredict(std::chrono::duration_cast<std::chrono::nanoseconds>(waited).count());
    if (want != desiredFetchers_) {
      desiredFetchers_ = want;
      cvPark_.notify_all();
    }
  }
  lk.unlock();
  if (dn.errKind == 1) throw SparkException(dn.err);
  if (dn.errKind == 2) throw IOException(dn.err);
  return PrefetchedBlock{reqs_[dn.req].id, dn.out, dn.len};
}

void S3BufferedPrefetchIterator::release(PrefetchedBlock& b) {
  dec_.release(const_cast<uint8_t*>(b.data));
  b.data = nullptr;
  b.size = 0;
}

S3BufferedPrefetchIterator::Stats S3BufferedPrefetchIterator::stats() const {
  Stats s;
  {
    std::lock_guard<std::mutex> lk(mu_);
    s = stats_;
  }
  s.compHighWater = comp_.highWater();
  s.decodedHighWater = dec_.highWater();
  {
    std::lock_guard<std::mutex> lk(mu_);
    s.fetchThreads = desiredFetchers_;
  }
  return s;
}

}  // namespace s3shuffle

// ---- self-test of the staging pool (tests/test_host_mirror.py; runs on a CPU-only box too, where staging is
// plain memory) ------------------------------------------------------------------------------------------------
// the tuner alone: feeds `n` consumer wait times and returns the predicted thread count after each (tests/test_host_mirror.py
// compares it with a line-by-line Python restatement of the Scala class)
extern "C" void s3sh_thread_predictor_run(int maxThreads, const long long* latenciesNs, int n, int* out) {
  s3shuffle::ThreadPredictor p(maxThreads);
  for (int i = 0; i < n; i++) out[i] = p.addMeasurementAndPredict((int64_t)latenciesNs[i]);
}

extern "C" int s3sh_pool_selftest() {
  using namespace s3shuffle;
  try {
    // 1. budget accounting, reuse, high-water mark
    {
      PinnedPool pool(8 << 20);
      uint8_t* a = pool.acquire(3 << 20);  // rounds to 4 MiB
      uint8_t* b = pool.acquire(1 << 20);
      if (!a || !b || pool.inUse() != (5 << 20)) return 1;
      memset(a, 0xAB, 3 << 20);
      pool.release(a);
      uint8_t* c = pool.acquire(4 << 20);  // the released buffer comes back
      if (pool.inUse() != (5 << 20) || pool.highWater() != (5 << 20)) return 2;
      pool.release(b);
      pool.release(c);
      pool.release(c);  // double release is ignored
      if (pool.inUse() != 0) return 3;
    }
    // 2. a waiter is released when memory comes back; a request larger than the budget runs alone
    {
      PinnedPool pool(2 << 20);
      uint8_t* a = pool.acquire(2 << 20);
      std::atomic<int> stage{0};
      std::thread t([&] {
        uint8_t* big = pool.acquire(16 << 20);  // > budget: waits until nothing else is in use
        stage = 1;
        pool.release(big);
      });
      std::this_thread::sleep_for(std::chrono::milliseconds(50));
      if (stage != 0) return 4;  // must still be waiting
      pool.release(a);
      t.join();
      if (stage != 1 || pool.inUse() != 0) return 5;
    }
    // 3. cancel wakes a waiter with an exception
    {
      PinnedPool pool(1 << 20);
      uint8_t* a = pool.acquire(1 << 20);
      std::atomic<int> threw{0};
      std::thread t([&] {
        try {
          pool.acquire(1 << 20);
        } catch (const IOException&) {
          threw = 1;
        }
      });
      std::this_thread::sleep_for(std::chrono::milliseconds(50));
      pool.cancel();
      t.join();
      pool.release(a);
      if (!threw) return 6;
    }
    // 4. the non-blocking process pool never waits
    {
      PinnedPool pool(1 << 20, /*blocking=*/false);
      uint8_t* a = pool.acquire(4 << 20);
      uint8_t* b = pool.acquire(4 << 20);
      if (!a || !b) return 7;
      pool.release(a);
      pool.release(b);
    }
    releasePinnedCache();
    return 0;
  } catch (const std::exception&) {
    return 100;
  }
}
