// Race / memory check of the C++ host mirror (spark-s3-shuffle_amd/host/*.cpp: map output writers, the page-locked pools, the
// context cache, the prefetch pipeline with its fetch / decode threads and the ThreadPredictor) — TEST INFRASTRUCTURE.
// tests/test_host_race.py compiles the host sources together with this driver and tests/mock_jni/fake_codec.c (a toy
// stand-in for the codec library, so no GPU is needed) once under ThreadSanitizer and once under ASan / UBSan and runs it:
// several task threads write map outputs at the same time, several read through the pipeline at the same time with
// budgets small enough that fetchers block on them, iterators are abandoned half-way, a damaged object raises in the
// consumer.  Any report of a sanitizer fails the test; so does a block that differs from what was written.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <thread>

#include "../../spark-s3-shuffle_amd/host/s3shuffle_host.h"

using namespace s3shuffle;

#define CHECK(x)                                                          \
  do {                                                                    \
    if (!(x)) {                                                           \
      printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #x);               \
      fflush(stdout);                                                     \
      std::_Exit(1);                                                      \
    }                                                                     \
  } while (0)

static constexpr int kMaps = 12, kParts = 16;

static std::vector<uint8_t> partition_bytes(int shuffle, int64_t map, int part) {
  uint32_t s = (uint32_t)(shuffle * 7919 + map * 104729 + part * 31 + 1);
  size_t n = (part % 5 == 3) ? 0 : 1000 + (s % 60000);  // every fifth partition is empty
  if (part == 7) n = 3 << 20;                             // one whose stream is larger than the whole fetch budget below (served alone)
  std::vector<uint8_t> v(n);
  for (size_t i = 0; i < n; i++) {
    s = s * 1664525u + 1013904223u;
    v[i] = (uint8_t)((s >> 24) % 23 + 'a');
  }
  return v;
}

static void write_map(const S3ShuffleDispatcher& d, int shuffle, int64_t map) {
  S3ShuffleMapOutputWriter w(d, shuffle, map, kParts);
  for (int p = 0; p < kParts; p++) {
    const std::vector<uint8_t> b = partition_bytes(shuffle, map, p);
    if (b.empty() && (map & 1)) continue;  // an empty partition may also never be opened
    w.getPartitionWriter(p);
    const size_t half = b.size() / 2;
    w.write(b.data(), half);
    if (map % 3 == 0 && half) w.markSegment();  // a multi-spill merge: two pieces
    w.write(b.data() + half, b.size() - half);
    w.closePartition();
  }
  const std::vector<int64_t> lengths = w.commitAllPartitions();
  CHECK((int)lengths.size() == kParts);
}

static void check_blocks(const std::vector<FetchedBlock>& blocks, int shuffle, int r0, int r1, bool batch) {
  size_t k = 0;  // empty blocks are filtered (S3ShuffleReader.scala:91-93): the list holds the non-empty ones in map order
  for (int64_t m = 0; m < kMaps; m++) {
    std::vector<uint8_t> want;
    for (int p = r0; p < r1; p++) {
      const std::vector<uint8_t> b = partition_bytes(shuffle, m, p);
      if (batch) {
        want.insert(want.end(), b.begin(), b.end());
      } else if (!b.empty()) {
        CHECK(k < blocks.size() && blocks[k].id.mapId == m && blocks[k].id.reduceId == p);
        CHECK(blocks[k].bytes == b);
        k++;
      }
    }
    if (batch && !want.empty()) {
      CHECK(k < blocks.size() && blocks[k].id.mapId == m && blocks[k].bytes == want);
      k++;
    }
  }
  CHECK(k == blocks.size());
}

int main(int argc, char** argv) {
  CHECK(argc > 1);
  Conf conf;
  conf.rootDir = std::string(argv[1]) + "/";
  conf.folderPrefixes = 3;
  conf.maxBufferSizeTask = 2 << 20;            // two blocks in flight (a buffer counts at least 1 MiB): fetchers wait for the budget
  conf.maxConcurrencyTask = 6;
  conf.gpuDecodeThreads = 3;
  conf.gpuMaxDecodedBufferSizeTask = 4 << 20;  // four decoded blocks: the consumer below never holds more than three
  S3ShuffleDispatcher d(conf);
  S3ShuffleDispatcher dp(conf);
  dp.setFetchThreadPredictor(true);

  // ---- map side: four task threads, two shuffles ------------------------------------------------------------------------
  {
    std::vector<std::thread> ts;
    std::atomic<int> next{0};
    for (int t = 0; t < 4; t++)
      ts.emplace_back([&] {
        for (int i; (i = next.fetch_add(1)) < 2 * kMaps;) write_map(d, i / kMaps, i % kMaps);
      });
    for (auto& t : ts) t.join();
  }
  CHECK((int)d.listShuffleIndices(0).size() == kMaps && (int)d.listShuffleIndices(1).size() == kMaps);

  // ---- reduce side: concurrent readers on the pipeline, single blocks and batches, predictor on and off ------------------
  {
    std::vector<std::thread> ts;
    for (int t = 0; t < 4; t++)
      ts.emplace_back([&, t] {
        const int shuffle = t & 1, r0 = (t * 3) % 8, r1 = r0 + 6;
        const bool batch = t >= 2;
        S3ShuffleReader rd(t == 3 ? dp : d, shuffle, r0, r1, batch);
        check_blocks(rd.read(), shuffle, r0, r1, batch);
        check_blocks(rd.readSequential(), shuffle, r0, r1, batch);
      });
    for (auto& t : ts) t.join();
  }

  // ---- an iterator abandoned with blocks in flight, and one whose consumer holds several blocks before releasing ---------
  for (int round = 0; round < 6; round++) {
    S3ShuffleReader rd(round & 1 ? dp : d, 0, 0, kParts, false);
    S3BufferedPrefetchIterator it(round & 1 ? dp : d, rd.blockRequests());
    std::vector<PrefetchedBlock> held;
    for (int k = 0; k < 1 + round % 3 && it.hasNext(); k++) held.push_back(it.next());
    for (size_t k = 0; k + 1 < held.size(); k++) it.release(held[k]);  // (the last one is still out when the iterator dies)
    (void)it.stats();
  }

  // ---- a damaged object: the consumer gets the block's exception, the pipeline shuts down cleanly -------------------------
  {
    const std::string path = d.getPath(BlockId::ShuffleDataBlockId(1, 5));
    std::fstream f(path, std::ios::in | std::ios::out | std::ios::binary);
    CHECK(f.good());
    f.seekp(20);
    const char x = 0x7f;
    f.write(&x, 1);
    f.close();
    S3ShuffleReader rd(d, 1, 0, kParts, true);
    bool threw = false;
    try {
      (void)rd.read();
    } catch (const SparkException&) {
      threw = true;
    } catch (const IOException&) {
      threw = true;
    }
    CHECK(threw);
  }
  d.removeShuffle(0);
  d.removeShuffle(1);
  releaseContextCache();
  releasePinnedCache();
  printf("host_race ok\n");
  return 0;
}
