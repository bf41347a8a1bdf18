// Host build of the product's Zstandard decoder core (spark-s3-shuffle_amd/csrc/zstd_decode_core.h) with one "lane":
// TEST INFRASTRUCTURE — tests/test_zstd_model.py compares it with libzstd 1.4.8 on the CPU (every block type, every
// table mode, concatenated / skippable frames, malformed input) before the same code runs on the GPU.
#include <vector>

#include "../../spark-s3-shuffle_amd/csrc/zstd_decode_core.h"

using namespace s3s_zstd;

extern "C" {
int zs_decoded_size(const uint8_t* src, int64_t size, int64_t* total) {
  static thread_local Work w;
  static thread_local LitPipe lp;
  Lanes L{0, 1};
  return decode_partition(w, lp, src, size, nullptr, 0, false, nullptr, 0, L, total);
}
int zs_decode(const uint8_t* src, int64_t size, uint8_t* dst, int64_t cap, int64_t* total) {
  static thread_local Work w;
  static thread_local LitPipe lp;  // (one thread does both sides, block by block)
  std::vector<uint8_t> lit(kMaxBlock + 64);
  Lanes L{0, 1};
  return decode_partition(w, lp, src, size, dst, cap, true, lit.data(), 0, L, total);
}
}
