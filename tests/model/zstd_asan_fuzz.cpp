// Memory-safety fuzz of the product's Zstandard decoder core (spark-s3-shuffle_amd/csrc/zstd_decode_core.h), host build under
// ASan / UBSan: TEST INFRASTRUCTURE (tests/tools/zstd_asan_fuzz.py builds and runs it; tests/test_zstd_model.py runs a short
// campaign).  Seed streams come from libzstd (written by the Python side); every mutation is decoded from a heap copy of
// EXACTLY its size into a destination of EXACTLY the declared capacity, so a read or write one byte outside either — which
// the guard-byte tests cannot see for reads — stops the run with a sanitizer report.  Results are not compared here (that is
// test_mutated_streams_behave_like_libzstd); the only outcomes are "refused", "decoded" and "sanitizer report".
//   usage: zstd_asan_fuzz <seed file> <rng seed> <seconds> [max mutations]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <random>
#include <vector>

#include "../../spark-s3-shuffle_amd/csrc/zstd_decode_core.h"

using namespace s3s_zstd;

struct Seed {
  std::vector<uint8_t> comp;
  int64_t size;
};

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 2;
  std::vector<Seed> seeds;
  for (;;) {
    uint32_t n = 0, usize = 0;
    if (fread(&n, 4, 1, f) != 1 || fread(&usize, 4, 1, f) != 1) break;
    Seed s;
    s.comp.resize(n);
    s.size = usize;
    if (n && fread(s.comp.data(), 1, n, f) != n) return 2;
    seeds.push_back(std::move(s));
  }
  fclose(f);
  if (seeds.empty()) return 2;
  std::mt19937_64 rng(strtoull(argv[2], nullptr, 10));
  const double seconds = atof(argv[3]);
  const long max_mut = argc > 4 ? atol(argv[4]) : -1;
  auto R = [&](uint64_t n) { return n ? rng() % n : 0; };
  static Work w_size, w_dec;
  static LitPipe lp_size, lp_dec;
  std::vector<uint8_t> lit(kMaxBlock + 64);
  long n_mut = 0, n_ok = 0, n_refused = 0, n_same = 0;
  const auto t0 = std::chrono::steady_clock::now();
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds && (max_mut < 0 || n_mut < max_mut)) {
    const Seed& s = seeds[R(seeds.size())];
    std::vector<uint8_t> m = s.comp;
    const int kind = (int)R(8);
    const int reps = 1 + (int)R(3);
    for (int r = 0; r < reps && !m.empty(); r++) switch (kind) {
        case 0: m[R(m.size())] ^= (uint8_t)(1u << R(8)); break;                                    // bit flip
        case 1: m.resize(1 + R(m.size())); break;                                                   // truncation
        case 2: { const size_t i = R(m.size()); for (size_t k = i; k < m.size() && k < i + 4; k++) m[k] = (uint8_t)rng(); } break;  // 4 random bytes
        case 3: { const size_t i = R(m.size() < 64 ? m.size() : 64); m[i] = (uint8_t)rng(); } break;  // a header byte
        case 4: { const Seed& o = seeds[R(seeds.size())]; const size_t a = R(m.size()), b = R(o.comp.size());    // splice of two streams
                  m.resize(a); m.insert(m.end(), o.comp.begin() + (long)b, o.comp.end()); } break;
        case 5: { const size_t i = R(m.size()); m[i] = (uint8_t)(R(2) ? 0xFF : 0x00); } break;     // extreme byte
        case 6: { const size_t i = R(m.size()), n = 1 + R(16); m.insert(m.begin() + (long)i, n, (uint8_t)rng()); } break;  // inserted run
        default: { const size_t i = R(m.size()), n = 1 + R(16); m.erase(m.begin() + (long)i, m.begin() + (long)(i + n < m.size() ? i + n : m.size())); } break;  // deleted run
      }
    n_mut++;
    // exact-size heap copies: the sanitizer's red zones start at the first byte outside
    std::unique_ptr<uint8_t[]> comp(new uint8_t[m.size() ? m.size() : 1]);
    memcpy(comp.get(), m.data(), m.size());
    const int64_t cap = kind == 1 || R(4) ? s.size + (int64_t)R(3) * 4096 : (int64_t)R((uint64_t)s.size + 1);  // sometimes too small
    std::unique_ptr<uint8_t[]> dst(new uint8_t[cap ? cap : 1]);
    Lanes L{0, 1};
    int64_t total = -1, total2 = -1;
    const int rc0 = decode_partition(w_size, lp_size, comp.get(), (int64_t)m.size(), nullptr, 0, false, nullptr, 0, L, &total);
    if (rc0 != 0) { n_refused++; continue; }
    const int rc = decode_partition(w_dec, lp_dec, comp.get(), (int64_t)m.size(), dst.get(), cap, true, lit.data(), 0, L, &total2);
    if (rc != 0) { n_refused++; continue; }
    if (total2 != total || total2 > cap) { printf("size pass %lld, decode pass %lld, capacity %lld\n", (long long)total, (long long)total2, (long long)cap); return 1; }
    n_ok++;
    n_same += m == s.comp;
  }
  printf("zstd_asan_fuzz: %ld mutations, %ld refused, %ld decoded (%ld unchanged)\n", n_mut, n_refused, n_ok, n_same);
  return 0;
}
