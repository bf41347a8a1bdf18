// snappy_wave_model.cpp — CPU lock-step model of the wave64 Snappy fragment compressor
// (spark-s3-shuffle_amd/csrc/snappy_compress.hip).  TEST INFRASTRUCTURE: checks the batching
// algorithm bit-for-bit against the oracle / libsnappy 1.1.8 on the CPU-only box, under an
// adversarial choice of which lane wins a same-address LDS store.
#include <cstdint>
#include <cstring>
#include <vector>

namespace {
constexpr int WAVE = 64;
struct Rng {
  uint64_t s;
  uint32_t next() {
    s += 0x9E3779B97F4A7C15ull;
    uint64_t z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return (uint32_t)((z ^ (z >> 31)) >> 16);
  }
};
inline uint32_t rd32(const uint8_t* p) {
  uint32_t v;
  memcpy(&v, p, 4);
  return v;
}
int sched[272];
void init_sched() {
  uint32_t skip = 32;
  int pos = 0;
  for (int t = 0; t < 272; t++) {
    sched[t] = pos;
    uint32_t step = skip >> 5;
    skip += step;
    pos += (int)step;
  }
}
inline int Q(int u) {
  if (u <= 33) return u;
  int t = u - 1;
  return t < 272 ? 1 + sched[t] : (1 << 20);
}
uint8_t* emit_literal(uint8_t* op, const uint8_t* lit, int len) {
  int n = len - 1;
  if (n < 60) *op++ = (uint8_t)(n << 2);
  else {
    int count = n < 256 ? 1 : 2;
    *op++ = (uint8_t)((59 + count) << 2);
    for (int i = 0; i < count; i++) *op++ = (uint8_t)(n >> (8 * i));
  }
  memcpy(op, lit, (size_t)len);
  return op + len;
}
uint8_t* copy64(uint8_t* op, int offset, int len, bool lt12) {
  if (lt12 && offset < 2048) {
    *op++ = (uint8_t)(1 + ((len - 4) << 2) + ((offset >> 3) & 0xe0));
    *op++ = (uint8_t)offset;
  } else {
    *op++ = (uint8_t)(2 + ((len - 1) << 2));
    *op++ = (uint8_t)offset;
    *op++ = (uint8_t)(offset >> 8);
  }
  return op;
}
uint8_t* emit_copy(uint8_t* op, int offset, int len, bool lt12) {
  if (lt12) return copy64(op, offset, len, true);
  while (len >= 68) {
    op = copy64(op, offset, 64, false);
    len -= 64;
  }
  if (len > 64) {
    op = copy64(op, offset, 60, false);
    len -= 60;
  }
  return copy64(op, offset, len, len < 12);
}
}  // namespace

extern "C" int snappy_wave_model_compress(const uint8_t* src, int len, uint8_t* dst, int winner_mode,
                                          uint64_t seed, int64_t* stats) {
  static bool inited = false;
  if (!inited) {
    init_sched();
    inited = true;
  }
  std::vector<uint8_t> pad((size_t)len + 512, 0);
  memcpy(pad.data(), src, (size_t)len);
  const uint8_t* in = pad.data();
  uint8_t* op = dst;
  uint32_t v = (uint32_t)len;
  while (v >= 0x80) {
    *op++ = (uint8_t)(v | 0x80);
    v >>= 7;
  }
  *op++ = (uint8_t)v;
  if (len == 0) return (int)(op - dst);
  int tsize = 256;
  while (tsize < 16384 && tsize < len) tsize <<= 1;
  int lg = 0;
  while ((1 << lg) < tsize) lg++;
  const int shift = 32 - lg;
  std::vector<uint16_t> T((size_t)tsize, 0);
  Rng rng{seed};
  int next_emit = 0;
  int64_t batches = 0;
  if (len >= 15) {
    const int ip_limit = len - 15;
    int rbase = 0, u0 = 1;
    for (;;) {
      batches++;
      int nl = WAVE;
      if (u0 <= 1) nl = 34 - u0;
      int pos[WAVE];
      uint32_t vv[WAVE], h[WAVE];
      uint16_t c[WAVE], r[WAVE];
      int nvalid = 0;
      for (int i = 0; i < nl; i++) {
        int u = u0 + i;
        pos[i] = rbase + Q(u);
        int nextpos = rbase + Q(u + 1);
        if (!(u == 0 || nextpos <= ip_limit)) break;
        nvalid++;
      }
      for (int i = 0; i < nvalid; i++) {
        vv[i] = rd32(in + pos[i]);
        h[i] = (vv[i] * 0x1e35a7bdu) >> shift;
        c[i] = T[h[i]];
      }
      {
        int order[WAVE];
        for (int i = 0; i < nvalid; i++) order[i] = i;
        if (winner_mode == 1)
          for (int i = 0; i < nvalid; i++) order[i] = nvalid - 1 - i;
        else if (winner_mode == 2)
          for (int i = nvalid - 1; i > 0; i--) {
            int j = (int)(rng.next() % (uint32_t)(i + 1));
            int t = order[i];
            order[i] = order[j];
            order[j] = t;
          }
        for (int k = 0; k < nvalid; k++) T[h[order[k]]] = (uint16_t)pos[order[k]];
      }
      uint64_t L = 0, M = 0, A = 0;
      for (int i = 0; i < nvalid; i++) {
        r[i] = T[h[i]];
        if (r[i] != (uint16_t)pos[i]) L |= 1ull << i;
        if (rd32(in + c[i]) == vv[i]) M |= 1ull << i;
        if (i > 0 && vv[i] == vv[i - 1]) A |= 1ull << i;
      }
      int B = WAVE, c0 = -1;
      bool clean0 = false;
      if (L) {
        c0 = __builtin_ctzll(L);
        clean0 = r[c0] > (uint16_t)pos[c0];
        B = c0 + (clean0 ? 1 : 0);
      }
      const int lim = B < nvalid ? B : nvalid;
      const uint64_t Mv = lim >= 64 ? M : (M & ((1ull << lim) - 1));
      int m = -1, keep = lim;
      bool adj = false;
      if (Mv) {
        m = __builtin_ctzll(Mv);
        keep = m + 1;
      } else if (lim < nvalid && ((A >> lim) & 1)) {
        m = lim;
        adj = true;
        keep = lim + 1;
      }
      for (int i = keep; i < nvalid; i++)
        if (r[i] == (uint16_t)pos[i]) T[h[i]] = c[i];
      if (clean0 && c0 < keep && !(adj && c0 == m - 1)) T[h[c0]] = (uint16_t)pos[c0];
      if (adj) T[h[m]] = (uint16_t)pos[m];
      if (m < 0) {
        if (lim == nvalid && nvalid < nl) break;
        u0 += lim;
        continue;
      }
      const int ip0 = pos[m], cand = adj ? pos[m - 1] : c[m];
      // FindMatchLength the way the kernel does it: 64 dwords per round, dwords that would cross the
      // end of the chunk are read at len-4 and shifted
      int extra = 0;
      const int last4 = len - 4;
      for (;;) {
        const int avail = len - (ip0 + 4 + extra);
        if (avail <= 0) break;
        int got = 4 * WAVE;
        for (int k = 0; k < WAVE; k++) {
          const int fpi = ip0 + 4 + extra + 4 * k;
          const int fp = fpi < last4 ? fpi : last4;
          uint32_t x = rd32(in + fp) ^ rd32(in + fp - (ip0 - cand));
          const int over = fpi - fp;
          x = over >= 4 ? 0u : (x >> (8 * over));
          if (x) {
            got = 4 * k + (__builtin_ctz(x) >> 3);
            break;
          }
        }
        if (got > avail) got = avail;
        extra += got;
        if (got < 4 * WAVE) break;
      }
      if (ip0 > next_emit) op = emit_literal(op, in + next_emit, ip0 - next_emit);
      op = emit_copy(op, ip0 - cand, 4 + extra, extra < 8);
      const int ipe = ip0 + 4 + extra;
      next_emit = ipe;
      if (ipe >= ip_limit) break;
      T[(rd32(in + ipe - 1) * 0x1e35a7bdu) >> shift] = (uint16_t)(ipe - 1);
      rbase = ipe;
      u0 = 0;
    }
  }
  if (next_emit < len) op = emit_literal(op, in + next_emit, len - next_emit);
  if (stats) stats[0] += batches;
  return (int)(op - dst);
}
