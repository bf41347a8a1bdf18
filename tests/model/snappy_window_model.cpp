// snappy_window_model.cpp — CPU lock-step model of the exact-window wave64 Snappy fragment
// compressor (spark-s3-shuffle_amd/csrc/snappy_compress.hip, window path + general batch).
// TEST INFRASTRUCTURE (not the product path, not the oracle): checks the algorithm bit-for-bit
// against the oracle / libsnappy 1.1.8 on the CPU-only box, under an adversarial choice of which
// lane wins a same-address LDS store, and reports how the work splits between the two paths.
//
// Window path: the fragment is cut into aligned 64-byte windows, lane i <-> position 64k+i.
//   cp   = T[h]                 table candidate of every lane (nothing of the window is in T yet)
//   grp  = lane shares its hash with another live lane (two speculative store passes, rolled back)
//   em   = bytes at cp match, with the forward match length
//   The probes a run makes inside the window are a fixed pattern of the distance d to the run's
//   base (d <= 33 every byte, 35..65 every 2nd, 68..98 every 3rd), so the runs of a window are
//   resolved with mask arithmetic only: first event lane of the run; a grp lane's true candidate
//   is the highest kept lane below it with the same hash, else cp.  K collects what the sequential
//   code inserts (probes, and ip-1 after every copy); the highest kept lane of each hash commits.
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

namespace {
// probe index of the position d bytes behind a run's base, or -1 if the run does not probe it
// (closed form of Q^-1 for d <= 98; checked against the schedule in init)
inline int probe_u(int d) {
  if (d <= 33) return d;
  if (d <= 65) return (d & 1) ? 34 + ((d - 35) >> 1) : -1;
  if (d <= 98) return ((d - 68) % 3 == 0 && d >= 68) ? 50 + (d - 68) / 3 : -1;
  return -2;
}
constexpr int kMaxD = 98;
}  // namespace

// stats: [0] general batches [1] windows [2] copies found in windows [3] copies found by batches
//        [4] in-window candidates [5] capped lengths (cooperative extension) [6] suspect events
extern "C" int snappy_window_model_compress(const uint8_t* src, int len, uint8_t* dst, int winner_mode,
                                            uint64_t seed, int use_windows, int64_t* stats) {
  static bool inited = false;
  if (!inited) {
    init_sched();
    for (int u = 0; u <= 60; u++)
      if (probe_u(Q(u)) != u) __builtin_trap();
    for (int d = 0, u = 0; d <= kMaxD; d++) {
      if (Q(u) == d) {
        u++;
      } else if (probe_u(d) != -1) {
        __builtin_trap();
      }
    }
    inited = true;
  }
  std::vector<uint8_t> pad((size_t)len + 512, 0);
  memcpy(pad.data(), src, (size_t)len);
  const uint8_t* in = pad.data();
  uint8_t* op = dst;
  uint32_t vl = (uint32_t)len;
  while (vl >= 0x80) {
    *op++ = (uint8_t)(vl | 0x80);
    vl >>= 7;
  }
  *op++ = (uint8_t)vl;
  if (len == 0) return (int)(op - dst);
  int tsize = 256;
  while (tsize < 16384 && tsize < len) tsize <<= 1;
  int lg = 0;
  while ((1 << lg) < tsize) lg++;
  const int shift = 32 - lg;
  std::vector<uint16_t> T((size_t)tsize, 0);
  Rng rng{seed};
  auto shuffled = [&](int* order, int n) {
    for (int i = 0; i < n; i++) order[i] = i;
    if (winner_mode == 1)
      for (int i = 0; i < n; i++) order[i] = n - 1 - i;
    else if (winner_mode == 2)
      for (int i = n - 1; i > 0; i--) {
        int j = (int)(rng.next() % (uint32_t)(i + 1));
        int t = order[i];
        order[i] = order[j];
        order[j] = t;
      }
  };
  auto find_match_length = [&](int a, int b) {  // bytes equal from in+a / in+b, b < len side limit
    int n = 0;
    while (b + n < len && in[a + n] == in[b + n]) n++;
    return n;
  };
  int next_emit = 0;
  int64_t st[8] = {0};
  if (len >= 15) {
    const int ip_limit = len - 15;
    const int fast_limit = len - 192;
    int rbase = 0, u0 = 1;
    for (;;) {
      const int pos0 = rbase + Q(u0);
      const int wbase = pos0 & ~63;
      if (use_windows && u0 <= 60 && wbase + 63 - rbase <= kMaxD && wbase <= fast_limit) {
        // ======================= exact window =====================================================
        st[1]++;
        const int rs0 = pos0 - wbase;
        uint32_t v[WAVE], h[WAVE];
        uint16_t cp[WAVE];
        bool grp[WAVE], em[WAVE];
        for (int i = 0; i < WAVE; i++) {
          v[i] = rd32(in + wbase + i);
          h[i] = (v[i] * 0x1e35a7bdu) >> shift;
          cp[i] = T[h[i]];
          grp[i] = false;
        }
        {  // duplicate-hash groups among the live lanes: two speculative store passes, rolled back
          int order[WAVE];
          const int nlive = WAVE - rs0;
          shuffled(order, nlive);
          for (int k = 0; k < nlive; k++) T[h[rs0 + order[k]]] = (uint16_t)(wbase + rs0 + order[k]);
          bool lost1[WAVE] = {false};
          for (int i = rs0; i < WAVE; i++) lost1[i] = T[h[i]] != (uint16_t)(wbase + i);
          shuffled(order, nlive);
          for (int k = 0; k < nlive; k++)
            if (lost1[rs0 + order[k]]) T[h[rs0 + order[k]]] = (uint16_t)(wbase + rs0 + order[k]);
          bool own[WAVE] = {false};
          for (int i = rs0; i < WAVE; i++) {
            own[i] = T[h[i]] == (uint16_t)(wbase + i);
            grp[i] = lost1[i] || !own[i];
          }
          for (int i = rs0; i < WAVE; i++)
            if (own[i]) T[h[i]] = cp[i];
        }
        uint64_t Ecp = 0, Dp = 0, Hsame[WAVE];
        for (int i = 0; i < WAVE; i++) {
          em[i] = i >= rs0 && rd32(in + cp[i]) == v[i];
          if (em[i]) Ecp |= 1ull << i;
          if (grp[i]) Dp |= 1ull << i;
          Hsame[i] = 0;
          for (int j = 0; j < WAVE; j++)
            if (h[j] == h[i]) Hsame[i] |= 1ull << j;
        }
        uint64_t ED = Ecp | Dp;
        // probe pattern of the run that enters the window
        uint64_t runmask = 0;
        for (int i = rs0; i < WAVE; i++)
          if (probe_u(wbase + i - rbase) >= 0) runmask |= 1ull << i;
        const uint64_t PM = 0xAAAAAAAA00000000ull | ((1ull << 34) - 1ull);  // d: 0..33, 35,37,..,63
        uint64_t K = 0;
        int rs = rs0, rt = u0, pend_q = -1;
        bool to_remainder = false;
        for (;;) {
          const uint64_t cm = ED & runmask;
          if (cm == 0) {
            K |= runmask;
            u0 = rt + __builtin_popcountll(runmask);
            break;
          }
          const int m = __builtin_ctzll(cm);
          const uint64_t bit = 1ull << m;
          int cand = cp[m];
          bool is_match = (Ecp & bit) != 0;
          if (Dp & bit) {
            st[6]++;
            const uint64_t dk = Hsame[m] & (bit - 1) & (K | runmask);
            if (dk) {
              const int d = 63 - __builtin_clzll(dk);
              is_match = v[d] == v[m];
              cand = wbase + d;
              if (is_match) st[4]++;
            }
          }
          if (!is_match) {
            ED &= ~bit;
            continue;
          }
          st[2]++;
          K |= runmask & ((bit << 1) - 1);
          const int ip0 = wbase + m;
          const int extra = find_match_length(cand + 4, ip0 + 4);
          if (extra >= 64) st[5]++;
          if (ip0 > next_emit) op = emit_literal(op, in + next_emit, ip0 - next_emit);
          op = emit_copy(op, ip0 - cand, 4 + extra, extra < 8);
          const int ipe = ip0 + 4 + extra;
          next_emit = ipe;
          if (ipe >= ip_limit) {
            to_remainder = true;
            break;
          }
          const int q = ipe - 1 - wbase;
          if (q < WAVE) K |= 1ull << q;
          else pend_q = q + wbase;
          rbase = ipe;
          u0 = 0;
          if (ipe >= wbase + WAVE) break;
          rs = ipe - wbase;
          rt = 0;
          runmask = PM << rs;
        }
        if (to_remainder) break;
        // commit: the highest kept lane of every hash writes
        for (int i = 0; i < WAVE; i++)
          if ((K >> i) & 1) {
            const uint64_t above = i == 63 ? 0 : (Hsame[i] & K & ~((2ull << i) - 1));
            if (!above) T[h[i]] = (uint16_t)(wbase + i);
          }
        if (pend_q >= 0) T[(rd32(in + pend_q) * 0x1e35a7bdu) >> shift] = (uint16_t)pend_q;
        continue;
      }
      // ======================= general batch ========================================================
      st[0]++;
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
        shuffled(order, nvalid);
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
      st[3]++;
      const int ip0 = pos[m], cand = adj ? pos[m - 1] : c[m];
      const int extra = find_match_length(cand + 4, ip0 + 4);
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
  if (stats)
    for (int i = 0; i < 8; i++) stats[i] += st[i];
  return (int)(op - dst);
}
