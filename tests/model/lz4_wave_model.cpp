// lz4_wave_model.cpp — CPU lockstep model of the wave64 LZ4 block compressor.
//
// TEST INFRASTRUCTURE.  This is NOT the product path and is not the oracle either: it is a
// lane-by-lane simulation of the batching algorithm that
// spark-s3-shuffle_amd/csrc/lz4_compress.hip runs on one wavefront, so the algorithm (batch
// schedule, duplicate-hash cut, table rollback, cooperative extension, emission) can be
// checked bit-for-bit against the oracle / liblz4 on the CPU-only build box, including under
// an ADVERSARIAL choice of which lane wins a same-address LDS store (the hardware does not
// define it).  tests/test_wave_model.py drives it.
//
// Structure mirrors the kernel one "vector instruction" at a time: every per-lane value is a
// 64-entry array, every LDS store of a batch is applied in a caller-chosen lane order.
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

constexpr int WAVE = 64;
constexpr int MFLIMIT = 12, LASTLITERALS = 5, MINMATCH = 4;

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
inline uint32_t hash13(uint32_t v) { return (v * 2654435761u) >> 19; }

// cumulative step schedule: S(t) = sum_{tau<t} step_tau, step_0 = step_1 = 1,
// step_tau = (62+tau)>>6 for tau >= 2   (LZ4 skip acceleration, searchMatchNb from 64)
inline int sched_F(int X) {
  int q = X >> 6, r = X & 63;
  return 32 * q * (q - 1) + q * r;
}
inline int sched_S(int t) { return t < 2 ? t : 2 + sched_F(62 + t); }

struct Out {
  uint8_t* dst;
  int cap;  // == chunk length: anything longer is stored RAW by the framing layer
  int op = 0;
  bool overflow = false;
};

// emit token + literal-length bytes + literals (+ optionally offset & match length)
void emit_sequence(Out& o, const uint8_t* in, int anchor, int lit, bool has_match, int offset,
                   int mcode) {
  int need = 1 + lit + (lit >= 15 ? (lit - 15) / 255 + 1 : 0);
  if (has_match) need += 2 + (mcode >= 15 ? (mcode - 15) / 255 + 1 : 0);
  if (o.op + need > o.cap) {
    o.overflow = true;
    return;
  }
  uint8_t* d = o.dst;
  int tok = o.op++;
  uint8_t token = (uint8_t)((lit >= 15 ? 15 : lit) << 4);
  if (lit >= 15) {
    int l = lit - 15;
    for (; l >= 255; l -= 255) d[o.op++] = 255;
    d[o.op++] = (uint8_t)l;
  }
  memcpy(d + o.op, in + anchor, (size_t)lit);
  o.op += lit;
  if (has_match) {
    d[o.op++] = (uint8_t)offset;
    d[o.op++] = (uint8_t)(offset >> 8);
    if (mcode >= 15) {
      token |= 15;
      int m = mcode - 15;
      for (; m >= 255; m -= 255) d[o.op++] = 255;
      d[o.op++] = (uint8_t)m;
    } else {
      token |= (uint8_t)mcode;
    }
  }
  d[tok] = token;
}

}  // namespace

// winner_mode: 0 = highest lane wins a same-address store, 1 = lowest lane wins,
//              2 = pseudo-random lane order (seeded)
// Returns compressed size, or -1 if the output would exceed `len` (framing stores RAW).
// stats[0] += batches, stats[1] += cut-restarts, stats[2] += sequences
extern "C" int lz4_wave_model_compress(const uint8_t* src, int len, uint8_t* dst, int winner_mode,
                                       uint64_t seed, int64_t* stats) {
  std::vector<uint8_t> lds_in((size_t)len + 512, 0);  // padded: wide compares may over-read
  memcpy(lds_in.data(), src, (size_t)len);
  const uint8_t* in = lds_in.data();
  std::vector<uint16_t> T(8192, 0);
  Rng rng{seed};
  Out o{dst, len};
  int64_t nb_batches = 0, nb_cuts = 0, nb_seq = 0;

  const int mflimit_plus_one = len - MFLIMIT + 1;
  const int matchlimit = len - LASTLITERALS;
  int anchor = 0;

  if (len >= MFLIMIT + 1) {
    T[hash13(rd32(in))] = 0;
    int base = 1, t0 = 1;
    for (;;) {
      nb_batches++;
      // ---- per-lane schedule --------------------------------------------------------------
      int pos[WAVE], valid[WAVE];
      uint32_t v[WAVE], h[WAVE], w[WAVE];
      uint16_t c[WAVE], r[WAVE];
      const int S0 = sched_S(t0);
      int nvalid = 0;
      for (int i = 0; i < WAVE; i++) {
        int t = t0 + i;
        pos[i] = base + sched_S(t) - S0;
        int nextpos = base + sched_S(t + 1) - S0;
        valid[i] = (t == 0) || (nextpos <= mflimit_plus_one);
      }
      for (int i = 0; i < WAVE; i++) {
        if (!valid[i]) break;
        nvalid++;
      }
      for (int i = nvalid; i < WAVE; i++) valid[i] = 0;  // monotone
      // ---- v, hash, old candidate -----------------------------------------------------------
      for (int i = 0; i < nvalid; i++) {
        v[i] = rd32(in + pos[i]);
        h[i] = hash13(v[i]);
      }
      for (int i = 0; i < nvalid; i++) c[i] = T[h[i]];
      // ---- speculative insert (one ds_write_b16, undefined winner) + readback ---------------
      {
        int order[WAVE];
        for (int i = 0; i < nvalid; i++) order[i] = i;
        if (winner_mode == 1) {
          for (int i = 0; i < nvalid; i++) order[i] = nvalid - 1 - i;
        } else if (winner_mode == 2) {
          for (int i = nvalid - 1; i > 0; i--) {
            int j = (int)(rng.next() % (uint32_t)(i + 1));
            int tmp = order[i];
            order[i] = order[j];
            order[j] = tmp;
          }
        }
        for (int k = 0; k < nvalid; k++) T[h[order[k]]] = (uint16_t)pos[order[k]];
      }
      uint64_t C = 0, M = 0;
      for (int i = 0; i < nvalid; i++) {
        r[i] = T[h[i]];
        w[i] = rd32(in + c[i]);
        if (r[i] != (uint16_t)pos[i]) C |= 1ull << i;
        if (w[i] == v[i]) M |= 1ull << i;
      }
      // ---- clean prefix: lanes whose start-of-batch candidate is the true one ------------------
      // A lane is "clean" when no earlier lane of the batch shares its hash.  Every lane below the
      // smallest loser c0 is clean; c0 itself is clean iff its slot's winner is a LATER lane (an
      // earlier member of its group would have lost too, contradicting minimality).
      uint64_t A = 0;  // lane i repeats lane i-1's 4 bytes: true candidate = pos[i-1], and it matches
      for (int i = 1; i < nvalid; i++)
        if (v[i] == v[i - 1]) A |= 1ull << i;
      int B = WAVE, c0 = -1;
      bool clean0 = false;
      if (C) {
        c0 = __builtin_ctzll(C);
        clean0 = r[c0] > (uint16_t)pos[c0];
        B = c0 + (clean0 ? 1 : 0);
      }
      int lim = B < nvalid ? B : nvalid;
      uint64_t Mv = lim >= 64 ? M : (M & ((1ull << lim) - 1));
      int keep;  // lanes [0,keep) stay inserted
      int m = -1;
      bool adj = false;
      if (Mv) {
        m = __builtin_ctzll(Mv);
        keep = m + 1;
      } else if (lim < nvalid && ((A >> lim) & 1)) {
        m = lim;  // first non-clean lane repeats its (clean) predecessor: a certain match
        adj = true;
        keep = lim + 1;
      } else {
        keep = lim;
      }
      // ---- table fix-up: (1) winners never reached restore the old entry; (2) committed losers
      //      re-insert (c0 unless the adjacent match lane right after it overrides the same slot)
      for (int i = keep; i < nvalid; i++)
        if (r[i] == (uint16_t)pos[i]) T[h[i]] = c[i];
      if (clean0 && c0 < keep && !(adj && c0 == m - 1)) T[h[c0]] = (uint16_t)pos[c0];
      if (adj) T[h[m]] = (uint16_t)pos[m];

      if (m < 0) {
        if (lim == nvalid && nvalid < WAVE) break;  // search loop ran into mflimit: last literals
        if (lim < nvalid) nb_cuts++;
        // continue the same no-match run at lane `lim`
        base = pos[lim - 1] + (sched_S(t0 + lim) - sched_S(t0 + lim - 1));
        t0 += lim;
        continue;
      }

      // ---- match at lane m -------------------------------------------------------------------
      nb_seq++;
      int ip = pos[m], match = adj ? pos[m - 1] : c[m];
      // catch-up (backward extension), 64 bytes per round
      for (;;) {
        int maxback = ip - anchor < match ? ip - anchor : match;
        if (maxback <= 0) break;
        int round = maxback < WAVE ? maxback : WAVE;
        uint64_t E = 0;
        for (int k = 0; k < round; k++)
          if (in[ip - 1 - k] == in[match - 1 - k]) E |= 1ull << k;
        int nbk = (~E == 0) ? 64 : __builtin_ctzll(~E);
        if (nbk > round) nbk = round;
        ip -= nbk;
        match -= nbk;
        if (nbk < WAVE) break;
      }
      // zero or more matches in a row (the "test next position" path re-enters here)
      for (;;) {
        // forward extension: 256 bytes per round, limited by matchlimit
        int count = 0;
        for (;;) {
          int avail = matchlimit - (ip + MINMATCH + count);
          if (avail <= 0) break;
          uint64_t D = 0;
          int first_byte[WAVE];
          for (int k = 0; k < WAVE; k++) {
            uint32_t x = rd32(in + ip + MINMATCH + count + 4 * k) ^
                         rd32(in + match + MINMATCH + count + 4 * k);
            first_byte[k] = x ? (__builtin_ctz(x) >> 3) : 4;
            if (x) D |= 1ull << k;
          }
          int got = D ? 4 * __builtin_ctzll(D) + first_byte[__builtin_ctzll(D)] : 256;
          if (got > avail) got = avail;
          count += got;
          if (got < 256) break;
        }
        emit_sequence(o, in, anchor, ip - anchor, true, ip - match, count);
        if (o.overflow) return -1;
        ip += MINMATCH + count;
        anchor = ip;
        if (ip >= mflimit_plus_one) goto last_literals;
        T[hash13(rd32(in + ip - 2))] = (uint16_t)(ip - 2);
        // model choice: the post-match probe is lane 0 (t = 0) of the next batch
        base = ip;
        t0 = 0;
        break;
      }
    }
  }
last_literals:
  emit_sequence(o, in, anchor, len - anchor, false, 0, 0);
  if (o.overflow) return -1;
  if (stats) {
    stats[0] += nb_batches;
    stats[1] += nb_cuts;
    stats[2] += nb_seq;
  }
  return o.op;
}
