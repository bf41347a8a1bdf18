// lz4_window_model.cpp — CPU lock-step model of the window-speculative wave64 LZ4 compressor
// (the algorithm of spark-s3-shuffle_amd/csrc/lz4_compress.hip, "fast window" + "general batch").
//
// TEST INFRASTRUCTURE (not the product path, not the oracle).  It simulates, one wave-wide
// operation at a time, what one wavefront does to one chunk, so the algorithm can be checked
// bit-for-bit against liblz4 / the oracle on the CPU-only box — including under an ADVERSARIAL
// choice of which lane wins a same-address LDS store, and with candidate snapshots taken
// `lookahead` windows early (software pipelining on the GPU) — and so that its statistics
// (windows, fast sequences, bail reasons) can guide the kernel design.
//
// Algorithm:
//   general batch   lane i = i-th probe of the current no-match run (any probe index t0), one
//                   speculative insert + read-back, clean prefix, first match, cooperative
//                   extension.  Handles everything; one match per memory round trip.
//   fast window     the chunk is cut into aligned 64-byte windows, lane i <-> position 64k+i.
//                   For every lane the candidate c = T[h] is SNAPSHOT ahead of time together
//                   with what only depends on (pos, c): "candidate matches" (M), forward match
//                   length (capped) and backward equal count (capped).  The window is then
//                   resolved run by run with table traffic only: re-read T[h] (c'), a lane is
//                   usable iff c' == c (nothing inserted into its bucket since the snapshot);
//                   a stale lane whose new candidate matches, or a capped length, bails out to
//                   the general batch at exactly that probe.
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
inline int sched_F(int X) {
  int q = X >> 6, r = X & 63;
  return 32 * q * (q - 1) + q * r;
}
inline int sched_S(int t) { return t < 2 ? t : 2 + sched_F(62 + t); }

struct Ctx {
  const uint8_t* in;
  int len;
  uint8_t* dst;
  int cap;
  int op = 0;
  bool overflow = false;
  std::vector<uint16_t> T;
  Rng rng;
  int winner_mode;
  int mfl1, matchlimit;
  int anchor = 0;
  // parameters
  int fwd_cap, back_cap, lookahead;
  // stats
  int64_t st[16] = {0};
};
enum { ST_GENERAL = 0, ST_CUTS, ST_SEQ, ST_WINDOWS, ST_FAST_SEQ, ST_BAIL_STALE, ST_BAIL_FWD, ST_BAIL_BACK,
       ST_RUN_ITERS, ST_SNAPSHOTS, ST_WIN_SKIPPED_SNAP,
       ST_SPEC_VISITED_STALE, ST_SPEC_LIVE_STALE, ST_LIVE_DUP, ST_SPEC_UNFIXABLE };

void emit_sequence(Ctx& o, int anchor, int lit, bool has_match, int offset, int mcode) {
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
  memcpy(d + o.op, o.in + anchor, (size_t)lit);
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

// one wave-wide "ds_write_b16" of lanes [lo,hi) in an order the caller cannot rely on
void lds_store_all(Ctx& c, const uint32_t* h, const int* pos, int lo, int hi) {
  int order[WAVE], n = 0;
  for (int i = lo; i < hi; i++) order[n++] = i;
  if (c.winner_mode == 1) {
    for (int i = 0; i < n / 2; i++) {
      int t = order[i];
      order[i] = order[n - 1 - i];
      order[n - 1 - i] = t;
    }
  } else if (c.winner_mode == 2) {
    for (int i = n - 1; i > 0; i--) {
      int j = (int)(c.rng.next() % (uint32_t)(i + 1));
      int t = order[i];
      order[i] = order[j];
      order[j] = t;
    }
  }
  for (int k = 0; k < n; k++) c.T[h[order[k]]] = (uint16_t)pos[order[k]];
}

// Cooperative extension + emission of the match found at probe position ip against `match`.
// no_catchup: the probe was the "test next position" probe (t == 0).  Returns the end of the
// match (new anchor), or -1 on output overflow.
int finish_match(Ctx& c, int ip, int match, bool no_catchup) {
  const uint8_t* in = c.in;
  if (!no_catchup)
    while (ip > c.anchor && match > 0 && in[ip - 1] == in[match - 1]) {
      ip--;
      match--;
    }
  int count = 0;
  {
    const int limit = c.matchlimit - (ip + MINMATCH);
    while (count < limit && in[ip + MINMATCH + count] == in[match + MINMATCH + count]) count++;
  }
  emit_sequence(c, c.anchor, ip - c.anchor, true, ip - match, count);
  if (c.overflow) return -1;
  c.st[ST_SEQ]++;
  ip += MINMATCH + count;
  c.anchor = ip;
  return ip;
}

// ---- general batch: returns 0 = continue (state updated), 1 = reached last literals, -1 overflow
int general_batch(Ctx& c, int& base, int& t0) {
  const uint8_t* in = c.in;
  c.st[ST_GENERAL]++;
  int pos[WAVE];
  uint32_t v[WAVE], h[WAVE];
  uint16_t cand[WAVE], r[WAVE];
  const int S0 = sched_S(t0);
  int nvalid = 0;
  for (int i = 0; i < WAVE; i++) {
    const int t = t0 + i;
    pos[i] = base + sched_S(t) - S0;
    const int nextpos = base + sched_S(t + 1) - S0;
    const bool valid = (t == 0) || (nextpos <= c.mfl1);
    if (!valid) break;
    nvalid++;
  }
  for (int i = 0; i < nvalid; i++) {
    v[i] = rd32(in + pos[i]);
    h[i] = hash13(v[i]);
    cand[i] = c.T[h[i]];
  }
  lds_store_all(c, h, pos, 0, nvalid);
  uint64_t C = 0, M = 0, A = 0;
  for (int i = 0; i < nvalid; i++) {
    r[i] = c.T[h[i]];
    if (r[i] != (uint16_t)pos[i]) C |= 1ull << i;
    if (rd32(in + cand[i]) == v[i]) M |= 1ull << i;
    if (i > 0 && v[i] == v[i - 1]) A |= 1ull << i;
  }
  int B = WAVE, c0 = -1;
  bool clean0 = false;
  if (C) {
    c0 = __builtin_ctzll(C);
    clean0 = r[c0] > (uint16_t)pos[c0];
    B = c0 + (clean0 ? 1 : 0);
  }
  const int lim = B < nvalid ? B : nvalid;
  const uint64_t Mv = lim >= 64 ? M : (M & ((1ull << lim) - 1));
  int keep, m = -1;
  bool adj = false;
  if (Mv) {
    m = __builtin_ctzll(Mv);
    keep = m + 1;
  } else if (lim < nvalid && ((A >> lim) & 1)) {
    m = lim;
    adj = true;
    keep = lim + 1;
  } else {
    keep = lim;
  }
  for (int i = keep; i < nvalid; i++)
    if (r[i] == (uint16_t)pos[i]) c.T[h[i]] = cand[i];
  if (clean0 && c0 < keep && !(adj && c0 == m - 1)) c.T[h[c0]] = (uint16_t)pos[c0];
  if (adj) c.T[h[m]] = (uint16_t)pos[m];

  if (m < 0) {
    if (lim == nvalid && nvalid < WAVE) return 1;  // ran into mflimit
    if (lim < nvalid) c.st[ST_CUTS]++;
    base = pos[lim - 1] + (sched_S(t0 + lim) - sched_S(t0 + lim - 1));
    t0 += lim;
    return 0;
  }
  const int ipe = finish_match(c, pos[m], adj ? pos[m - 1] : cand[m], (t0 + m) == 0);
  if (ipe < 0) return -1;
  if (ipe >= c.mfl1) return 1;
  c.T[hash13(rd32(in + ipe - 2))] = (uint16_t)(ipe - 2);
  base = ipe;
  t0 = 0;
  return 0;
}

struct Snap {
  int k = -1;
  uint16_t c[WAVE];
};

void take_snapshot(Ctx& c, Snap& s, int k) {
  s.k = k;
  c.st[ST_SNAPSHOTS]++;
  for (int i = 0; i < WAVE; i++) {
    const int p = 64 * k + i;
    s.c[i] = (p + 4 <= c.len) ? c.T[hash13(rd32(c.in + p))] : 0;
  }
}

// ---- fast window ("exact window"): same return convention as general_batch; on "bail" it leaves
// (base,t0) at the probe the general batch must take next and sets bailed = true.
//
// No speculative inserts.  For every lane i of the aligned window:
//   cp_i    = T[h_i] read once (nothing of this window is in the table yet),
//   eq_i    = mask of window lanes with the same hash (wave "match-any" over the 13 hash bits),
//   Ecp     = lanes whose table candidate matches (rd32(cp_i) == v_i),
//   D       = lanes that have an EARLIER same-hash lane in the window ("suspect").
// Runs are then resolved by mask arithmetic.  K collects the lanes the sequential code inserts
// (probes and ip-2 positions).  For a suspect probe lane the true candidate is the highest lane
// of eq & K below it (an in-window position), else cp.  At the end ONE store commits K: lane i
// writes iff no higher kept lane shares its hash.
int fast_window(Ctx& c, int& base, int& t0, const Snap& snap, bool& bailed) {
  const uint8_t* in = c.in;
  bailed = false;
  c.st[ST_WINDOWS]++;
  const int k = base >> 6, wbase = k << 6;
  int pos[WAVE];
  uint32_t v[WAVE], h[WAVE];
  uint16_t cp[WAVE];
  uint64_t eq[WAVE];
  uint64_t D = 0, Ecp = 0;
  for (int i = 0; i < WAVE; i++) {
    pos[i] = wbase + i;
    v[i] = (pos[i] + 4 <= c.len) ? rd32(in + pos[i]) : 0;
    h[i] = hash13(v[i]);
    cp[i] = c.T[h[i]];
  }
  for (int i = 0; i < WAVE; i++) {
    eq[i] = 0;
    for (int d = 0; d < WAVE; d++)
      if (h[d] == h[i]) eq[i] |= 1ull << d;
    if (eq[i] & ((1ull << i) - 1ull)) D |= 1ull << i;
    if (pos[i] + 4 <= c.len && rd32(in + cp[i]) == v[i]) Ecp |= 1ull << i;
  }
  auto range = [](int lo, int hi) -> uint64_t {  // lanes [lo,hi)
    if (hi <= lo) return 0ull;
    const uint64_t top = hi >= 64 ? ~0ull : ((1ull << hi) - 1ull);
    return top & ~((1ull << lo) - 1ull);
  };
  const int lim_mf = c.mfl1 - wbase;  // a probe at p is valid iff p < mfl1
  uint64_t K = 0;
  int rs = base - wbase, rt = t0;
  int pending_q = -1;
  int rc = -2;
  // statistics for the speculative next-window gather: is the candidate snapshot taken one
  // window early still the table's answer for the lanes that matter?
  uint64_t stale = 0;
  for (int i = 0; i < WAVE; i++)
    if (cp[i] != snap.c[i]) stale |= 1ull << i;
  if (stale & range(rs, WAVE)) c.st[ST_SPEC_LIVE_STALE]++;
  {
    bool dup = false;
    for (int i = rs; i < WAVE && !dup; i++)
      if (eq[i] & range(rs, i)) dup = true;
    if (dup) c.st[ST_LIVE_DUP]++;
  }
  uint64_t visited = 0;
  for (;;) {  // ---- runs
    int e = rs + (65 - rt) + 1;  // consecutive probes need t <= 65
    bool t_limited = true;
    if (e >= WAVE) {
      e = WAVE;
      t_limited = false;
    }
    bool hit_mf = false;
    if (e >= lim_mf) {
      e = lim_mf;
      hit_mf = true;
      t_limited = false;
    }
    int scan = rs, m = -1, match_pos = -1;
    bool dup_match = false;
    for (;;) {
      const uint64_t cm = (Ecp | D) & range(scan, e);
      if (!cm) break;
      const int i = __builtin_ctzll(cm);
      if ((D >> i) & 1) {
        c.st[ST_CUTS]++;  // (stat reused: suspect probes examined)
        const uint64_t dk = eq[i] & ((1ull << i) - 1ull) & (K | range(rs, i));
        if (dk) {
          const int d = 63 - __builtin_clzll(dk);
          if (v[d] == v[i]) {
            m = i;
            match_pos = pos[d];
            dup_match = true;
            break;
          }
          scan = i + 1;
          continue;
        }
      }
      if ((Ecp >> i) & 1) {
        m = i;
        match_pos = cp[i];
        break;
      }
      scan = i + 1;
    }
    visited |= range(rs, m < 0 ? e : m + 1);
    if (m < 0) {
      K |= range(rs, e);
      if (e <= rs && !hit_mf) {  // t too large for consecutive probes: the general batch takes over
        base = wbase + rs;
        t0 = rt;
        bailed = true;
        rc = 0;
      } else if (hit_mf) {
        rc = 1;
      } else {
        base = wbase + e;
        t0 = rt + (e - rs);
        bailed = t_limited;
        rc = 0;
      }
      break;
    }
    K |= range(rs, m + 1);
    const bool t_zero = (rt + (m - rs)) == 0;
    const int ip = pos[m];
    if (dup_match) c.st[ST_BAIL_STALE]++;  // lengths by cooperative extension
    else if (cp[m] != snap.c[m]) c.st[ST_BAIL_BACK]++;  // pipelined snapshot went stale
    else {
      int fwd = 0;
      const int flimit = c.matchlimit - (ip + MINMATCH);
      while (fwd < flimit && fwd < c.fwd_cap && in[ip + MINMATCH + fwd] == in[match_pos + MINMATCH + fwd]) fwd++;
      if ((fwd == c.fwd_cap) && (fwd < flimit)) c.st[ST_BAIL_FWD]++;
    }
    const int ipe = finish_match(c, ip, match_pos, t_zero);
    if (ipe < 0) return -1;  // output overflow: the chunk is stored RAW, table state is moot
    c.st[ST_FAST_SEQ]++;
    if (ipe >= c.mfl1) {
      rc = 1;
      break;
    }
    const int q = ipe - 2;  // LZ4_putPosition(ip - 2)
    if (q < wbase + WAVE) K |= 1ull << (q - wbase);
    else pending_q = q;
    if (ipe >= wbase + WAVE) {
      base = ipe;
      t0 = 0;
      rc = 0;
      break;
    }
    rs = ipe - wbase;
    rt = 0;
  }
  // ---- commit: the last kept lane of every hash group writes its position ----------------------
  for (int i = 0; i < WAVE; i++)
    if (((K >> i) & 1) && !(eq[i] & K & ~((2ull << i) - 1ull))) c.T[h[i]] = (uint16_t)pos[i];
  if (pending_q >= 0) c.T[hash13(rd32(in + pending_q))] = (uint16_t)pending_q;
  c.st[ST_RUN_ITERS]++;
  if (stale & visited) c.st[ST_SPEC_VISITED_STALE]++;
  {  // ... and is the fresh candidate of every such lane a position of the PREVIOUS window (its bytes are in registers)?
    bool unfixable = false;
    for (int i = 0; i < WAVE; i++)
      if (((stale & visited) >> i) & 1)
        if (!(cp[i] >= wbase - 64 && cp[i] < wbase)) unfixable = true;
    if (unfixable) c.st[ST_SPEC_UNFIXABLE]++;
  }
  return rc;
}

}  // namespace

// mode bit 0..1: winner_mode (0 highest lane, 1 lowest lane, 2 random); fast: 0 = general batches
// only, 1 = fast windows enabled.  stats[16] accumulates.  Returns size or -1 (store RAW).
extern "C" int lz4_window_model_compress(const uint8_t* src, int len, uint8_t* dst, int winner_mode,
                                         uint64_t seed, int fast, int fwd_cap, int back_cap,
                                         int lookahead, int64_t* stats) {
  std::vector<uint8_t> padded((size_t)len + 512, 0);
  memcpy(padded.data(), src, (size_t)len);
  Ctx c;
  c.in = padded.data();
  c.len = len;
  c.dst = dst;
  c.cap = len;
  c.T.assign(8192, 0);
  c.rng = Rng{seed};
  c.winner_mode = winner_mode;
  c.mfl1 = len - MFLIMIT + 1;
  c.matchlimit = len - LASTLITERALS;
  c.fwd_cap = fwd_cap;
  c.back_cap = back_cap;
  c.lookahead = lookahead;

  if (len >= MFLIMIT + 1) {
    c.T[hash13(rd32(c.in))] = 0;
    int base = 1, t0 = 1;
    Snap ring[8];
    bool force_general = false;
    for (;;) {
      int rc;
      if (fast && !force_general && t0 <= 48) {
        const int k = base >> 6;
        // snapshots: window k must have one (taken now if the pipeline did not provide it), and
        // windows k+1..k+lookahead are snapshot BEFORE window k is resolved
        for (int d = 0; d <= lookahead; d++) {
          Snap& s = ring[(k + d) & 7];
          if (s.k != k + d && 64 * (k + d) < len) take_snapshot(c, s, k + d);
        }
        bool bailed = false;
        rc = fast_window(c, base, t0, ring[k & 7], bailed);
        force_general = bailed;
      } else {
        rc = general_batch(c, base, t0);
        force_general = false;
      }
      if (rc < 0) return -1;
      if (rc == 1) break;
    }
  }
  emit_sequence(c, c.anchor, len - c.anchor, false, 0, 0);
  if (c.overflow) return -1;
  if (stats)
    for (int i = 0; i < 16; i++) stats[i] += c.st[i];
  return c.op;
}
