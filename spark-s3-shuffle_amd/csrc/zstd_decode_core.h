// zstd_decode_core.h — Zstandard frame decoder (RFC 8878), written once for the device and for the host.
//
// Reduce side of SURVEY §8 f4: with spark.io.compression.codec=zstd the reference's reader hands every fetched block
// range to zstd-jni's ZstdInputStream through serializerManager.wrapStream (storage/S3ShuffleReader.scala:98-110,
// codec matrix at :55-75); one frame per non-empty partition (several when a partition was written in spill pieces:
// frames concatenate).  This file is the decoder behind S3S_CODEC_ZSTD in s3s_decompress_range*: COMPRESSION stays on
// the JVM (libzstd's output is not reproducible by a data-parallel program), so the parity bar here is
// "decodes every stream libzstd 1.4.8 — the library in this image, and the format zstd-jni writes — produces".
//
// Shape: one SEQUENCE wavefront per partition.  The entropy stages are serial by construction (one FSE bit stream for a
// block's sequences, four Huffman streams for its literals), so the wave runs the control flow uniformly — every lane
// computes the same header / table / sequence values — and spreads only what has width: literal runs and matches are copied
// 64 bytes per step, table fills are strided over the lanes.  The four Huffman streams of a block go to lanes 0..3 of a
// LITERAL wavefront (round 4), which works one block ahead of the sequence wavefront and serves two partitions (literal_side,
// LitPipe below).  Throughput therefore comes from the number of partitions in flight (a batched call holds thousands), not
// from a single frame; a 1 GiB single-partition block is one wavefront's serial work and belongs on the host.
//
// Compiled by hipcc (S3S_ZSTD_DEVICE: zstd_decompress.hip) and by g++ (tests/model/zstd_decode_model.cpp: the same
// code with one "lane", checked against libzstd on the CPU before it ever reaches a GPU).
#pragma once
#include <stdint.h>
#include <string.h>

#ifdef S3S_ZSTD_DEVICE
#define ZS_HD __device__ __forceinline__
#define ZS_RARE __device__ __attribute__((noinline))
#define ZS_FENCE() __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup")
// every lane holds the same value: say so (the value moves to a scalar register, what depends on it runs on the scalar unit)
#define ZS_UNI32(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
#else
#define ZS_HD static inline
#define ZS_RARE static
#define ZS_FENCE() ((void)0)
#define ZS_UNI32(x) ((uint32_t)(x))
#endif

#ifdef ZS_TRACE  // host debugging only: which check refused the stream
#include <stdio.h>
#define ZS_FAIL() (fprintf(stderr, "zstd_decode_core.h:%d\n", __LINE__), ZS_BAD)
#else
#define ZS_FAIL() ZS_BAD
#endif

namespace s3s_zstd {

enum { ZS_OK = 0, ZS_BAD = -3, ZS_CAPACITY = -2, ZS_UNSUPPORTED = -6 };
constexpr int kMaxBlock = 1 << 17;
constexpr int kHufLogMax = 11;
constexpr int kLLLogMax = 9, kMLLogMax = 9, kOFLogMax = 8;
constexpr int kRing = 4096, kLitW = 1024;
constexpr uint32_t kWaveLanes = 64;

// The Huffman side of a partition (round 4: its own wavefront on the device, see literal_side): the literals table and the
// scratch its description is read through.
struct HufWork {
  uint16_t huf[1 << kHufLogMax];   // (symbol << 8) | nbBits
  int16_t norm[16];                // normalized counts of the FSE-coded weights (weights are 0..12)
  uint16_t next[16];
  uint8_t weights[256];
  uint32_t wtab[64];               // FSE table of the Huffman weights (tableLog <= 6)
  int32_t huf_log, have_huf;       // have_huf: "treeless" literals need a previous table (of the same frame)
};
// What the two sides of a partition share.  Device: the literal wavefront decodes the Huffman streams of block k + 1 into
// literal buffer (k + 1) & 1 while the sequence wavefront executes block k from buffer k & 1; `ready` / `consumed` count the
// Huffman-coded blocks handed over / given back, `err` is the literal side's verdict, `quit` the sequence side's leave.
// Host model: one thread does both, block by block (the counters stay unused).
struct LitPipe {
  HufWork h;
  int32_t ready, consumed, err, quit;
};

// per-wavefront tables of the sequence side (LDS on the device)
struct Work {
  uint32_t ll[1 << kLLLogMax];     // baseline | nbBits << 16 | symbol << 24
  uint32_t ml[1 << kMLLogMax];
  uint32_t of[1 << kOFLogMax];
  uint32_t llv[1 << kLLLogMax];    // per state: literal-length base | extra bits << 24 (so the loop reads no constant tables)
  uint32_t mlv[1 << kMLLogMax];    // per state: match-length base | extra bits << 24
  int16_t norm[64];                // normalized counts (FSE header; at most 53 symbols: match-length codes)
  uint16_t next[64];               // symbolNext
  int32_t ll_log, ml_log, of_log;
  int32_t have_ll, have_ml, have_of;  // "repeat" modes need a previous table
  int32_t vals_ll, vals_ml;
  // the frame's most recent output (position q lives at ring[q & (kRing - 1)]): match sources come from here, not from
  // global memory (a global source would need the stores of the sequences before it to have completed: ~1 us each)
  uint8_t ring[kRing];
  uint8_t litw[kLitW];             // window over the block's literals, refilled 1 KiB at a time
#ifdef S3S_ZSTD_DEVICE
  // what the compiled sequence loop hands to the lean one and gets back (seq_fast below)
  struct Fast {
    const uint8_t* bits;           // the block's sequence bit stream
    uint8_t* bdst;                 // where the block's output starts
    int32_t bits_size, pos, cbase;
    uint32_t cache_lo, cache_hi;
    uint32_t sl, so, sm, rep0, rep1, rep2;
    uint32_t i, nseq;              // next sequence, sequences of the block
    uint32_t lit_pos, bout, regen, bcap, rq0;
    int32_t litw_base;
    uint32_t visible;              // bytes of the block behind which a fence has been issued (far match sources)
    uint32_t hist_lo, hist_hi;     // bytes of the frame in front of the block
  } fast;
#endif
};

struct Lanes {  // who am I in the wavefront (host: lane 0 of 1)
  int lane, n;
};

ZS_HD uint32_t rd_le(const uint8_t* p, int nbytes) {
  uint32_t v = 0;
  for (int i = 0; i < nbytes; i++) v |= (uint32_t)p[i] << (8 * i);
  return v;
}
ZS_HD int highbit(uint32_t v) { return 31 - __builtin_clz(v); }

// ---- backward bit stream (the convention of FSE / Huffman streams in zstd) ----------------------------------------------
// The stream is bytes [0, size); its last byte carries a final 1 bit as end mark.  Bits are consumed from the top.
// `pos` = number of unread bits below the cursor; a read of n bits takes bits [pos - n, pos).  Reading below bit 0 gives
// zeros and makes pos negative: the callers check pos at the points where the format demands it.
struct BitR {           // (a stream never exceeds one block, 128 KiB: 32-bit positions)
  const uint8_t* p;
  int32_t size;
  int32_t pos;
  uint64_t cache;      // bits [cbase, cbase + 64)
  int32_t cbase;
  bool uniform;        // every lane reads this stream in lock step (sequences, weights): values can live in scalar registers
};
ZS_HD uint64_t load_bits64(const uint8_t* p, int32_t size, int32_t byte0) {  // bytes [byte0, byte0+8) with zeros outside [0,size)
  uint64_t v = 0;
  if (byte0 >= 0 && byte0 + 8 <= size) {
    memcpy(&v, p + byte0, 8);
    return v;
  }
  for (int i = 0; i < 8; i++) {
    const int32_t b = byte0 + i;
    if (b >= 0 && b < size) v |= (uint64_t)p[b] << (8 * i);
  }
  return v;
}
ZS_HD bool bitr_init(BitR& r, const uint8_t* p, int64_t size64, bool uniform = true) {
  r.uniform = uniform;
  r.p = p;
  r.size = (int32_t)size64;
  r.cbase = 1 << 30;  // (no cache)
  r.cache = 0;
  r.pos = 0;
  if (size64 <= 0 || size64 > kMaxBlock) return false;
  const uint32_t last = p[size64 - 1];
  if (last == 0) return false;  // no end mark
  r.pos = (r.size - 1) * 8 + highbit(last);
  return true;
}
ZS_HD uint32_t bitr_peek(BitR& r, int n) {  // n <= 32; bits below the start of the stream read as zero
  if (n == 0) return 0;
  const int32_t lo = r.pos - n;
  if (!(lo >= r.cbase && r.pos <= r.cbase + 64)) {
    // window whose top byte holds bit pos-1 (arithmetic shift: positions below the stream stay consistent)
    const int32_t top = (r.pos - 1) >> 3;
    const int32_t b0 = top - 7;
    r.cbase = b0 * 8;
    r.cache = load_bits64(r.p, r.size, b0);
#ifdef S3S_ZSTD_DEVICE
    if (r.uniform) r.cache = (uint64_t)ZS_UNI32((uint32_t)r.cache) | ((uint64_t)ZS_UNI32((uint32_t)(r.cache >> 32)) << 32);
#endif
  }
  return (uint32_t)((r.cache >> (lo - r.cbase)) & ((n == 32) ? 0xFFFFFFFFull : ((1ull << n) - 1)));
}
ZS_HD uint32_t bitr_read(BitR& r, int n) {
  const uint32_t v = bitr_peek(r, n);
  r.pos -= n;
  return v;
}

// ---- forward bit reader for FSE table descriptions ---------------------------------------------------------------------------
// Normalized counts (FSE_readNCount): returns bytes consumed (> 0) or a negative error.  *max_sym out, *table_log out.
ZS_HD int read_ncount(int16_t* norm, int max_sym_allowed, int max_log, const uint8_t* p, int64_t size, int* out_max_sym,
                      int* out_log) {
  if (size < 1) return ZS_FAIL();
  uint64_t bits = 0;
  int nb = 0;      // valid bits in `bits`
  int64_t ip = 0;  // next byte to load
  auto fill = [&]() {
    while (nb <= 56 && ip < size) {
      bits |= (uint64_t)p[ip++] << nb;
      nb += 8;
    }
  };
  int64_t used_bits = 0;
  auto take = [&](int n) -> uint32_t {
    fill();
    const uint32_t v = (uint32_t)(bits & ((1ull << n) - 1));
    bits >>= n;
    nb -= n;
    used_bits += n;
    return v;
  };
  const int table_log = (int)take(4) + 5;
  if (table_log > max_log) return ZS_FAIL();
  int remaining = (1 << table_log) + 1;
  int threshold = 1 << table_log;
  int nbits = table_log + 1;
  int sym = 0;
  bool prev0 = false;
  while (remaining > 1 && sym <= max_sym_allowed) {
    if (prev0) {
      // repeat flags: 2 bits each, value 3 = three more zero symbols and another flag follows
      int n0 = sym;
      for (;;) {
        const uint32_t f = take(2);
        n0 += (int)f;
        if (f != 3) break;
        if (n0 > max_sym_allowed + 1) return ZS_FAIL();
      }
      if (n0 > max_sym_allowed + 1) return ZS_FAIL();
      while (sym < n0) norm[sym++] = 0;
      prev0 = false;
      if (sym > max_sym_allowed) break;
    }
    fill();
    const int max = (2 * threshold - 1) - remaining;
    int count;
    const uint32_t low = (uint32_t)(bits & (uint64_t)(threshold - 1));
    if ((int)low < max) {
      count = (int)low;
      bits >>= (nbits - 1);
      nb -= nbits - 1;
      used_bits += nbits - 1;
    } else {
      count = (int)(bits & (uint64_t)(2 * threshold - 1));
      if (count >= threshold) count -= max;
      bits >>= nbits;
      nb -= nbits;
      used_bits += nbits;
    }
    count--;  // -1 means "less than one" probability
    remaining -= count < 0 ? -count : count;
    norm[sym++] = (int16_t)count;
    prev0 = count == 0;
    while (remaining < threshold) {
      nbits--;
      threshold >>= 1;
    }
    if (nb < 0) return ZS_FAIL();
  }
  if (remaining != 1) return ZS_FAIL();
  if (sym > max_sym_allowed + 1) return ZS_FAIL();
  *out_max_sym = sym - 1;
  *out_log = table_log;
  const int64_t bytes = (used_bits + 7) >> 3;
  if (bytes > size) return ZS_FAIL();
  return (int)bytes;
}

// FSE decoding table from normalized counts (FSE_buildDTable).  entry = baseline | nbBits << 16 | symbol << 24
ZS_HD int build_fse(uint32_t* tab, const int16_t* norm, int max_sym, int table_log, uint16_t* next) {
  const int size = 1 << table_log;
  int high = size - 1;
  for (int s = 0; s <= max_sym; s++) {
    if (norm[s] == -1) {
      tab[high--] = (uint32_t)s << 24;
      next[s] = 1;
    } else {
      next[s] = (uint16_t)norm[s];
    }
  }
  const int mask = size - 1, step = (size >> 1) + (size >> 3) + 3;
  int pos = 0;
  for (int s = 0; s <= max_sym; s++) {
    for (int i = 0; i < norm[s]; i++) {
      tab[pos] = (uint32_t)s << 24;
      pos = (pos + step) & mask;
      while (pos > high) pos = (pos + step) & mask;
    }
  }
  if (pos != 0) return ZS_FAIL();
  for (int u = 0; u < size; u++) {
    const int s = (int)(tab[u] >> 24);
    const uint32_t ns = next[s]++;
    const int nbb = table_log - highbit(ns);
    const uint32_t base = (ns << nbb) - (uint32_t)size;
    tab[u] = (base & 0xFFFFu) | ((uint32_t)nbb << 16) | ((uint32_t)s << 24);
  }
  return ZS_OK;
}
ZS_HD void build_rle(uint32_t* tab, int sym) { tab[0] = (uint32_t)sym << 24; }  // tableLog 0: one state, 0 bits

// predefined distributions (RFC 8878 §3.1.1.3.2.2)
ZS_HD int default_norm(int which, int16_t* norm, int* log) {  // 0 = LL, 1 = OF, 2 = ML; returns max symbol
  static const int8_t LL[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
  static const int8_t OF[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};
  static const int8_t ML[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
  if (which == 0) {
    for (int i = 0; i < 36; i++) norm[i] = LL[i];
    *log = 6;
    return 35;
  }
  if (which == 1) {
    for (int i = 0; i < 29; i++) norm[i] = OF[i];
    *log = 5;
    return 28;
  }
  for (int i = 0; i < 53; i++) norm[i] = ML[i];
  *log = 6;
  return 52;
}

ZS_HD uint32_t ll_base(int c) {
  static const uint32_t B[36] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 22, 24, 28, 32, 40,
                                 48, 64, 0x80, 0x100, 0x200, 0x400, 0x800, 0x1000, 0x2000, 0x4000, 0x8000, 0x10000};
  return B[c];
}
ZS_HD int ll_bits(int c) {
  static const uint8_t B[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
  return B[c];
}
ZS_HD uint32_t ml_base(int c) {
  static const uint32_t B[53] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29,
                                 30, 31, 32, 33, 34, 35, 37, 39, 41, 43, 47, 51, 59, 67, 83, 99, 0x83, 0x103, 0x203, 0x403, 0x803,
                                 0x1003, 0x2003, 0x4003, 0x8003, 0x10003};
  return B[c];
}
ZS_HD int ml_bits(int c) {
  static const uint8_t B[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
  return B[c];
}

// ---- Huffman ----------------------------------------------------------------------------------------------------------------
// Tree description -> decoding table.  Returns bytes consumed or a negative error.
ZS_HD int read_huf_table(HufWork& w, const uint8_t* p, int64_t size) {
  if (size < 1) return ZS_FAIL();
  const int hb = p[0];
  int nsym = 0;  // weights given explicitly
  int used;
  if (hb >= 128) {  // direct: 4-bit weights
    nsym = hb - 127;
    used = 1 + (nsym + 1) / 2;
    if (used > size) return ZS_FAIL();
    for (int i = 0; i < nsym; i++) w.weights[i] = (i & 1) ? (p[1 + i / 2] & 15) : (p[1 + i / 2] >> 4);
  } else {  // FSE-compressed weights: two interleaved states over one backward stream
    used = 1 + hb;
    if (hb == 0 || used > size) return ZS_FAIL();
    int max_sym, log;
    const int hdr = read_ncount(w.norm, 12, 6, p + 1, hb, &max_sym, &log);  // weights are 0..12 (HUF_TABLELOG_ABSOLUTEMAX)
    if (hdr < 0) return hdr;
    if (build_fse(w.wtab, w.norm, max_sym, log, w.next) != ZS_OK) return ZS_FAIL();
    BitR r;
    if (!bitr_init(r, p + 1 + hdr, hb - hdr)) return ZS_FAIL();
    uint32_t s1 = bitr_read(r, log), s2 = bitr_read(r, log);
    if (r.pos < 0) return ZS_FAIL();
    // FSE_decompress_usingDTable: alternate the two states; when the stream runs dry the other state still holds a symbol
    for (;;) {
      if (nsym > 253) return ZS_FAIL();
      uint32_t e = w.wtab[s1];
      w.weights[nsym++] = (uint8_t)(e >> 24);
      int nb = (int)((e >> 16) & 0xff);
      if (r.pos < nb) {  // cannot update s1: s2's symbol is the last one
        w.weights[nsym++] = (uint8_t)(w.wtab[s2] >> 24);
        break;
      }
      s1 = (e & 0xFFFFu) + bitr_read(r, nb);
      if (nsym > 253) return ZS_FAIL();
      e = w.wtab[s2];
      w.weights[nsym++] = (uint8_t)(e >> 24);
      nb = (int)((e >> 16) & 0xff);
      if (r.pos < nb) {
        w.weights[nsym++] = (uint8_t)(w.wtab[s1] >> 24);
        break;
      }
      s2 = (e & 0xFFFFu) + bitr_read(r, nb);
    }
  }
  if (nsym < 1 || nsym > 255) return ZS_FAIL();
  // the last weight follows from the sum being a power of two
  uint32_t total = 0;
  for (int i = 0; i < nsym; i++) {
    if (w.weights[i] > 12) return ZS_FAIL();
    if (w.weights[i]) total += 1u << (w.weights[i] - 1);
  }
  if (total == 0) return ZS_FAIL();
  const int log = highbit(total) + 1;
  if (log > kHufLogMax) return ZS_FAIL();
  const uint32_t rest = (1u << log) - total;
  if (rest == 0 || (rest & (rest - 1))) return ZS_FAIL();  // must be a power of two
  w.weights[nsym] = (uint8_t)(highbit(rest) + 1);
  nsym++;
  // table: symbols of weight W occupy 2^(W-1) consecutive cells each, smaller weights first, symbols in order
  uint32_t rank_start[14];
  for (int i = 0; i < 14; i++) rank_start[i] = 0;
  for (int i = 0; i < nsym; i++) rank_start[w.weights[i]]++;  // counts
  if (rank_start[1] < 2 || (rank_start[1] & 1)) return ZS_FAIL();  // (at least two symbols of the longest length, in pairs)
  uint32_t at = 0;
  for (int wt = 1; wt <= log; wt++) {
    const uint32_t cnt = rank_start[wt];
    rank_start[wt] = at;
    at += cnt << (wt - 1);
  }
  if (at != (1u << log)) return ZS_FAIL();
  for (int s = 0; s < nsym; s++) {
    const int wt = w.weights[s];
    if (!wt) continue;
    const uint32_t len = 1u << (wt - 1), nb = (uint32_t)(log + 1 - wt);
    const uint32_t st = rank_start[wt];
    for (uint32_t u = 0; u < len; u++) w.huf[st + u] = (uint16_t)((s << 8) | nb);
    rank_start[wt] += len;
  }
  w.huf_log = log;
  w.have_huf = 1;
  return used;
}

// one Huffman stream: n symbols into dst
ZS_HD int huf_decode_stream(const HufWork& w, const uint8_t* p, int64_t size, uint8_t* dst, int64_t n) {
  BitR r;
  if (!bitr_init(r, p, size, false)) return ZS_FAIL();
  const int log = w.huf_log;
  int64_t i = 0;
#ifndef ZS_NO_HUF4
  // Four symbols per refill (round 4): a window of 64 bits whose top byte holds bit pos - 1 has at least 57 bits below the cursor, four
  // codes take at most 44 (kHufLogMax = 11), so the cursor never leaves the window in between: one bounds-free 8-byte load, four
  // table lookups, one 4-byte store - instead of a refill check and a byte store per symbol.  pos >= 64 keeps the window inside the
  // stream (no zero fill below its first byte); the last symbols of a stream take the loop below.
  {
    const int32_t n5 = (int32_t)n - 4;  // (a stream regenerates at most 128 KiB: 32-bit counters)
    const int up = 32 - log;
    int32_t i4 = 0;
    while (i4 < n5 && r.pos >= 64) {  // FIVE symbols per window: five codes take at most 55 of the >= 57 bits below the cursor
      const int32_t b0 = ((r.pos - 1) >> 3) - 7;
      uint64_t cache;
      memcpy(&cache, r.p + b0, 8);
      const int32_t rel = r.pos - b0 * 8;  // 57 .. 64 bits below the cursor
      uint64_t top = cache << (64 - rel);  // the unread bits, left-aligned: a code is the top `log` bits of the high word
      uint32_t out4 = 0, used = 0;
      for (int k = 0; k < 4; k++) {  // (unrolled by both compilers)
        const uint16_t e = w.huf[(uint32_t)(top >> 32) >> up];
        const uint32_t nbits = e & 0xff;
        top <<= nbits;
        used += nbits;
        out4 |= (uint32_t)(e >> 8) << (8 * k);
      }
      const uint16_t e5 = w.huf[(uint32_t)(top >> 32) >> up];
      used += e5 & 0xff;
      r.pos -= (int32_t)used;
      memcpy(dst + i4, &out4, 4);
      dst[i4 + 4] = (uint8_t)(e5 >> 8);
      i4 += 5;
    }
    i = i4;
    r.cbase = 1 << 30;  // (the per-symbol loop refills its own window)
  }
#endif
  for (; i < n; i++) {
    const uint32_t v = bitr_peek(r, log);  // (bits below the start of the stream read as zero, like libzstd's container)
    const uint16_t e = w.huf[v];
    r.pos -= e & 0xff;
    if (r.pos < 0) return ZS_FAIL();
    dst[i] = (uint8_t)(e >> 8);
  }
  return r.pos == 0 ? ZS_OK : ZS_BAD;  // the stream must be consumed exactly
}

// ---- cooperative copies ---------------------------------------------------------------------------------------------------------
// Every output byte goes to global memory AND into the ring.  o = where byte 0 of the run goes, rq = its position in the frame
// (32 bits of it: the ring index).  32-bit lane arithmetic on purpose: the runs are tens of bytes, 64-bit index math per lane
// was half of the decode pass.
ZS_HD void put_plain(Work& w, uint8_t* o, uint32_t rq, const uint8_t* src, uint32_t n, Lanes L) {  // src: global, not the output
  for (uint32_t i = (uint32_t)L.lane; i < n; i += (uint32_t)L.n) {
    const uint8_t v = src[i];
    w.ring[(rq + i) & (kRing - 1)] = v;
    o[i] = v;
  }
}
ZS_HD void put_fill(Work& w, uint8_t* o, uint32_t rq, uint8_t v, uint32_t n, Lanes L) {
  for (uint32_t i = (uint32_t)L.lane; i < n; i += (uint32_t)L.n) {
    w.ring[(rq + i) & (kRing - 1)] = v;
    o[i] = v;
  }
}
// literals [lp, lp + n) of the block: through the LDS window when the run fits it (one coalesced refill per 1 KiB of
// literals instead of a global round trip per sequence)
// litw_base: literal position of litw[0] (-1: nothing loaded) - the caller's register, not LDS: it is read per sequence
ZS_HD void put_literals(Work& w, uint8_t* o, uint32_t rq, const uint8_t* lit, uint32_t lit_total, uint32_t lp, uint32_t n, Lanes L,
                        int32_t& litw_base) {
  if (n > (uint32_t)kLitW) {
    put_plain(w, o, rq, lit + lp, n, L);
    return;
  }
  int32_t base = litw_base;
  if (base < 0 || (int32_t)lp < base || (int32_t)(lp + n) > base + kLitW) {
    const int32_t left = (int32_t)(lit_total - lp);
    const int32_t m = left < kLitW ? left : kLitW;
    for (int32_t i = L.lane; i < m; i += L.n) w.litw[i] = lit[lp + i];
    litw_base = (int32_t)lp;
    base = (int32_t)lp;
#ifdef S3S_ZSTD_DEVICE
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
#endif
  }
  const uint8_t* wsrc = w.litw + ((int32_t)lp - base);
  for (uint32_t i = (uint32_t)L.lane; i < n; i += (uint32_t)L.n) {
    const uint8_t v = wsrc[i];
    w.ring[(rq + i) & (kRing - 1)] = v;
    o[i] = v;
  }
}
// match: o[i] = o[i - offset], i in [0, n); with offset < n the source repeats with period `offset`, and every
// byte of the period was written BEFORE this copy started, so no lane depends on another lane's store.  near = the
// source is still in the ring (offset + n <= kRing: this copy does not overwrite what it reads).
ZS_HD void put_match(Work& w, uint8_t* o, uint32_t rq, uint32_t off, uint32_t n, bool near, Lanes L) {
  if (near) {
    const uint32_t rs = rq - off;
    if (off >= n) {
      for (uint32_t i = (uint32_t)L.lane; i < n; i += (uint32_t)L.n) {
        const uint8_t v = w.ring[(rs + i) & (kRing - 1)];
        w.ring[(rq + i) & (kRing - 1)] = v;
        o[i] = v;
      }
    } else if (off == 1) {  // a run of one byte
      const uint8_t v = w.ring[rs & (kRing - 1)];
      for (uint32_t i = (uint32_t)L.lane; i < n; i += (uint32_t)L.n) {
        w.ring[(rq + i) & (kRing - 1)] = v;
        o[i] = v;
      }
    } else {
      for (uint32_t i = (uint32_t)L.lane; i < n; i += (uint32_t)L.n) {
        const uint8_t v = w.ring[(rs + i % off) & (kRing - 1)];
        w.ring[(rq + i) & (kRing - 1)] = v;
        o[i] = v;
      }
    }
    return;
  }
  const uint8_t* pat = o - (int64_t)off;
  if (off >= n) {
    for (uint32_t i = (uint32_t)L.lane; i < n; i += (uint32_t)L.n) {
      const uint8_t v = pat[i];
      w.ring[(rq + i) & (kRing - 1)] = v;
      o[i] = v;
    }
  } else {  // (offset + n > kRing and offset < n: a long run, rare)
    for (uint32_t i = (uint32_t)L.lane; i < n; i += (uint32_t)L.n) {
      const uint8_t v = pat[i % off];
      w.ring[(rq + i) & (kRing - 1)] = v;
      o[i] = v;
    }
  }
}

// ---- XXH64 of a frame's content (RFC 8878 3.1.1: the optional Content_Checksum is its low 32 bits, seed 0) --------------------------
// xxHash's published algorithm restated (third party: Cyan4973/xxHash, the code libzstd calls; pinned through libzstd itself -
// every checksummed frame libzstd writes must verify, a damaged one must not).  A Spark writer's frames carry no checksum
// (zstd-jni's default), so this runs for foreign frames only.  Device: the four accumulators of a 32-byte stripe are lanes
// 0..3; the host walks them one after the other.  p[0, n) is the frame's output in global memory, fenced by the caller.
ZS_HD uint64_t xxh64_rotl(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
ZS_HD uint64_t xxh64_rd64(const uint8_t* p) {
  uint64_t v;
  memcpy(&v, p, 8);
  return v;
}
// (not inlined on the device: a rare path whose 64-bit multiplies must not cost the sequence loop its registers)
ZS_RARE uint64_t xxh64_content(const uint8_t* p, int64_t n, Lanes L) {
  const uint64_t P1 = 0x9E3779B185EBCA87ull, P2 = 0xC2B2AE3D27D4EB4Full, P3 = 0x165667B19E3779F9ull, P4 = 0x85EBCA77C2B2AE63ull,
                 P5 = 0x27D4EB2F165667C5ull;
  uint64_t h;
  int64_t at = 0;
  if (n >= 32) {
    const int64_t stripes = n >> 5;
    uint64_t v0 = P1 + P2, v1 = P2, v2 = 0, v3 = 0ull - P1;  // (four scalars: an indexed array would live in scratch memory)
#ifdef S3S_ZSTD_DEVICE
    {
      const int k = L.lane & 3;
      uint64_t acc = k == 0 ? v0 : k == 1 ? v1 : k == 2 ? v2 : v3;
      const uint8_t* q = p + 8 * k;
      for (int64_t t = 0; t < stripes; t++) acc = xxh64_rotl(acc + xxh64_rd64(q + 32 * t) * P2, 31) * P1;
      const int lo = (int)(uint32_t)acc, hi = (int)(uint32_t)(acc >> 32);  // (lanes 0..3 hold v1..v4: to everybody)
      v0 = ((uint64_t)(uint32_t)__shfl(hi, 0) << 32) | (uint32_t)__shfl(lo, 0);
      v1 = ((uint64_t)(uint32_t)__shfl(hi, 1) << 32) | (uint32_t)__shfl(lo, 1);
      v2 = ((uint64_t)(uint32_t)__shfl(hi, 2) << 32) | (uint32_t)__shfl(lo, 2);
      v3 = ((uint64_t)(uint32_t)__shfl(hi, 3) << 32) | (uint32_t)__shfl(lo, 3);
    }
#else
    (void)L;
    for (int64_t t = 0; t < stripes; t++) {
      const uint8_t* q = p + 32 * t;
      v0 = xxh64_rotl(v0 + xxh64_rd64(q) * P2, 31) * P1;
      v1 = xxh64_rotl(v1 + xxh64_rd64(q + 8) * P2, 31) * P1;
      v2 = xxh64_rotl(v2 + xxh64_rd64(q + 16) * P2, 31) * P1;
      v3 = xxh64_rotl(v3 + xxh64_rd64(q + 24) * P2, 31) * P1;
    }
#endif
    h = xxh64_rotl(v0, 1) + xxh64_rotl(v1, 7) + xxh64_rotl(v2, 12) + xxh64_rotl(v3, 18);
    h = (h ^ (xxh64_rotl(v0 * P2, 31) * P1)) * P1 + P4;
    h = (h ^ (xxh64_rotl(v1 * P2, 31) * P1)) * P1 + P4;
    h = (h ^ (xxh64_rotl(v2 * P2, 31) * P1)) * P1 + P4;
    h = (h ^ (xxh64_rotl(v3 * P2, 31) * P1)) * P1 + P4;
    at = stripes << 5;
  } else {
    h = P5;
  }
  h += (uint64_t)n;
  for (; at + 8 <= n; at += 8) h = xxh64_rotl(h ^ (xxh64_rotl(xxh64_rd64(p + at) * P2, 31) * P1), 27) * P1 + P4;
  if (at + 4 <= n) {
    uint32_t w4;
    memcpy(&w4, p + at, 4);
    h = xxh64_rotl(h ^ ((uint64_t)w4 * P1), 23) * P2 + P3;
    at += 4;
  }
  for (; at < n; at++) h = xxh64_rotl(h ^ ((uint64_t)p[at] * P5), 11) * P1;
  h ^= h >> 33;
  h *= P2;
  h ^= h >> 29;
  h *= P3;
  h ^= h >> 32;
  return h;
}

// ---- one frame ------------------------------------------------------------------------------------------------------------------
struct FrameOut {
  int64_t consumed;  // bytes of src this frame occupied
  int64_t produced;  // decoded bytes
  int64_t lit_need;  // largest regenerated (Huffman-coded) literals section of any block: what one literal buffer must hold
};

struct FrameHdr {
  int64_t bytes;     // header bytes (skippable frame: the whole frame)
  int64_t fcs;       // frame content size, -1 when absent
  int has_check, skippable;
};
// Frame header at src[0, size) (RFC 8878 3.1.1.1); both sides of a partition walk the same headers.
ZS_HD int frame_header(const uint8_t* src, int64_t size, FrameHdr* fh) {
  fh->fcs = -1;
  fh->has_check = fh->skippable = 0;
  if (size < 4) return ZS_FAIL();
  const uint32_t magic = rd_le(src, 4);
  if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) {  // skippable frame
    if (size < 8) return ZS_FAIL();
    const int64_t n = rd_le(src + 4, 4);
    if (8 + n > size) return ZS_FAIL();
    fh->bytes = 8 + n;
    fh->skippable = 1;
    return ZS_OK;
  }
  if (magic != 0xFD2FB528u) return ZS_FAIL();
  if (size < 6) return ZS_FAIL();
  const uint32_t fhd = src[4];
  const int fcs_flag = (int)(fhd >> 6), single = (int)((fhd >> 5) & 1), did_flag = (int)(fhd & 3);
  fh->has_check = (int)((fhd >> 2) & 1);
  if (fhd & 0x08) return ZS_FAIL();  // reserved bit
  int64_t ip = 5;
  if (!single) ip++;  // window descriptor: a block is at most kMaxBlock whatever it says, and history is the whole frame here
  static const int did_bytes[4] = {0, 1, 2, 4};
  if (ip + did_bytes[did_flag] > size) return ZS_FAIL();
  if (did_flag && rd_le(src + ip, did_bytes[did_flag]) != 0) return ZS_UNSUPPORTED;  // dictionaries are not used by Spark
  ip += did_bytes[did_flag];
  const int fcs_bytes = fcs_flag == 0 ? (single ? 1 : 0) : (fcs_flag == 1 ? 2 : (fcs_flag == 2 ? 4 : 8));
  if (ip + fcs_bytes > size) return ZS_FAIL();
  if (fcs_bytes) {
    uint64_t v = 0;
    for (int i = 0; i < fcs_bytes; i++) v |= (uint64_t)src[ip + i] << (8 * i);
    if (fcs_bytes == 2) v += 256;
    fh->fcs = (int64_t)v;
    ip += fcs_bytes;
  }
  fh->bytes = ip;
  return ZS_OK;
}

struct LitHdr {
  int ltype;         // 0 raw, 1 RLE, 2 Huffman with a table, 3 Huffman with the previous table
  int lh, streams;   // header bytes, Huffman streams (1 / 4)
  int64_t regen, lcomp;  // regenerated size; compressed size incl. the table (Huffman types)
};
// Literals section header of a compressed block b[0, bsize) (RFC 8878 3.1.1.3.1.1), bounds included.
ZS_HD int literals_header(const uint8_t* b, int64_t bsize, LitHdr* h) {
  const int ltype = b[0] & 3, sf = (b[0] >> 2) & 3;
  h->ltype = ltype;
  h->streams = 1;
  h->lcomp = 0;
  if (ltype < 2) {  // raw / RLE
    if (sf == 0 || sf == 2) { h->lh = 1; h->regen = b[0] >> 3; }
    else if (sf == 1) { h->lh = 2; if (bsize < 2) return ZS_FAIL(); h->regen = (b[0] >> 4) + ((int64_t)b[1] << 4); }
    else { h->lh = 3; if (bsize < 3) return ZS_FAIL(); h->regen = (b[0] >> 4) + ((int64_t)b[1] << 4) + ((int64_t)b[2] << 12); }
    if (h->regen > kMaxBlock) return ZS_FAIL();
    if (h->lh + (ltype == 0 ? h->regen : 1) > bsize) return ZS_FAIL();
    return ZS_OK;
  }
  if (sf == 0 || sf == 1) {
    h->lh = 3; if (bsize < 3) return ZS_FAIL();
    const uint32_t v = rd_le(b, 3);
    h->regen = (v >> 4) & 0x3FF; h->lcomp = v >> 14;
    h->streams = sf == 0 ? 1 : 4;
  } else if (sf == 2) {
    h->lh = 4; if (bsize < 4) return ZS_FAIL();
    const uint32_t v = rd_le(b, 4);
    h->regen = (v >> 4) & 0x3FFF; h->lcomp = v >> 18;
    h->streams = 4;
  } else {
    h->lh = 5; if (bsize < 5) return ZS_FAIL();
    const uint64_t v = (uint64_t)rd_le(b, 4) | ((uint64_t)b[4] << 32);
    h->regen = (int64_t)((v >> 4) & 0x3FFFF); h->lcomp = (int64_t)(v >> 22);
    h->streams = 4;
  }
  if (h->regen > kMaxBlock || h->lh + h->lcomp > bsize) return ZS_FAIL();
  // a Huffman code spends at least one bit per symbol: a section that claims more than 8 literals per compressed byte is
  // corrupt.  The single-pass literal scratch (zstd_decompress.hip: min(8 * partition bytes + 256, kMaxBlock) + 64 per
  // buffer) relies on this bound, so it is enforced here, where every caller passes.
  if (h->regen > 8 * h->lcomp) return ZS_FAIL();
  return ZS_OK;
}

// The Huffman-coded literals of one block: table (ltype 2) or the frame's previous one (3), then the streams - stream k on
// lane k (device); the host walks them one after the other.  hp[0, hleft) = table description + streams.  Every lane
// returns the same code.
ZS_HD int block_literals(HufWork& h, const LitHdr& lh, const uint8_t* hp, int64_t hleft, uint8_t* lit_buf, Lanes L) {
  if (lh.ltype == 2) {
    const int used = read_huf_table(h, hp, hleft);
    if (used < 0) return used;
    hp += used;
    hleft -= used;
#ifdef S3S_ZSTD_DEVICE
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // (the table is in LDS before lanes 0..3 read it)
#endif
  } else if (!h.have_huf) {
    return ZS_FAIL();
  }
  const int64_t regen = lh.regen;
  int rc = ZS_OK;
  if (lh.streams == 1) {
    if (L.lane == 0) rc = huf_decode_stream(h, hp, hleft, lit_buf, regen);
  } else {
    if (hleft < 6) return ZS_FAIL();
    const int64_t s1 = rd_le(hp, 2), s2 = rd_le(hp + 2, 2), s3 = rd_le(hp + 4, 2);
    const int64_t s4 = hleft - 6 - s1 - s2 - s3;
    if (s4 < 0) return ZS_FAIL();
    const int64_t q = (regen + 3) / 4;
    const int64_t n4 = regen - 3 * q;
    if (n4 < 0) return ZS_FAIL();
    const uint8_t* sp = hp + 6;
    for (int k = 0; k < 4; k++) {
      const int64_t sz = k == 0 ? s1 : k == 1 ? s2 : k == 2 ? s3 : s4;
      const int64_t off = k == 0 ? 0 : k == 1 ? s1 : k == 2 ? s1 + s2 : s1 + s2 + s3;
      if (L.lane == k % L.n) {
        const int r1 = huf_decode_stream(h, sp + off, sz, lit_buf + k * q, k == 3 ? n4 : q);
        if (r1 != ZS_OK) rc = r1;
      }
    }
  }
#ifdef S3S_ZSTD_DEVICE
  {  // the first failing lane's code, to every lane
    const unsigned long long bad = __ballot(rc != ZS_OK);
    if (bad) rc = __shfl(rc, __builtin_ctzll(bad));
  }
#endif
  return rc;
}

// ---- hand-over between the two sides (device) ------------------------------------------------------------------------------------
#ifdef S3S_ZSTD_DEVICE
ZS_HD int32_t pipe_get(const int32_t* p) { return (int32_t)ZS_UNI32(__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)); }
ZS_HD void pipe_set(int32_t* p, int32_t v, Lanes L) {  // what this wavefront stored to memory before is visible to whoever sees v
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if (L.lane == 0) __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// sequence side: the literals of Huffman block number `want - 1` are in their buffer (or the literal side has failed)
// kPipeSpins: a waiter gives up after this many polls (~0.2 us each: seconds, where a block's literals take a millisecond or
// two) - the two sides cannot wait for each other forever by construction, and this makes a hang impossible by arithmetic too.
constexpr int32_t kPipeSpins = 1 << 23;
ZS_HD int pipe_wait_ready(LitPipe& lp, int32_t want) {
  for (int32_t spins = 0; spins < kPipeSpins; spins++) {
    if (pipe_get(&lp.ready) >= want) return ZS_OK;
    const int32_t e = pipe_get(&lp.err);
    if (e != ZS_OK) return pipe_get(&lp.ready) >= want ? ZS_OK : e;  // (err is set after the last ready)
    __builtin_amdgcn_s_sleep(8);
  }
  return ZS_FAIL();
}
#endif

#ifdef S3S_ZSTD_DEVICE
// ---- the lean sequence loop (device, round 4) ------------------------------------------------------------------------------------
// The sequence loop of decode_frame carries the whole frame's state, and hipcc - which cannot prove anything uniform that came
// through a flat pointer - turns it into ~300 instructions per sequence, most of them vector instructions under exec masks; the
// kernel is bound by instruction issue, so that IS its speed.  This function is the same loop for the common case only, out of
// line (its own register allocation: nothing of the frame is live in it), with LDS and global pointers typed as such and every
// loaded value declared uniform: the scalar unit does the decoding, the vector unit only the copies.  It takes sequences for as
// long as they are plain - at most 57 bits, window inside the stream, literal run inside the literal window and at most 64
// bytes, match source not overlapping its own output (in the ring or, fenced, in memory), nothing wrong with it - and stops IN FRONT of the
// first one that is not (or of the block's last sequence): the compiled loop does that one, with all its checks, its error codes
// and its slow paths, and calls again.  State is committed per sequence, so stopping costs nothing but the call.
typedef __attribute__((address_space(3))) Work* WorkLds;
typedef __attribute__((address_space(1))) const uint8_t* GlobalIn;
typedef __attribute__((address_space(1))) uint8_t* GlobalOut;
__device__ __attribute__((noinline)) void seq_fast(WorkLds w, int lane) {
  const GlobalIn bits = (GlobalIn)w->fast.bits;
  const GlobalOut bdst = (GlobalOut)w->fast.bdst;
  const int32_t bits_size = (int32_t)ZS_UNI32(w->fast.bits_size);
  int32_t pos = (int32_t)ZS_UNI32(w->fast.pos), cbase = (int32_t)ZS_UNI32(w->fast.cbase);
  uint64_t cache = (uint64_t)ZS_UNI32(w->fast.cache_lo) | ((uint64_t)ZS_UNI32(w->fast.cache_hi) << 32);
  uint32_t sl = ZS_UNI32(w->fast.sl), so = ZS_UNI32(w->fast.so), sm = ZS_UNI32(w->fast.sm);
  uint32_t rep0 = w->fast.rep0, rep1 = w->fast.rep1, rep2 = w->fast.rep2;  // (vector registers: see the loop)
  uint32_t i = ZS_UNI32(w->fast.i);
  const uint32_t nseq = ZS_UNI32(w->fast.nseq);
  uint32_t lit_pos = ZS_UNI32(w->fast.lit_pos), bout = ZS_UNI32(w->fast.bout);
  const uint32_t regen = ZS_UNI32(w->fast.regen), bcap = ZS_UNI32(w->fast.bcap), rq0 = ZS_UNI32(w->fast.rq0);
  const int32_t litw_base = (int32_t)ZS_UNI32(w->fast.litw_base);
  const uint32_t litw_at = litw_base < 0 ? 0x40000000u : (uint32_t)litw_base;
  uint32_t visible = ZS_UNI32(w->fast.visible);
  const uint64_t hist = (uint64_t)ZS_UNI32(w->fast.hist_lo) | ((uint64_t)ZS_UNI32(w->fast.hist_hi) << 32);
  const uint32_t hist30 = hist > 0x3fffffffull ? 0x3fffffffu : (uint32_t)hist;  // (history beyond 1 GiB: no offset reaches further)
  while (i + 1 < nseq) {
    const uint32_t el = w->ll[sl], eo = w->of[so], em = w->ml[sm];  // (uniform, but left in vector registers: the field
    const uint32_t vl = w->llv[sl], vm = w->mlv[sm];                // arithmetic below then runs on the vector unit, the checks on the scalar one)
    const uint32_t co = eo >> 24, bm = vm >> 24, bl = vl >> 24;
    const uint32_t nl = (el >> 16) & 0xff, nm = (em >> 16) & 0xff, no = (eo >> 16) & 0xff;
    const uint32_t nb = ZS_UNI32(co + bm + bl + nl + nm + no);
    // (conditions as sign bits of differences, see "plain" below)
    if ((int32_t)((31u - ZS_UNI32(co)) | (57u - nb)) < 0) break;
    const int32_t npos = pos - (int32_t)nb;
    if (((npos - cbase) | (cbase + 64 - pos)) < 0) {  // the cursor's window is not the cached one: the window whose top byte holds it
      const int32_t b0 = ((pos - 1) >> 3) - 7;
      if ((b0 | (bits_size - 8 - b0)) < 0) break;  // (the stream's first and last bytes: the compiled loop's zero fill)
      uint32_t lo, hi;
      __builtin_memcpy(&lo, (const void*)(bits + b0), 4);
      __builtin_memcpy(&hi, (const void*)(bits + b0 + 4), 4);
      cache = (uint64_t)ZS_UNI32(lo) | ((uint64_t)ZS_UNI32(hi) << 32);
      cbase = b0 * 8;
    }
    // (npos < 0 - the stream read below its first bit - is one of the "no" bits below; the shift is 64 only when the sequence
    //  has no bits at all, and then nothing is taken from fb)
    const uint64_t fb = cache >> ((uint32_t)(npos - cbase) & 63u);
    uint32_t rem = nb;
    auto take = [&](uint32_t n) -> uint32_t {
      rem -= n;
      return (uint32_t)(fb >> rem) & ((1u << n) - 1u);
    };
    const uint32_t ov = take(co);
    const uint32_t mlen = ZS_UNI32((vm & 0xFFFFFFu) + take(bm));
    const uint32_t llen = ZS_UNI32((vl & 0xFFFFFFu) + take(bl));
    const uint32_t nsl = ZS_UNI32((el & 0xFFFFu) + take(nl)), nsm = ZS_UNI32((em & 0xFFFFu) + take(nm)), nso = ZS_UNI32((eo & 0xFFFFu) + take(no));
    const uint32_t oval = (1u << co) + ov;
    // (the repeat-offset history too is vector arithmetic on values every lane holds: the two units share the work, and this
    //  kernel is bound by what they can issue.  rep0..2 live in vector registers; the offset comes back to the scalar side.)
    const bool is_rep = oval <= 3;
    const uint32_t idx = oval - 1 + (llen == 0 ? 1u : 0u);
    uint32_t rv = rep0;
    rv = idx == 1 ? rep1 : rv;
    rv = idx == 2 ? rep2 : rv;
    rv = idx == 3 ? rep0 - 1 : rv;
    const uint32_t offset_v = is_rep ? rv : oval - 3;
    const bool change = !is_rep || idx != 0, deep = !is_rep || idx >= 2;
    const uint32_t n2 = deep ? rep1 : rep2, n1 = change ? rep0 : rep1, n0 = change ? offset_v : rep0;
    const uint32_t offset = ZS_UNI32(offset_v);
    const uint32_t reach = bout + llen, nbout = reach + mlen;
    // plain enough?  (anything wrong with the sequence is "not plain": the compiled loop names the error.)  Every condition is
    // a difference whose sign bit says "no" - one OR chain and one branch instead of a compare, a select and a branch each; all
    // quantities are below 2^30 here except an offset out of range, which then reads as "no" too.
    const uint32_t lrel = lit_pos - litw_at;  // (no window loaded: litw_at is 2^30, lrel wraps to something huge)
    const uint32_t lit_no = (64u - llen) | lrel | ((uint32_t)kLitW - (lrel + llen));
    const uint32_t no_bits = (uint32_t)npos | (llen != 0 ? lit_no : 0u) | (offset - 1u) | (offset - mlen) | (regen - (lit_pos + llen)) |
                             ((uint32_t)kMaxBlock - nbout) | (bcap - nbout) | (reach + hist30 - offset);
    const bool plain = (int32_t)no_bits >= 0;
    if (!plain) break;
    const uint32_t rq = rq0 + bout;
    if ((uint32_t)lane < llen) {
      const uint8_t v = w->litw[lrel + (uint32_t)lane];
      w->ring[(rq + (uint32_t)lane) & (kRing - 1)] = v;
      bdst[bout + (uint32_t)lane] = v;
    }
    const uint32_t rqm = rq0 + reach;
    if (offset <= (uint32_t)kRing && offset + mlen <= (uint32_t)kRing) {  // the source is still in the ring
      const uint32_t rs = rqm - offset;
      for (uint32_t j = (uint32_t)lane; j < mlen; j += kWaveLanes) {
        const uint8_t v = w->ring[(rs + j) & (kRing - 1)];
        w->ring[(rqm + j) & (kRing - 1)] = v;
        bdst[reach + j] = v;
      }
    } else {  // from memory (two thirds of a level-1 TeraSort frame's matches): what was stored since the last fence must have landed
      if (offset < nbout - visible) {  // (the source ends at reach - offset + mlen; it does not overlap its output here)
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        visible = reach;
      }
      const GlobalIn from = (GlobalIn)bdst + ((int64_t)reach - (int64_t)offset);
      for (uint32_t j = (uint32_t)lane; j < mlen; j += kWaveLanes) {
        const uint8_t v = from[j];
        w->ring[(rqm + j) & (kRing - 1)] = v;
        bdst[reach + j] = v;
      }
    }
    rep2 = n2;
    rep1 = n1;
    rep0 = n0;
    sl = nsl;
    sm = nsm;
    so = nso;
    pos = npos;
    lit_pos += llen;
    bout = nbout;
    i++;
  }
  if (lane == 0) {
    w->fast.pos = pos;
    w->fast.cbase = cbase;
    w->fast.cache_lo = (uint32_t)cache;
    w->fast.cache_hi = (uint32_t)(cache >> 32);
    w->fast.sl = sl;
    w->fast.so = so;
    w->fast.sm = sm;
    w->fast.rep0 = rep0;
    w->fast.rep1 = rep1;
    w->fast.rep2 = rep2;
    w->fast.i = i;
    w->fast.lit_pos = lit_pos;
    w->fast.bout = bout;
    w->fast.visible = visible;
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
}
#endif

// Decodes (execute = true) or only sizes (execute = false) the frame at src[0, size).  dst = where this frame's output
// starts (history never reaches in front of it), cap = bytes available there.  lit_buf: literal scratch — the host model
// decodes a block's Huffman literals right here into lit_buf (kMaxBlock + 32 bytes, lit_stride 0); on the device they come
// from the literal wavefront through lit_buf + (hblock & 1) * lit_stride.  hblock: Huffman-coded blocks of the partition so
// far (runs on over the partition's frames; both sides count alike).
ZS_HD int decode_frame(Work& w, LitPipe& lp, const uint8_t* src, int64_t size, uint8_t* dst, int64_t cap, bool execute,
                       uint8_t* lit_buf, int64_t lit_stride, int32_t& hblock, Lanes L, FrameOut* out) {
  FrameHdr fh;
  {
    const int rc = frame_header(src, size, &fh);
    if (rc != ZS_OK) return rc;
  }
  if (fh.skippable) {
    out->consumed = fh.bytes;
    out->produced = 0;
    out->lit_need = 0;
    return ZS_OK;
  }
  int64_t ip = fh.bytes;
  const int64_t fcs = fh.fcs;
  const int has_check = fh.has_check;

  uint32_t rep0 = 1, rep1 = 4, rep2 = 8;
  int64_t lit_need = 0;
#ifndef S3S_ZSTD_DEVICE
  lp.h.have_huf = 0;
#endif
  w.have_ll = w.have_ml = w.have_of = 0;
  w.vals_ll = w.vals_ml = 0;
  int64_t op = 0;          // bytes produced in this frame
  for (;;) {
    if (ip + 3 > size) return ZS_FAIL();
    const uint32_t bh = rd_le(src + ip, 3);
    ip += 3;
    const int last = (int)(bh & 1), type = (int)((bh >> 1) & 3);
    const int64_t bsize = bh >> 3;
    if (type == 3) return ZS_FAIL();
    if (type == 0) {  // raw
      if (bsize > kMaxBlock || ip + bsize > size) return ZS_FAIL();
      if (execute) {
        if (op + bsize > cap) return ZS_CAPACITY;
        put_plain(w, dst + op, (uint32_t)op, src + ip, (uint32_t)bsize, L);
        ZS_FENCE();  // (every block ends with one: a later block's far match may read this output from memory)
      }
      ip += bsize;
      op += bsize;
    } else if (type == 1) {  // RLE: bsize copies of one byte
      if (bsize > kMaxBlock || ip + 1 > size) return ZS_FAIL();
      if (execute) {
        if (op + bsize > cap) return ZS_CAPACITY;
        put_fill(w, dst + op, (uint32_t)op, src[ip], (uint32_t)bsize, L);
        ZS_FENCE();
      }
      ip += 1;
      op += bsize;
    } else {
      if (bsize > kMaxBlock || bsize < 2 || ip + bsize > size) return ZS_FAIL();
      const uint8_t* b = src + ip;
      const uint8_t* const bend = b + bsize;
      // ---- literals section ----
      LitHdr lh;
      {
        const int rc = literals_header(b, bsize, &lh);
        if (rc != ZS_OK) return rc;
      }
      const int64_t regen = lh.regen;
      const uint8_t* lit = nullptr;   // where the regenerated literals are (raw: inside src)
      int lit_rle = -1;
      const bool huf_block = lh.ltype >= 2;
      if (lh.ltype == 0) {
        lit = b + lh.lh;
        b += lh.lh + regen;
      } else if (lh.ltype == 1) {
        lit_rle = b[lh.lh];
        b += lh.lh + 1;
      } else {  // Huffman-compressed (2) / treeless (3)
        lit_need = regen > lit_need ? regen : lit_need;
        uint8_t* lb = lit_buf + (hblock & 1) * lit_stride;
#ifndef S3S_ZSTD_DEVICE
        if (execute) {
          const int rc = block_literals(lp.h, lh, b + lh.lh, lh.lcomp, lb, L);
          if (rc != ZS_OK) return rc;
        }
#endif
        lit = lb;
        b += lh.lh + lh.lcomp;
      }
      // ---- sequences section ----
      if (b >= bend) return ZS_FAIL();
      int64_t nseq = b[0];
      if (nseq == 0) {
        b += 1;
      } else if (nseq < 128) {
        b += 1;
      } else if (nseq < 255) {
        if (b + 2 > bend) return ZS_FAIL();
        nseq = ((nseq - 128) << 8) + b[1];
        b += 2;
      } else {
        if (b + 3 > bend) return ZS_FAIL();
        nseq = (int64_t)b[1] + ((int64_t)b[2] << 8) + 0x7F00;
        b += 3;
      }
      // Inside a block everything is 32 bits and relative to the block (round 4: 64-bit positions per sequence were a tenth
      // of the loop): bout = bytes this block has produced, lit_pos = literals it has used.
      const int64_t bop = op;  // output position at the start of the block
      uint8_t* const bdst = dst + bop;
      const uint32_t rq0 = (uint32_t)bop;
      const uint32_t regen32 = (uint32_t)regen;
      const uint32_t bcap = !execute ? 0u : (cap - bop > 0x7fffffffll ? 0x7fffffffu : (uint32_t)(cap - bop));
      uint32_t bout = 0, lit_pos = 0;
      uint32_t visible = 0;  // (a fence follows every block: at its start everything before it is readable)
      int32_t litw_base = -1;
      if (nseq > 0) {
        if (b >= bend) return ZS_FAIL();
        const int modes = b[0];
        b += 1;
        if (modes & 3) return ZS_FAIL();
        for (int t = 0; t < 3; t++) {  // LL, OF, ML
          const int mode = (modes >> (6 - 2 * t)) & 3;
          uint32_t* tab = t == 0 ? w.ll : t == 1 ? w.of : w.ml;
          int32_t& tlog = t == 0 ? w.ll_log : t == 1 ? w.of_log : w.ml_log;
          int32_t& have = t == 0 ? w.have_ll : t == 1 ? w.have_of : w.have_ml;
          const int max_sym = t == 0 ? 35 : t == 1 ? 31 : 52, max_log = t == 0 ? kLLLogMax : t == 1 ? kOFLogMax : kMLLogMax;
          if (mode == 0) {
            int log;
            const int ms = default_norm(t, w.norm, &log);
            if (build_fse(tab, w.norm, ms, log, w.next) != ZS_OK) return ZS_FAIL();
            tlog = log;
            have = 1;
          } else if (mode == 1) {
            if (b >= bend) return ZS_FAIL();
            if (b[0] > max_sym) return ZS_FAIL();
            build_rle(tab, b[0]);
            tlog = 0;
            have = 1;
            b += 1;
          } else if (mode == 2) {
            int ms, log;
            const int used = read_ncount(w.norm, max_sym, max_log, b, bend - b, &ms, &log);
            if (used < 0) return used;
            if (build_fse(tab, w.norm, ms, log, w.next) != ZS_OK) return ZS_FAIL();
            tlog = log;
            have = 1;
            b += used;
          } else if (!have) {
            return ZS_FAIL();
          }
        }
        if (modes >> 6 != 3 || !w.vals_ll)  // (a repeated table keeps its values)
          for (int u = L.lane; u < (1 << w.ll_log); u += L.n) {
            const int c = (int)(w.ll[u] >> 24);
            w.llv[u] = ll_base(c) | ((uint32_t)ll_bits(c) << 24);
          }
        if (((modes >> 2) & 3) != 3 || !w.vals_ml)
          for (int u = L.lane; u < (1 << w.ml_log); u += L.n) {
            const int c = (int)(w.ml[u] >> 24);
            w.mlv[u] = ml_base(c) | ((uint32_t)ml_bits(c) << 24);
          }
        w.vals_ll = w.vals_ml = 1;
#ifdef S3S_ZSTD_DEVICE
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // (lane-strided LDS fills above, read by every lane below: one
                                                                // wavefront, its LDS accesses complete in order - no s_barrier, the
                                                                // workgroup's other wavefront is the literal side and never comes)
#endif
#ifdef S3S_ZSTD_DEVICE
        if (execute && huf_block) {  // the literal wavefront's hand-over, as late as possible: the tables above did not need it
          const int rc = pipe_wait_ready(lp, hblock + 1);
          if (rc != ZS_OK) return rc;
        }
#endif
        BitR r;
        if (!bitr_init(r, b, bend - b)) return ZS_FAIL();
        uint32_t sl = bitr_read(r, w.ll_log), so = bitr_read(r, w.of_log), sm = bitr_read(r, w.ml_log);
        if (r.pos < 0) return ZS_FAIL();
#ifdef S3S_ZSTD_DEVICE
        int fast_pause = 0;  // sequences to take here before the lean loop is tried again (it made no progress last time)
#endif
        for (int64_t i = 0; i < nseq; i++) {
#ifdef S3S_ZSTD_DEVICE
          if (execute && lit_rle < 0 && i + 1 < nseq) {  // plain sequences: the lean loop (seq_fast), this one takes the rest
            if (fast_pause > 0) {
              fast_pause--;
            } else {
              Work::Fast& f = w.fast;
              f.bits = r.p;
              f.bdst = bdst;
              f.bits_size = r.size;
              f.pos = r.pos;
              f.cbase = r.cbase;
              f.cache_lo = (uint32_t)r.cache;
              f.cache_hi = (uint32_t)(r.cache >> 32);
              f.sl = sl;
              f.so = so;
              f.sm = sm;
              f.rep0 = rep0;
              f.rep1 = rep1;
              f.rep2 = rep2;
              f.i = (uint32_t)i;
              f.nseq = (uint32_t)nseq;
              f.lit_pos = lit_pos;
              f.bout = bout;
              f.regen = regen32;
              f.bcap = bcap;
              f.rq0 = rq0;
              f.litw_base = litw_base;
              f.visible = visible;
              f.hist_lo = (uint32_t)(uint64_t)bop;
              f.hist_hi = (uint32_t)((uint64_t)bop >> 32);
              __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
              seq_fast((WorkLds)&w, L.lane);
              const uint32_t ni = ZS_UNI32(f.i);
              if (ni == (uint32_t)i) {
                fast_pause = 3;
              } else {
                i = ni;
                r.pos = (int32_t)ZS_UNI32(f.pos);
                r.cbase = (int32_t)ZS_UNI32(f.cbase);
                r.cache = (uint64_t)ZS_UNI32(f.cache_lo) | ((uint64_t)ZS_UNI32(f.cache_hi) << 32);
                sl = ZS_UNI32(f.sl);
                so = ZS_UNI32(f.so);
                sm = ZS_UNI32(f.sm);
                rep0 = ZS_UNI32(f.rep0);
                rep1 = ZS_UNI32(f.rep1);
                rep2 = ZS_UNI32(f.rep2);
                lit_pos = ZS_UNI32(f.lit_pos);
                bout = ZS_UNI32(f.bout);
                visible = ZS_UNI32(f.visible);
              }
            }
          }
#endif
          const uint32_t el = ZS_UNI32(w.ll[sl]), eo = ZS_UNI32(w.of[so]), em = ZS_UNI32(w.ml[sm]);
          const uint32_t vl = ZS_UNI32(w.llv[sl]), vm = ZS_UNI32(w.mlv[sm]);
          const int co = (int)(eo >> 24);
          if (co > 31) return ZS_FAIL();
          uint32_t ov;
          uint32_t mlen, llen;  // (at most 131 074 + 65 535 and 65 536 + 65 535)
          // (round 4, measured +8.5 % on TeraSort frames, profiles/r04a_first_call.txt) All fields of a sequence — offset, match-length and literal-length extra bits,
          // then the three state updates — from ONE window: a refill puts at least 57 bits below the cursor into the
          // cache, and a sequence of a level-1 stream needs far fewer; the six reads become shifts of one register.
          const int bm = (int)(vm >> 24), bl = (int)(vl >> 24);
          const bool more = i + 1 < nseq;
          const int nl = more ? (int)((el >> 16) & 0xff) : 0, nm = more ? (int)((em >> 16) & 0xff) : 0,
                    no = more ? (int)((eo >> 16) & 0xff) : 0;
          const int nb = co + bm + bl + nl + nm + no;
          if (nb <= 57) {
            if (!(r.pos - nb >= r.cbase && r.pos <= r.cbase + 64)) {  // the window whose top byte holds the cursor: >= 57 bits below it
              const int32_t top = (r.pos - 1) >> 3;
              r.cbase = (top - 7) * 8;
              r.cache = load_bits64(r.p, r.size, top - 7);
#ifdef S3S_ZSTD_DEVICE
              r.cache = (uint64_t)ZS_UNI32((uint32_t)r.cache) | ((uint64_t)ZS_UNI32((uint32_t)(r.cache >> 32)) << 32);
#endif
            }
            const int sh = r.pos - nb - r.cbase;  // 0 .. 64 (64 only when nb == 0 and the cursor sits on the cache's top bit)
            uint64_t bits = sh < 64 ? r.cache >> sh : 0;  // the nb bits of this sequence, first field on top
            int rem = nb;
            auto take = [&](int n) -> uint32_t {
              rem -= n;
              return (uint32_t)(bits >> rem) & ((1u << n) - 1u);  // (every field is at most 31 bits)
            };
            ov = take(co);
            mlen = (vm & 0xFFFFFFu) + take(bm);
            llen = (vl & 0xFFFFFFu) + take(bl);
            sl = (el & 0xFFFFu) + take(nl);  // (after the last sequence: 0 bits each, states nobody reads)
            sm = (em & 0xFFFFu) + take(nm);
            so = (eo & 0xFFFFu) + take(no);
            r.pos -= nb;
          } else
          {
          // extra bits: offset, match length, literal length
          if (co > 24) {  // more than 32 bits cannot be peeked at once: two reads
            const uint32_t hi = bitr_read(r, co - 16);
            ov = (hi << 16) | bitr_read(r, 16);
          } else {
            ov = bitr_read(r, co);
          }
          mlen = (vm & 0xFFFFFFu) + bitr_read(r, (int)(vm >> 24));
          llen = (vl & 0xFFFFFFu) + bitr_read(r, (int)(vl >> 24));
          if (i + 1 < nseq) {  // state updates: LL, ML, OF
            sl = (el & 0xFFFFu) + bitr_read(r, (int)((el >> 16) & 0xff));
            sm = (em & 0xFFFFu) + bitr_read(r, (int)((em >> 16) & 0xff));
            so = (eo & 0xFFFFu) + bitr_read(r, (int)((eo >> 16) & 0xff));
          }
          }
          const uint32_t oval = (1u << co) + ov;  // (co <= 31 and ov < 2^co: fits)
          // Strict (RFC 8878 §3.1.1.3.2.1.2: the stream ends exactly at its first bit): libzstd >= 1.4.5 lets a damaged
          // stream read below its start — what it returns there depends on its 64-bit container's state — and only demands
          // that no bits are left over; such streams are refused here ("Stream is corrupted") instead of decoded to
          // whatever the container held.  Valid streams never get here, and damaged ones were stopped by the partition's
          // Adler32 / CRC32 before the decoder saw them.
          uint32_t bad = r.pos < 0 ? 1u : 0u;
          // offset with the repeat history.  Three scalars on purpose: an array indexed by the repeat code lives in scratch
          // memory on the device - a store and a load round trip on every sequence's critical path.  Selects instead of
          // branches, and ONE exit for everything that can be wrong with a sequence: a branch costs this loop an exec-mask
          // save, a jump and a restore, and there were seven of them per sequence.
          //   new offset (oval > 3):   rep = {offset, rep0, rep1}
          //   repeat code idx 0:       unchanged;   idx 1: {rep1, rep0, rep2};   idx 2: {rep2, rep0, rep1};   idx 3: {rep0 - 1, rep0, rep1}
          const bool is_rep = oval <= 3;
          const uint32_t idx = oval - 1 + (llen == 0 ? 1u : 0u);  // 0..3 (meaningful when is_rep)
          uint32_t rv = rep0;
          rv = idx == 1 ? rep1 : rv;
          rv = idx == 2 ? rep2 : rv;
          rv = idx == 3 ? rep0 - 1 : rv;
          const uint32_t offset = is_rep ? rv : oval - 3;  // (oval < 2^32: the offset code is at most 31)
          const bool change = !is_rep || idx != 0, deep = !is_rep || idx >= 2;
          rep2 = deep ? rep1 : rep2;
          rep1 = change ? rep0 : rep1;
          rep0 = change ? offset : rep0;
          const uint32_t reach = bout + llen;  // bytes of this block in front of the match
          const uint32_t nbout = reach + mlen;
          // (no dictionary: history starts with the frame; offset 0 is also what repeat code 3 gives when rep0 is 1)
          bad |= (offset == 0 ? 1u : 0u) | (lit_pos + llen > regen32 ? 1u : 0u) | (nbout > (uint32_t)kMaxBlock ? 1u : 0u) |
                 ((offset > reach ? 1u : 0u) & ((uint64_t)(offset - reach) > (uint64_t)bop ? 1u : 0u));
          const bool over = execute && nbout > bcap;
          if (bad | (over ? 1u : 0u)) {
            if (bad) return ZS_FAIL();
            return ZS_CAPACITY;
          }
#ifdef ZS_STATS_HOOK
          ZS_STATS_HOOK((int64_t)offset, (int64_t)mlen, (int64_t)llen);
#endif
          if (execute) {
            if (llen) {
              if (lit_rle >= 0) put_fill(w, bdst + bout, rq0 + bout, (uint8_t)lit_rle, llen, L);
              else put_literals(w, bdst + bout, rq0 + bout, lit, regen32, lit_pos, llen, L, litw_base);
            }
            const bool near = offset <= (uint32_t)kRing && offset + mlen <= (uint32_t)kRing;
            if (!near) {  // a far source comes from global memory: what was stored since the last fence must have landed
              // (the source ends at reach - offset + mlen, or at reach when it overlaps its own output; everything in front of
              //  `visible` - bytes of this block behind which a fence has been issued, 0 = the block's start - is readable)
              const bool fence = offset >= mlen ? offset < nbout - visible : reach > visible;
              if (fence) {
                ZS_FENCE();
                visible = reach;
              }
            }
            put_match(w, bdst + reach, rq0 + reach, offset, mlen, near, L);
          }
          lit_pos += llen;
          bout = nbout;
        }
        if (r.pos != 0) return ZS_FAIL();  // the sequence stream must be consumed exactly
      } else {
        if (b != bend) return ZS_FAIL();
#ifdef S3S_ZSTD_DEVICE
        if (execute && huf_block) {  // (a block of literals only)
          const int rc = pipe_wait_ready(lp, hblock + 1);
          if (rc != ZS_OK) return rc;
        }
#endif
      }
      // literals behind the last sequence
      const uint32_t rest = regen32 - lit_pos;
      if (execute) {
        if (bout + rest > bcap) return ZS_CAPACITY;
        if (rest) {
          if (lit_rle >= 0) put_fill(w, bdst + bout, rq0 + bout, (uint8_t)lit_rle, rest, L);
          else put_literals(w, bdst + bout, rq0 + bout, lit, regen32, lit_pos, rest, L, litw_base);
        }
      }
      bout += rest;
      if (bout > (uint32_t)kMaxBlock) return ZS_FAIL();
      op = bop + bout;
      ip += bsize;
      if (huf_block) {
        hblock++;
#ifdef S3S_ZSTD_DEVICE
        if (execute) pipe_set(&lp.consumed, hblock, L);  // (a release: this block's reads of its literal buffer are done)
#endif
      }
      if (execute) ZS_FENCE();
    }
    if (last) break;
  }
  if (has_check) {
    if (ip + 4 > size) return ZS_FAIL();
    // Content_Checksum: the low 32 bits of XXH64 over the decoded frame (round 4: verified like libzstd does - without it a
    // frame of a checksumming writer that was damaged where it still decodes went through whenever the shuffle checksums
    // were off; the size pass has no bytes to hash, the decode pass of the same call does)
    if (execute && rd_le(src + ip, 4) != (uint32_t)xxh64_content(dst, op, L)) return ZS_FAIL();
    ip += 4;
  }
  if (fcs >= 0 && fcs != op) return ZS_FAIL();
  out->consumed = ip;
  out->produced = op;
  out->lit_need = lit_need;
  return ZS_OK;
}

// All frames of one partition (concatenated streams of a multi-spill map task).  *total = decoded bytes.
ZS_HD int decode_partition(Work& w, LitPipe& lp, const uint8_t* src, int64_t size, uint8_t* dst, int64_t cap, bool execute,
                           uint8_t* lit_buf, int64_t lit_stride, Lanes L, int64_t* total, int64_t* lit_need = nullptr) {
  int64_t ip = 0, op = 0, need = 0;
  int32_t hblock = 0;
  while (ip < size) {
    FrameOut fo;
    const int rc = decode_frame(w, lp, src + ip, size - ip, execute ? dst + op : dst, execute ? cap - op : 0, execute, lit_buf,
                                lit_stride, hblock, L, &fo);
    if (rc != ZS_OK) return rc;
    ip += fo.consumed;
    op += fo.produced;
    need = fo.lit_need > need ? fo.lit_need : need;
  }
  *total = op;
  if (lit_need) *lit_need = need;
  return ZS_OK;
}

#ifdef S3S_ZSTD_DEVICE
// ---- the literal wavefront (device) ------------------------------------------------------------------------------------------
// It walks the same frame and block headers as decode_partition, for each partition it serves, and regenerates the
// Huffman-coded literals of block k into buffer k & 1, one block ahead of that partition's sequence wavefront.  A malformed
// header simply ends a walk (the sequence side meets the same header and reports it); a bad table or stream is handed over
// through `err`.  Every way out of a walk other than the end of the partition sets `err`, so nobody is left waiting.
struct LitWalker {
  const uint8_t* src;
  int64_t size, ip;
  uint8_t* lit_buf;
  int64_t lit_stride;
  int32_t hblock;
  int in_frame, has_check;
  int done;          // nothing more to do for this partition
  int have;          // a Huffman block is waiting for its buffer: lh / hp describe it
  LitHdr lh;
  const uint8_t* hp;
  int64_t bsize;
};
// to the next Huffman-coded literals section (have = 1) or the end of the partition (done = 1)
ZS_HD void walker_advance(LitWalker& k, LitPipe& lp, Lanes L) {
  for (;;) {
    if (!k.in_frame) {
      if (k.ip >= k.size) { k.done = 1; return; }
      FrameHdr fh;
      if (frame_header(k.src + k.ip, k.size - k.ip, &fh) != ZS_OK) break;
      k.ip += fh.bytes;
      if (fh.skippable) continue;
      lp.h.have_huf = 0;
      k.in_frame = 1;
      k.has_check = fh.has_check;
    }
    if (k.ip + 3 > k.size) break;
    const uint32_t bh = rd_le(k.src + k.ip, 3);
    k.ip += 3;
    const int last = (int)(bh & 1), type = (int)((bh >> 1) & 3);
    const int64_t bsize = bh >> 3;
    if (type == 3 || bsize > kMaxBlock) break;
    if (last) {  // (what follows this block is the frame's end)
      k.in_frame = 0;
    }
    if (type == 1) {
      k.ip += 1;
    } else {
      if (k.ip + bsize > k.size) break;
      if (type == 2) {
        if (bsize < 2) break;
        if (literals_header(k.src + k.ip, bsize, &k.lh) != ZS_OK) break;
        if (k.lh.ltype >= 2) {
          k.hp = k.src + k.ip + k.lh.lh;
          k.have = 1;
        }
      }
      k.ip += bsize;
    }
    if (last && k.has_check) k.ip += 4;
    if (k.have) return;
  }
  pipe_set(&lp.err, ZS_BAD, L);
  k.done = 1;
}
// n_parts (1 or 2) partitions of the workgroup: whichever has a block to regenerate and a free buffer goes next
ZS_HD void literal_side(LitPipe* lps, LitWalker* ks, int n_parts, Lanes L) {
  int32_t idle = 0;
  for (;;) {
    bool any = false, progressed = false;
    for (int t = 0; t < n_parts; t++) {
      LitWalker& k = ks[t];
      LitPipe& lp = lps[t];
      if (k.done) continue;
      if (!k.have) {
        walker_advance(k, lp, L);
        if (k.done) continue;
      }
      any = true;
      if (pipe_get(&lp.quit)) { k.done = 1; continue; }            // its sequence side has left
      if (pipe_get(&lp.consumed) < k.hblock - 1) continue;         // buffer hblock & 1 still holds block hblock - 2
      // (second line of defence: the section must fit the buffer this partition was given, whatever sized it)
      const int rc = k.lh.regen > k.lit_stride - 64 ? (int)ZS_BAD
                                                    : block_literals(lp.h, k.lh, k.hp, k.lh.lcomp, k.lit_buf + (k.hblock & 1) * k.lit_stride, L);
      if (rc != ZS_OK) {
        pipe_set(&lp.err, rc, L);
        k.done = 1;
        continue;
      }
      k.hblock++;
      k.have = 0;
      pipe_set(&lp.ready, k.hblock, L);
      progressed = true;
    }
    if (!any) return;
    if (progressed) {
      idle = 0;
    } else {
      if (++idle >= kPipeSpins) {  // (see kPipeSpins: never in a working machine)
        for (int t = 0; t < n_parts; t++)
          if (!ks[t].done) pipe_set(&lps[t].err, ZS_BAD, L);
        return;
      }
      __builtin_amdgcn_s_sleep(8);
    }
  }
}
#endif

}  // namespace s3s_zstd
