/*
 * s3s_oracle_snappy.c — CPU restatement of the Snappy leg of the shuffle codec path.
 *
 * TEST INFRASTRUCTURE ONLY (see s3s_oracle.h).
 *
 * [EXT] Spark 3.5.5 SnappyCompressionCodec -> org.xerial.snappy SnappyOutputStream(out,
 * blockSize = spark.io.compression.snappy.blockSize = 32 KiB) -> JNI -> Google snappy
 * RawCompress.  Reference call sites: S3ShuffleReader.scala:59,108 (createCodec /
 * wrapStream); the compressed bytes arrive at S3ShuffleMapOutputWriter.scala:182-188.
 *
 * PARITY UNPINNED vs the JVM: snappy-java 1.1.10.x bundles snappy 1.1.10 whose
 * CompressFragment heuristics differ from 1.1.8; only libsnappy 1.1.8 exists in this
 * image, so the raw compressor below restates the 1.1.8 fragment compressor and is pinned
 * byte-for-byte against that library (tests/test_oracle_pins.py).  Streams are valid for
 * every snappy decoder; byte equality with a 1.1.10 JVM is not claimed.
 */
#include <string.h>

#include "s3s_oracle.h"

static inline uint32_t ld32(const uint8_t* p) {
  uint32_t v;
  memcpy(&v, p, 4);
  return v;
}
static inline uint64_t ld64(const uint8_t* p) {
  uint64_t v;
  memcpy(&v, p, 8);
  return v;
}

enum { SNAPPY_FRAGMENT = 65536, SNAPPY_MAX_TABLE = 1 << 14, SNAPPY_MIN_TABLE = 1 << 8 };

int s3o_snappy_max_compressed_length(int n) { return 32 + n + n / 6; }

static inline uint32_t sn_hash(uint32_t bytes, int shift) { return (bytes * 0x1e35a7bdu) >> shift; }

static int sn_log2_floor(uint32_t n) { return n == 0 ? -1 : 31 - __builtin_clz(n); }

static uint8_t* sn_emit_literal(uint8_t* op, const uint8_t* lit, int len) {
  int n = len - 1;
  if (n < 60) {
    *op++ = (uint8_t)(n << 2);
  } else {
    int count = (sn_log2_floor((uint32_t)n) >> 3) + 1;
    *op++ = (uint8_t)((59 + count) << 2);
    for (int i = 0; i < count; i++) *op++ = (uint8_t)(n >> (8 * i));
  }
  memcpy(op, lit, (size_t)len);
  return op + len;
}

static uint8_t* sn_emit_copy_upto64(uint8_t* op, int offset, int len, int len_lt_12) {
  if (len_lt_12 && offset < 2048) {
    *op++ = (uint8_t)(1 + ((len - 4) << 2) + ((offset >> 3) & 0xe0));
    *op++ = (uint8_t)(offset & 0xff);
  } else {
    *op++ = (uint8_t)(2 + ((len - 1) << 2));
    *op++ = (uint8_t)(offset & 0xff);
    *op++ = (uint8_t)(offset >> 8);
  }
  return op;
}

static uint8_t* sn_emit_copy(uint8_t* op, int offset, int len, int len_lt_12) {
  if (len_lt_12) return sn_emit_copy_upto64(op, offset, len, 1);
  while (len >= 68) {
    op = sn_emit_copy_upto64(op, offset, 64, 0);
    len -= 64;
  }
  if (len > 64) {
    op = sn_emit_copy_upto64(op, offset, 60, 0);
    len -= 60;
  }
  return sn_emit_copy_upto64(op, offset, len, len < 12);
}

/* number of equal bytes at s1/s2, s2 bounded by s2_limit */
static int sn_find_match_length(const uint8_t* s1, const uint8_t* s2, const uint8_t* s2_limit) {
  int matched = 0;
  while (s2 + 8 <= s2_limit) {
    uint64_t x = ld64(s1 + matched) ^ ld64(s2);
    if (x) return matched + (__builtin_ctzll(x) >> 3);
    s2 += 8;
    matched += 8;
  }
  while (s2 < s2_limit && s1[matched] == *s2) {
    s2++;
    matched++;
  }
  return matched;
}

/* One <=64 KiB fragment; table has table_size (power of two) zeroed u16 slots. */
static uint8_t* sn_compress_fragment(const uint8_t* input, int input_size, uint8_t* op,
                                     uint16_t* table, int table_size) {
  const uint8_t* ip = input;
  const int shift = 32 - sn_log2_floor((uint32_t)table_size);
  const uint8_t* ip_end = input + input_size;
  const uint8_t* base_ip = ip;
  const uint8_t* next_emit = ip;
  const int kInputMarginBytes = 15;

  if (input_size >= kInputMarginBytes) {
    const uint8_t* ip_limit = input + input_size - kInputMarginBytes;
    uint32_t next_hash = sn_hash(ld32(++ip), shift);
    for (;;) {
      /* scan forward for a 4-byte match; after 32 misses look at every 2nd byte, ... */
      uint32_t skip = 32;
      const uint8_t* next_ip = ip;
      const uint8_t* candidate;
      do {
        ip = next_ip;
        uint32_t hash = next_hash;
        uint32_t step = skip >> 5;
        skip += step;
        next_ip = ip + step;
        if (next_ip > ip_limit) goto emit_remainder;
        next_hash = sn_hash(ld32(next_ip), shift);
        candidate = base_ip + table[hash];
        table[hash] = (uint16_t)(ip - base_ip);
      } while (ld32(ip) != ld32(candidate));

      op = sn_emit_literal(op, next_emit, (int)(ip - next_emit));

      /* emit copies while the position right after each copy matches again */
      uint32_t cand_bytes, cur_bytes;
      do {
        const uint8_t* base = ip;
        int extra = sn_find_match_length(candidate + 4, ip + 4, ip_end);
        int matched = 4 + extra;
        ip += matched;
        int offset = (int)(base - candidate);
        op = sn_emit_copy(op, offset, matched, extra < 8);
        next_emit = ip;
        if (ip >= ip_limit) goto emit_remainder;
        /* refresh table[hash(ip-1)] and table[hash(ip)], probing the latter */
        uint32_t prev_hash = sn_hash(ld32(ip - 1), shift);
        table[prev_hash] = (uint16_t)(ip - base_ip - 1);
        cur_bytes = ld32(ip);
        uint32_t cur_hash = sn_hash(cur_bytes, shift);
        candidate = base_ip + table[cur_hash];
        cand_bytes = ld32(candidate);
        table[cur_hash] = (uint16_t)(ip - base_ip);
      } while (cur_bytes == cand_bytes);

      next_hash = sn_hash(ld32(ip + 1), shift);
      ++ip;
    }
  }
emit_remainder:
  if (next_emit < ip_end) op = sn_emit_literal(op, next_emit, (int)(ip_end - next_emit));
  return op;
}

int s3o_snappy_compress_block(const uint8_t* src, int n, uint8_t* dst, int cap) {
  if (n < 0 || cap < s3o_snappy_max_compressed_length(n)) return 0;
  uint8_t* op = dst;
  uint32_t v = (uint32_t)n; /* varint32 preamble */
  while (v >= 0x80) {
    *op++ = (uint8_t)(v | 0x80);
    v >>= 7;
  }
  *op++ = (uint8_t)v;
  static __thread uint16_t table[SNAPPY_MAX_TABLE];
  for (int pos = 0; pos < n; pos += SNAPPY_FRAGMENT) {
    int frag = n - pos < SNAPPY_FRAGMENT ? n - pos : SNAPPY_FRAGMENT;
    int tsize = SNAPPY_MIN_TABLE;
    while (tsize < SNAPPY_MAX_TABLE && tsize < frag) tsize <<= 1;
    memset(table, 0, (size_t)tsize * sizeof(uint16_t));
    op = sn_compress_fragment(src + pos, frag, op, table, tsize);
  }
  return (int)(op - dst);
}

/* Snappy format decoder (bounds-checked). Returns decoded size or <0. */
int s3o_snappy_decompress_block(const uint8_t* src, int src_len, uint8_t* dst, int dst_cap) {
  int ip = 0;
  uint32_t ulen = 0;
  int shift = 0;
  for (;;) {
    if (ip >= src_len || shift > 28) return -1;
    uint8_t b = src[ip++];
    ulen |= (uint32_t)(b & 0x7f) << shift;
    if (!(b & 0x80)) break;
    shift += 7;
  }
  if (ulen > (uint32_t)dst_cap) return -1;
  int op = 0;
  while (ip < src_len) {
    const uint8_t tag = src[ip++];
    int len, offset;
    switch (tag & 3) {
      case 0: {
        len = (tag >> 2) + 1;
        if (len > 60) {
          int nb = len - 60;
          if (src_len - ip < nb) return -1;
          uint32_t l = 0;
          for (int i = 0; i < nb; i++) l |= (uint32_t)src[ip + i] << (8 * i);
          ip += nb;
          if (l >= 0x7fffffffu) return -1;
          len = (int)l + 1;
        }
        if (len > src_len - ip || (uint32_t)len > ulen - (uint32_t)op) return -1;
        memcpy(dst + op, src + ip, (size_t)len);
        ip += len;
        op += len;
        continue;
      }
      case 1:
        if (src_len - ip < 1) return -1;
        len = ((tag >> 2) & 7) + 4;
        offset = ((tag >> 5) << 8) | src[ip];
        ip += 1;
        break;
      case 2:
        if (src_len - ip < 2) return -1;
        len = (tag >> 2) + 1;
        offset = src[ip] | (src[ip + 1] << 8);
        ip += 2;
        break;
      default:
        if (src_len - ip < 4) return -1;
        len = (tag >> 2) + 1;
        offset = (int)ld32(src + ip);
        ip += 4;
        break;
    }
    if (offset <= 0 || offset > op || (uint32_t)len > ulen - (uint32_t)op) return -1;
    for (int i = 0; i < len; i++) dst[op + i] = dst[op - offset + i];
    op += len;
  }
  return (uint32_t)op == ulen ? op : -1;
}

/* ---- snappy-java SnappyOutputStream / SnappyInputStream framing ---------------------- */
/* header: 0x82 'S' 'N' 'A' 'P' 'P' 'Y' 0x00 | version=1 (i32 BE) | compatible=1 (i32 BE);  */
/* then per chunk: compressedSize (i32 BE) | raw snappy.  No terminator.                    */
static const uint8_t SNJ_HEADER[16] = {0x82, 'S', 'N', 'A', 'P', 'P', 'Y', 0, 0, 0, 0, 1, 0, 0, 0, 1};
enum { SNJ_MIN_BLOCK = 1024 };

int64_t s3o_snappy_max_stream_size(int64_t ulen, int block_size) {
  if (ulen <= 0) return 0;
  if (block_size < SNJ_MIN_BLOCK) block_size = SNJ_MIN_BLOCK;
  int64_t full = ulen / block_size, rem = ulen % block_size;
  int64_t t = 16 + full * (4 + s3o_snappy_max_compressed_length(block_size));
  if (rem) t += 4 + s3o_snappy_max_compressed_length((int)rem);
  return t;
}

int64_t s3o_snappy_compress_stream(const uint8_t* src, int64_t ulen, int block_size,
                                   uint8_t* dst, int64_t dst_cap) {
  if (ulen < 0 || block_size <= 0) return S3O_E_INVALID;
  if (block_size < SNJ_MIN_BLOCK) block_size = SNJ_MIN_BLOCK; /* Math.max(MIN_BLOCK_SIZE, ..) */
  if (ulen == 0) return 0;
  if (dst_cap < s3o_snappy_max_stream_size(ulen, block_size)) return S3O_E_CAPACITY;
  int64_t op = 0;
  memcpy(dst, SNJ_HEADER, 16);
  op = 16;
  for (int64_t pos = 0; pos < ulen; pos += block_size) {
    int o = (int)(ulen - pos < block_size ? ulen - pos : block_size);
    int c = s3o_snappy_compress_block(src + pos, o, dst + op + 4, s3o_snappy_max_compressed_length(o));
    if (c <= 0) return S3O_E_INVALID;
    dst[op] = (uint8_t)(c >> 24);
    dst[op + 1] = (uint8_t)(c >> 16);
    dst[op + 2] = (uint8_t)(c >> 8);
    dst[op + 3] = (uint8_t)c;
    op += 4 + c;
  }
  return op;
}

/* SnappyInputStream: a chunk length equal to the first 4 magic bytes (0x82534e41) marks a
 * concatenated stream header, which is verified and skipped. */
int64_t s3o_snappy_decompress_stream(const uint8_t* src, int64_t clen, uint8_t* dst,
                                     int64_t dst_cap) {
  int64_t ip = 0, op = 0;
  if (clen == 0) return 0;
  if (clen < 16 || memcmp(src, SNJ_HEADER, 8) != 0) return S3O_E_BAD_FRAME;
  ip = 16;
  while (ip < clen) {
    if (clen - ip < 4) return S3O_E_BAD_FRAME;
    uint32_t c = ((uint32_t)src[ip] << 24) | ((uint32_t)src[ip + 1] << 16) |
                 ((uint32_t)src[ip + 2] << 8) | src[ip + 3];
    if (c == 0x82534e41u) {
      if (clen - ip < 16 || memcmp(src + ip, SNJ_HEADER, 8) != 0) return S3O_E_BAD_FRAME;
      ip += 16;
      continue;
    }
    ip += 4;
    if ((int64_t)c > clen - ip || c > 0x7fffffffu) return S3O_E_BAD_FRAME;
    int64_t room = dst_cap - op;
    int got = s3o_snappy_decompress_block(src + ip, (int)c, dst + op,
                                          room > 0x7fffffff ? 0x7fffffff : (int)room);
    if (got < 0) {
      /* distinguish capacity from corruption: peek the varint */
      uint32_t ulen = 0;
      int sh = 0;
      for (uint32_t i = 0; i < c && sh <= 28; i++, sh += 7) {
        ulen |= (uint32_t)(src[ip + i] & 0x7f) << sh;
        if (!(src[ip + i] & 0x80)) break;
      }
      return (int64_t)ulen > room ? S3O_E_CAPACITY : S3O_E_BAD_FRAME;
    }
    ip += c;
    op += got;
  }
  return op;
}
