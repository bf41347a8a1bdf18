/*
 * s3s_oracle.c — CPU restatement of the shuffle-block codec path (see s3s_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY: the checker for the HIP path and the cpu_baseline leg of
 * bench.py.  Never linked into, nor called from, spark-s3-shuffle_amd/.
 *
 * Every function names the behaviour it restates.  "[EXT]" = third-party code the
 * reference (IBM/spark-s3-shuffle) reaches through Spark's CompressionCodec /
 * java.util.zip; those sources are not under /root/reference, so the algorithms are
 * restated from their published formats and pinned against the native libraries in
 * this image by tests/test_oracle_pins.py.
 *
 * Little-endian host assumed (x86-64), like the JNI libraries being restated.
 */
#include "s3s_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------ */
/* small helpers                                                                         */
/* ------------------------------------------------------------------------------------ */
static inline uint32_t rd32(const uint8_t* p) {
  uint32_t v;
  memcpy(&v, p, 4);
  return v;
}
static inline void wr32le(uint8_t* p, uint32_t v) {
  p[0] = (uint8_t)v;
  p[1] = (uint8_t)(v >> 8);
  p[2] = (uint8_t)(v >> 16);
  p[3] = (uint8_t)(v >> 24);
}
static inline uint32_t rd32le(const uint8_t* p) {
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
static inline void wr32be(uint8_t* p, uint32_t v) {
  p[0] = (uint8_t)(v >> 24);
  p[1] = (uint8_t)(v >> 16);
  p[2] = (uint8_t)(v >> 8);
  p[3] = (uint8_t)v;
}
static inline uint32_t rd32be(const uint8_t* p) {
  return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3];
}
static inline uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

/* ------------------------------------------------------------------------------------ */
/* [EXT] xxHash32 — the hash lz4-java's StreamingXXHash32 computes                       */
/* ------------------------------------------------------------------------------------ */
#define XXP1 2654435761u
#define XXP2 2246822519u
#define XXP3 3266489917u
#define XXP4 668265263u
#define XXP5 374761393u

uint32_t s3o_xxh32(const void* data, size_t len, uint32_t seed) {
  const uint8_t* p = (const uint8_t*)data;
  const uint8_t* const end = p + len;
  uint32_t h;
  if (len >= 16) {
    const uint8_t* const limit = end - 16;
    uint32_t v1 = seed + XXP1 + XXP2, v2 = seed + XXP2, v3 = seed, v4 = seed - XXP1;
    do {
      v1 = rotl32(v1 + rd32le(p) * XXP2, 13) * XXP1;
      v2 = rotl32(v2 + rd32le(p + 4) * XXP2, 13) * XXP1;
      v3 = rotl32(v3 + rd32le(p + 8) * XXP2, 13) * XXP1;
      v4 = rotl32(v4 + rd32le(p + 12) * XXP2, 13) * XXP1;
      p += 16;
    } while (p <= limit);
    h = rotl32(v1, 1) + rotl32(v2, 7) + rotl32(v3, 12) + rotl32(v4, 18);
  } else {
    h = seed + XXP5;
  }
  h += (uint32_t)len;
  while (p + 4 <= end) {
    h = rotl32(h + rd32le(p) * XXP3, 17) * XXP4;
    p += 4;
  }
  while (p < end) {
    h = rotl32(h + (*p) * XXP5, 11) * XXP1;
    p++;
  }
  h ^= h >> 15;
  h *= XXP2;
  h ^= h >> 13;
  h *= XXP3;
  h ^= h >> 16;
  return h;
}

/* ------------------------------------------------------------------------------------ */
/* [EXT] java.util.zip.CRC32 (IEEE 802.3, reflected 0xEDB88320) — slice-by-8            */
/* reference call sites: S3ShuffleHelper.scala:94-103, S3ChecksumValidationStream:58,72 */
/* ------------------------------------------------------------------------------------ */
static uint32_t crc_tab[8][256];
static int crc_tab_ready = 0;
static void crc_init(void) {
  for (uint32_t i = 0; i < 256; i++) {
    uint32_t c = i;
    for (int k = 0; k < 8; k++) c = (c & 1) ? (0xEDB88320u ^ (c >> 1)) : (c >> 1);
    crc_tab[0][i] = c;
  }
  for (uint32_t i = 0; i < 256; i++)
    for (int t = 1; t < 8; t++)
      crc_tab[t][i] = (crc_tab[t - 1][i] >> 8) ^ crc_tab[0][crc_tab[t - 1][i] & 0xFF];
  crc_tab_ready = 1;
}

uint32_t s3o_crc32(uint32_t crc, const void* data, size_t len) {
  if (!crc_tab_ready) crc_init();
  const uint8_t* p = (const uint8_t*)data;
  uint32_t c = ~crc;
  while (len && ((uintptr_t)p & 7)) {
    c = crc_tab[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
    len--;
  }
  while (len >= 8) {
    uint32_t a = rd32le(p) ^ c, b = rd32le(p + 4);
    c = crc_tab[7][a & 0xFF] ^ crc_tab[6][(a >> 8) & 0xFF] ^ crc_tab[5][(a >> 16) & 0xFF] ^
        crc_tab[4][a >> 24] ^ crc_tab[3][b & 0xFF] ^ crc_tab[2][(b >> 8) & 0xFF] ^
        crc_tab[1][(b >> 16) & 0xFF] ^ crc_tab[0][b >> 24];
    p += 8;
    len -= 8;
  }
  while (len--) c = crc_tab[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
  return ~c;
}

/* ------------------------------------------------------------------------------------ */
/* [EXT] java.util.zip.CRC32C (Castagnoli, reflected 0x82F63B78: RFC 3720 B.4) — the     */
/* third spark.shuffle.checksum.algorithm of Spark 4 (the reference's own                */
/* createChecksumAlgorithm, S3ShuffleHelper.scala:94-103, knows ADLER32 and CRC32).      */
/* Bitwise-table restatement; pinned against the RFC's vectors and the x86 crc32          */
/* instruction (s3o_crc32c_hw, s3s_oracle_simd.c) by tests/test_oracle_pins.py.           */
/* ------------------------------------------------------------------------------------ */
static uint32_t crcc_tab[8][256];
static int crcc_tab_ready = 0;
static void crcc_init(void) {
  for (uint32_t i = 0; i < 256; i++) {
    uint32_t c = i;
    for (int k = 0; k < 8; k++) c = (c & 1) ? (0x82F63B78u ^ (c >> 1)) : (c >> 1);
    crcc_tab[0][i] = c;
  }
  for (uint32_t i = 0; i < 256; i++)
    for (int t = 1; t < 8; t++)
      crcc_tab[t][i] = (crcc_tab[t - 1][i] >> 8) ^ crcc_tab[0][crcc_tab[t - 1][i] & 0xFF];
  crcc_tab_ready = 1;
}
uint32_t s3o_crc32c(uint32_t crc, const void* data, size_t len) {
  if (!crcc_tab_ready) crcc_init();
  const uint8_t* p = (const uint8_t*)data;
  uint32_t c = ~crc;
  while (len >= 8) {
    uint32_t a = rd32le(p) ^ c, b = rd32le(p + 4);
    c = crcc_tab[7][a & 0xFF] ^ crcc_tab[6][(a >> 8) & 0xFF] ^ crcc_tab[5][(a >> 16) & 0xFF] ^
        crcc_tab[4][a >> 24] ^ crcc_tab[3][b & 0xFF] ^ crcc_tab[2][(b >> 8) & 0xFF] ^
        crcc_tab[1][(b >> 16) & 0xFF] ^ crcc_tab[0][b >> 24];
    p += 8;
    len -= 8;
  }
  while (len--) c = crcc_tab[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
  return ~c;
}

/* ------------------------------------------------------------------------------------ */
/* [EXT] java.util.zip.Adler32 (RFC 1950)                                               */
/* ------------------------------------------------------------------------------------ */
uint32_t s3o_adler32(uint32_t adler, const void* data, size_t len) {
  const uint8_t* p = (const uint8_t*)data;
  uint32_t a = adler & 0xFFFF, b = adler >> 16;
  while (len) {
    size_t n = len < 5552 ? len : 5552; /* largest n with 255n(n+1)/2 + (n+1)(65520) < 2^32 */
    len -= n;
    while (n--) {
      a += *p++;
      b += a;
    }
    a %= 65521u;
    b %= 65521u;
  }
  return (b << 16) | a;
}

int64_t s3o_checksum(int algo, const void* data, size_t len) {
  switch (algo) {
    case S3O_CHECKSUM_ADLER32:
      return (int64_t)s3o_adler32(1u, data, len);
    case S3O_CHECKSUM_CRC32:
      return (int64_t)s3o_crc32(0u, data, len);
    case S3O_CHECKSUM_CRC32C:
      return (int64_t)s3o_crc32c(0u, data, len);
    default:
      return 0;
  }
}

/* ------------------------------------------------------------------------------------ */
/* [EXT] liblz4 1.9.3 LZ4_compress_default, restated for the byU16 table mode            */
/*                                                                                      */
/* LZ4_compress_default(src,dst,n,cap) = LZ4_compress_fast(..., acceleration 1) on a     */
/* zero-initialised state; for n < LZ4_64Klimit (65536 + MFLIMIT - 1) the generic        */
/* compressor runs with tableType byU16 (8192 x u16, 13-bit hash of the next 4 bytes),   */
/* no dictionary.  Spark cuts shuffle streams into 32 KiB chunks, so this is the only    */
/* mode the path reaches.                                                                */
/* ------------------------------------------------------------------------------------ */
enum {
  LZ4_MINMATCH = 4,
  LZ4_MFLIMIT = 12,
  LZ4_LASTLITERALS = 5,
  LZ4_MINLENGTH = LZ4_MFLIMIT + 1,
  LZ4_64KLIMIT = 65536 + (LZ4_MFLIMIT - 1),
  LZ4_SKIPTRIGGER = 6,
  LZ4_MLBITS = 4,
  LZ4_MLMASK = 15,
  LZ4_RUNMASK = 15,
  LZ4_MAX_INPUT = 0x7E000000
};

int s3o_lz4_compress_bound(int n) {
  return (unsigned)n > (unsigned)LZ4_MAX_INPUT ? 0 : n + n / 255 + 16;
}

static inline uint32_t lz4_hash_u16(uint32_t seq) { return (seq * 2654435761u) >> (32 - 13); }

/* number of equal bytes at a[..] / b[..] while a-index < limit (LZ4_count) */
static inline int lz4_count(const uint8_t* s, int ia, int ib, int limit) {
  int n = 0;
  while (ia + n + 8 <= limit) { /* 8 bytes per step, like the library's fast path */
    uint64_t x, y;
    memcpy(&x, s + ia + n, 8);
    memcpy(&y, s + ib + n, 8);
    if (x != y) return n + (__builtin_ctzll(x ^ y) >> 3);
    n += 8;
  }
  while (ia + n < limit && s[ia + n] == s[ib + n]) n++;
  return n;
}

int s3o_lz4_compress_block(const uint8_t* src, int n, uint8_t* dst, int cap) {
  if (n < 0 || n >= LZ4_64KLIMIT) return 0;
  const int limited = cap < s3o_lz4_compress_bound(n);
  if (n == 0) {
    if (limited && cap <= 0) return 0;
    dst[0] = 0;
    return 1;
  }
  uint16_t table[1 << 13];
  memset(table, 0, sizeof table); /* LZ4_initStream: an unset slot reads as position 0 */

  int ip = 0, anchor = 0, op = 0;
  const int iend = n;
  const int mflimit_plus_one = iend - LZ4_MFLIMIT + 1;
  const int matchlimit = iend - LZ4_LASTLITERALS;
  uint32_t forward_h;
  int match, token;

  if (n < LZ4_MINLENGTH) goto last_literals; /* too small: all literals */

  /* first byte */
  table[lz4_hash_u16(rd32(src))] = 0;
  ip = 1;
  forward_h = lz4_hash_u16(rd32(src + 1));

  for (;;) {
    /* find a match: greedy first hit, with the skip acceleration of LZ4_compress_fast */
    {
      int forward_ip = ip;
      int step = 1;
      int search_match_nb = 1 << LZ4_SKIPTRIGGER;
      do {
        const uint32_t h = forward_h;
        const int current = forward_ip;
        const int match_index = table[h];
        ip = forward_ip;
        forward_ip += step;
        step = search_match_nb++ >> LZ4_SKIPTRIGGER;
        if (forward_ip > mflimit_plus_one) goto last_literals;
        match = match_index;
        forward_h = lz4_hash_u16(rd32(src + forward_ip));
        table[h] = (uint16_t)current;
        /* byU16: every stored index is within 64 KiB, no distance test */
      } while (rd32(src + match) != rd32(src + ip));
    }

    /* catch up: extend the match backwards over pending literals */
    while (ip > anchor && match > 0 && src[ip - 1] == src[match - 1]) {
      ip--;
      match--;
    }

    /* encode literal run */
    {
      const int lit = ip - anchor;
      token = op++;
      if (limited && op + lit + (2 + 1 + LZ4_LASTLITERALS) + lit / 255 > cap) return 0;
      if (lit >= LZ4_RUNMASK) {
        int len = lit - LZ4_RUNMASK;
        dst[token] = (uint8_t)(LZ4_RUNMASK << LZ4_MLBITS);
        for (; len >= 255; len -= 255) dst[op++] = 255;
        dst[op++] = (uint8_t)len;
      } else {
        dst[token] = (uint8_t)(lit << LZ4_MLBITS);
      }
      memcpy(dst + op, src + anchor, (size_t)lit);
      op += lit;
    }

  next_match:
    /* offset */
    dst[op++] = (uint8_t)(ip - match);
    dst[op++] = (uint8_t)((ip - match) >> 8);

    /* match length */
    {
      int code = lz4_count(src, ip + LZ4_MINMATCH, match + LZ4_MINMATCH, matchlimit);
      ip += code + LZ4_MINMATCH;
      if (limited && op + (1 + LZ4_LASTLITERALS) + (code + 240) / 255 > cap) return 0;
      if (code >= LZ4_MLMASK) {
        dst[token] += LZ4_MLMASK;
        code -= LZ4_MLMASK;
        for (; code >= 255; code -= 255) dst[op++] = 255;
        dst[op++] = (uint8_t)code;
      } else {
        dst[token] += (uint8_t)code;
      }
    }

    anchor = ip;
    if (ip >= mflimit_plus_one) break; /* end of chunk */

    /* fill table with ip-2, then test the position right after the match */
    table[lz4_hash_u16(rd32(src + ip - 2))] = (uint16_t)(ip - 2);
    {
      const uint32_t h = lz4_hash_u16(rd32(src + ip));
      const int match_index = table[h];
      table[h] = (uint16_t)ip;
      if (rd32(src + match_index) == rd32(src + ip)) {
        match = match_index;
        token = op++;
        dst[token] = 0;
        goto next_match;
      }
    }
    forward_h = lz4_hash_u16(rd32(src + ++ip));
  }

last_literals : {
  const int last_run = iend - anchor;
  if (limited && op + last_run + 1 + (last_run + 255 - LZ4_RUNMASK) / 255 > cap) return 0;
  if (last_run >= LZ4_RUNMASK) {
    int acc = last_run - LZ4_RUNMASK;
    dst[op++] = (uint8_t)(LZ4_RUNMASK << LZ4_MLBITS);
    for (; acc >= 255; acc -= 255) dst[op++] = 255;
    dst[op++] = (uint8_t)acc;
  } else {
    dst[op++] = (uint8_t)(last_run << LZ4_MLBITS);
  }
  memcpy(dst + op, src + anchor, (size_t)last_run);
  op += last_run;
}
  return op;
}

/* [EXT] LZ4 block format decoder (safe: both buffers bounds-checked).  Any conforming
 * decoder yields identical bytes, so this follows the published block format rather
 * than a particular liblz4 routine. */
int s3o_lz4_decompress_block(const uint8_t* src, int src_len, uint8_t* dst, int dst_cap,
                             int* consumed) {
  int ip = 0, op = 0;
  if (src_len <= 0) return -1;
  for (;;) {
    if (ip >= src_len) return -1;
    const unsigned token = src[ip++];
    int lit = (int)(token >> 4);
    if (lit == 15) {
      unsigned b;
      do {
        if (ip >= src_len) return -1;
        b = src[ip++];
        lit += (int)b;
        if (lit < 0) return -1;
      } while (b == 255);
    }
    if (lit > src_len - ip || lit > dst_cap - op) return -1;
    memcpy(dst + op, src + ip, (size_t)lit);
    ip += lit;
    op += lit;
    if (ip == src_len) break; /* last sequence: literals only */
    if (src_len - ip < 2) return -1;
    const int offset = (int)src[ip] | ((int)src[ip + 1] << 8);
    ip += 2;
    if (offset == 0 || offset > op) return -1;
    int ml = (int)(token & 15);
    if (ml == 15) {
      unsigned b;
      do {
        if (ip >= src_len) return -1;
        b = src[ip++];
        ml += (int)b;
        if (ml < 0) return -1;
      } while (b == 255);
    }
    ml += LZ4_MINMATCH;
    if (ml > dst_cap - op) return -1;
    for (int i = 0; i < ml; i++) dst[op + i] = dst[op - offset + i]; /* overlap-safe */
    op += ml;
  }
  if (consumed) *consumed = ip;
  return op;
}

/* ------------------------------------------------------------------------------------ */
/* [EXT] lz4-java 1.8.0 LZ4BlockOutputStream / LZ4BlockInputStream                       */
/*                                                                                      */
/* frame = "LZ4Block" | token(method|level) | compressedLen i32 LE | originalLen i32 LE  */
/*         | (xxh32(seed 0x9747b28c) & 0x0FFFFFFF) i32 LE | payload                      */
/* method 0x20 = LZ4, 0x10 = RAW (stored when compressedLen >= originalLen);             */
/* level = max(0, ceil(log2(blockSize)) - 10); finish() appends a 21-byte frame with     */
/* method RAW and three zero ints.  Spark 3.5.5 LZ4CompressionCodec builds it with       */
/* blockSize = spark.io.compression.lz4.blockSize (32 KiB) and syncFlush = false, so     */
/* chunks are cut only at exact multiples of blockSize.                                  */
/* ------------------------------------------------------------------------------------ */
static const uint8_t LZ4B_MAGIC[8] = {'L', 'Z', '4', 'B', 'l', 'o', 'c', 'k'};
enum {
  LZ4B_HEADER = 21,
  LZ4B_METHOD_RAW = 0x10,
  LZ4B_METHOD_LZ4 = 0x20,
  LZ4B_LEVEL_BASE = 10,
  LZ4B_MIN_BLOCK = 64,
  LZ4B_MAX_BLOCK_ORACLE = 65536 /* byU16 restatement only */
};
#define LZ4B_SEED 0x9747b28cu

static int lz4b_level(int block_size) {
  int level = 0;
  while ((1 << level) < block_size) level++; /* ceil(log2) */
  level -= LZ4B_LEVEL_BASE;
  return level < 0 ? 0 : level;
}

int64_t s3o_lz4block_max_stream_size(int64_t ulen, int block_size) {
  if (ulen <= 0) return 0;
  int64_t chunks = (ulen + block_size - 1) / block_size;
  /* RAW fallback caps every stored payload at its chunk length */
  return ulen + chunks * LZ4B_HEADER + LZ4B_HEADER;
}

static void lz4b_header(uint8_t* p, int token, uint32_t clen, uint32_t olen, uint32_t check) {
  memcpy(p, LZ4B_MAGIC, 8);
  p[8] = (uint8_t)token;
  wr32le(p + 9, clen);
  wr32le(p + 13, olen);
  wr32le(p + 17, check);
}

int64_t s3o_lz4block_compress_stream(const uint8_t* src, int64_t ulen, int block_size,
                                     uint8_t* dst, int64_t dst_cap) {
  if (ulen < 0 || block_size < LZ4B_MIN_BLOCK) return S3O_E_INVALID;
  if (block_size > LZ4B_MAX_BLOCK_ORACLE) return S3O_E_UNSUPPORTED;
  if (ulen == 0) return 0; /* partition writer never opened a stream */
  if (dst_cap < s3o_lz4block_max_stream_size(ulen, block_size)) return S3O_E_CAPACITY;
  const int level = lz4b_level(block_size);
  const int bound = s3o_lz4_compress_bound(block_size);
  uint8_t* tmp = (uint8_t*)malloc((size_t)bound);
  if (!tmp) return S3O_E_INVALID;
  int64_t op = 0;
  for (int64_t pos = 0; pos < ulen; pos += block_size) {
    const int o = (int)((ulen - pos) < block_size ? (ulen - pos) : block_size);
    const uint32_t check = s3o_xxh32(src + pos, (size_t)o, LZ4B_SEED) & 0x0FFFFFFFu;
    int clen = s3o_lz4_compress_block(src + pos, o, tmp, bound);
    int method;
    if (clen >= o) { /* flushBufferedData(): not smaller -> store raw */
      method = LZ4B_METHOD_RAW;
      clen = o;
      memcpy(dst + op + LZ4B_HEADER, src + pos, (size_t)o);
    } else {
      method = LZ4B_METHOD_LZ4;
      memcpy(dst + op + LZ4B_HEADER, tmp, (size_t)clen);
    }
    lz4b_header(dst + op, method | level, (uint32_t)clen, (uint32_t)o, check);
    op += LZ4B_HEADER + clen;
  }
  lz4b_header(dst + op, LZ4B_METHOD_RAW | level, 0, 0, 0); /* finish() */
  op += LZ4B_HEADER;
  free(tmp);
  return op;
}

/* LZ4BlockInputStream.refill() with stopOnEmptyBlock=false (what Spark constructs): an
 * end-of-stream frame is skipped and decoding continues with the next concatenated stream;
 * EOF exactly at a frame boundary ends the input, a partial header is an error. */
int64_t s3o_lz4block_decompress_stream(const uint8_t* src, int64_t clen, uint8_t* dst,
                                       int64_t dst_cap) {
  int64_t ip = 0, op = 0;
  while (ip < clen) {
    if (clen - ip < LZ4B_HEADER) return S3O_E_BAD_FRAME; /* "Stream ended prematurely" */
    const uint8_t* h = src + ip;
    if (memcmp(h, LZ4B_MAGIC, 8) != 0) return S3O_E_BAD_FRAME;
    const int token = h[8];
    const int method = token & 0xF0;
    const int level = LZ4B_LEVEL_BASE + (token & 0x0F);
    if (method != LZ4B_METHOD_RAW && method != LZ4B_METHOD_LZ4) return S3O_E_BAD_FRAME;
    const int32_t comp_len = (int32_t)rd32le(h + 9);
    const int32_t orig_len = (int32_t)rd32le(h + 13);
    const uint32_t check = rd32le(h + 17);
    if (orig_len > (1 << level) || orig_len < 0 || comp_len < 0 ||
        (orig_len == 0 && comp_len != 0) || (orig_len != 0 && comp_len == 0) ||
        (method == LZ4B_METHOD_RAW && orig_len != comp_len))
      return S3O_E_BAD_FRAME;
    ip += LZ4B_HEADER;
    if (orig_len == 0 && comp_len == 0) {
      if (check != 0) return S3O_E_BAD_FRAME;
      continue; /* end-of-stream marker: keep going (concatenated streams) */
    }
    if (clen - ip < comp_len) return S3O_E_BAD_FRAME;
    if (dst_cap - op < orig_len) return S3O_E_CAPACITY;
    if (method == LZ4B_METHOD_RAW) {
      memcpy(dst + op, src + ip, (size_t)orig_len);
    } else {
      int used = 0;
      const int got = s3o_lz4_decompress_block(src + ip, comp_len, dst + op, orig_len, &used);
      if (got != orig_len || used != comp_len) return S3O_E_BAD_FRAME;
    }
    if ((s3o_xxh32(dst + op, (size_t)orig_len, LZ4B_SEED) & 0x0FFFFFFFu) != check)
      return S3O_E_BAD_FRAME;
    ip += comp_len;
    op += orig_len;
  }
  return op;
}

/* ------------------------------------------------------------------------------------ */
/* [EXT] raw snappy (restated after snappy 1.1.8's CompressFragment) + snappy-java       */
/* SnappyOutputStream framing.  See s3s_oracle_snappy.c.                                 */
/* ------------------------------------------------------------------------------------ */

/* ------------------------------------------------------------------------------------ */
/* whole map output: .data image + index + per-partition checksums                       */
/* S3ShuffleMapOutputWriter.scala:58,67-83,91-118,197-201; S3ShuffleHelper.scala:44-47   */
/* ------------------------------------------------------------------------------------ */
int64_t s3o_max_compressed_size(int codec, int block_size, const int64_t* src_offsets,
                                int32_t n) {
  int64_t total = 0;
  for (int32_t p = 0; p < n; p++) {
    const int64_t u = src_offsets[p + 1] - src_offsets[p];
    if (u < 0) return S3O_E_INVALID;
    switch (codec) {
      case S3O_CODEC_NONE:
        total += u;
        break;
      case S3O_CODEC_LZ4:
        total += s3o_lz4block_max_stream_size(u, block_size);
        break;
      case S3O_CODEC_SNAPPY:
        total += s3o_snappy_max_stream_size(u, block_size);
        break;
      case S3O_CODEC_LZF:
        total += s3o_lzf_max_stream_size(u);
        break;
      default:
        return S3O_E_INVALID;
    }
  }
  return total;
}

int s3o_compress_map_output(int codec, int checksum_algo, int block_size, const uint8_t* src,
                            const int64_t* src_offsets, int32_t n, uint8_t* dst,
                            int64_t dst_capacity, int64_t* out_index, int64_t* out_checksums,
                            int64_t* out_total) {
  if (n < 0 || !src_offsets || !out_index) return S3O_E_INVALID;
  if (checksum_algo != S3O_CHECKSUM_NONE && !out_checksums) return S3O_E_INVALID;
  int64_t op = 0;
  out_index[0] = 0;
  for (int32_t p = 0; p < n; p++) {
    const int64_t u = src_offsets[p + 1] - src_offsets[p];
    if (u < 0) return S3O_E_INVALID;
    const uint8_t* s = src + src_offsets[p];
    int64_t w;
    switch (codec) {
      case S3O_CODEC_NONE:
        if (dst_capacity - op < u) return S3O_E_CAPACITY;
        memcpy(dst + op, s, (size_t)u);
        w = u;
        break;
      case S3O_CODEC_LZ4:
        w = s3o_lz4block_compress_stream(s, u, block_size, dst + op, dst_capacity - op);
        break;
      case S3O_CODEC_SNAPPY:
        w = s3o_snappy_compress_stream(s, u, block_size, dst + op, dst_capacity - op);
        break;
      case S3O_CODEC_LZF:
        w = s3o_lzf_compress_stream(s, u, dst + op, dst_capacity - op);
        break;
      default:
        return S3O_E_INVALID;
    }
    if (w < 0) return (int)w;
    /* partitionLengths(p) = bytes written for p; the checksum covers exactly those bytes */
    if (checksum_algo != S3O_CHECKSUM_NONE)
      out_checksums[p] = s3o_checksum(checksum_algo, dst + op, (size_t)w);
    op += w;
    out_index[p + 1] = op; /* cumulative, leading 0: writePartitionLengths */
  }
  if (out_total) *out_total = op;
  return S3O_OK;
}

int s3o_decompress_range(int codec, int checksum_algo, const uint8_t* comp, int64_t comp_len,
                         const int64_t* part_offsets, const int64_t* ref_checksums,
                         int32_t nparts, uint8_t* dst, int64_t dst_capacity, int64_t* out_len,
                         int32_t* out_bad_partition) {
  if (out_bad_partition) *out_bad_partition = -1;
  if (nparts < 0 || !part_offsets) return S3O_E_INVALID;
  if (part_offsets[0] != 0 || part_offsets[nparts] != comp_len) return S3O_E_INVALID;
  /* S3ChecksumValidationStream.validateChecksum() (:68-86): a partition is compared when
   * the stream position reaches its length.  A zero-length partition satisfies that at
   * once (constructor call at :39 for a leading one, the recursion at :82-84 for later
   * ones), so it IS compared — against the checksum of no bytes (Adler32 1, CRC32 0). */
  if (checksum_algo != S3O_CHECKSUM_NONE) {
    if (!ref_checksums) return S3O_E_INVALID;
    for (int32_t p = 0; p < nparts; p++) {
      const int64_t len = part_offsets[p + 1] - part_offsets[p];
      if (len < 0) return S3O_E_INVALID;
      const int64_t got = s3o_checksum(checksum_algo, comp + part_offsets[p], (size_t)len);
      if (got != ref_checksums[p]) {
        if (out_bad_partition) *out_bad_partition = p;
        return S3O_E_CHECKSUM;
      }
    }
  }
  int64_t w;
  switch (codec) {
    case S3O_CODEC_NONE:
      if (dst_capacity < comp_len) return S3O_E_CAPACITY;
      memcpy(dst, comp, (size_t)comp_len);
      w = comp_len;
      break;
    case S3O_CODEC_LZ4:
      w = s3o_lz4block_decompress_stream(comp, comp_len, dst, dst_capacity);
      break;
    case S3O_CODEC_SNAPPY:
      w = s3o_snappy_decompress_stream(comp, comp_len, dst, dst_capacity);
      break;
    case S3O_CODEC_LZF:
      w = s3o_lzf_decompress_stream(comp, comp_len, dst, dst_capacity);
      break;
    default:
      return S3O_E_INVALID;
  }
  if (w < 0) return (int)w;
  if (out_len) *out_len = w;
  return S3O_OK;
}

/* S3ShuffleHelper.writeArrayAsBlock / readBlockAsArray: DataOutputStream.writeLong */
void s3o_longs_to_be(const int64_t* v, int64_t n, uint8_t* out) {
  for (int64_t i = 0; i < n; i++) {
    const uint64_t x = (uint64_t)v[i];
    for (int b = 0; b < 8; b++) out[i * 8 + b] = (uint8_t)(x >> (56 - 8 * b));
  }
}

int s3o_longs_from_be(const uint8_t* in, int64_t nbytes, int64_t* v) {
  if (nbytes % 8 != 0) return S3O_E_INVALID; /* "Unexpected file length" */
  for (int64_t i = 0; i < nbytes / 8; i++) {
    uint64_t x = 0;
    for (int b = 0; b < 8; b++) x = (x << 8) | in[i * 8 + b];
    v[i] = (int64_t)x;
  }
  return S3O_OK;
}
