/*
 * s3s_oracle_lzf.c — LZF (Spark's LZFCompressionCodec) on the CPU: the checker for the GPU LZF decoder.
 *
 * TEST / BENCH INFRASTRUCTURE ONLY (see s3s_oracle.h).
 *
 * [EXT] Spark 3.5.5 LZFCompressionCodec -> com.ning:compress-lzf 1.1.2 LZFOutputStream / LZFInputStream (reached from
 * S3ShuffleReader.scala:59,108 when spark.io.compression.codec=lzf).  Neither the jar nor its source is in this image; the
 * format is Marc Lehmann's liblzf block format inside compress-lzf's chunk framing, restated here from their published
 * descriptions:
 *   chunk     'Z' 'V' 0x00 | len u16 BE | len raw bytes                      (non-compressed chunk)
 *             'Z' 'V' 0x01 | clen u16 BE | ulen u16 BE | clen LZF bytes      (compressed chunk), ulen <= 65535
 *             a stream is chunks until the end of input; concatenated streams are just more chunks
 *   LZF block ctrl < 32: literal run of ctrl + 1 bytes follows
 *             else     len = ctrl >> 5 (7: + next byte), offset = ((ctrl & 0x1f) << 8 | next byte) + 1, copy len + 2 bytes
 * Pinning (tests/test_oracle_pins.py): the block decoder against liblzf 3.6 itself (the C library, reached through the image's
 * conda python3.9 `imagecodecs.lzf_encode / lzf_decode`: both directions, committed fixtures in tests/golden/ for boxes
 * without that interpreter).  The chunk framing is a restatement ("parity unpinned" for the 5 / 7 header bytes).
 * The ENCODER below exists only to make test streams: it is a plain greedy LZF compressor, NOT compress-lzf's (whose output
 * depends on a hash table carried from chunk to chunk and from stream to stream) — the GPU path never
 * compresses LZF.
 */
#include <stdint.h>
#include <string.h>

#include "s3s_oracle.h"

enum { LZF_MAX_CHUNK = 0xFFFF, LZF_MAX_OFF = 1 << 13, LZF_MAX_REF = (1 << 8) + (1 << 3) /* 264 */, LZF_MAX_LIT = 32 };

/* -> decoded length, or S3O_E_BAD_FRAME / S3O_E_CAPACITY */
int s3o_lzf_decompress_block(const uint8_t* src, int slen, uint8_t* dst, int dcap) {
  int ip = 0, op = 0;
  while (ip < slen) {
    const unsigned ctrl = src[ip++];
    if (ctrl < 32) {
      const int n = (int)ctrl + 1;
      if (n > slen - ip) return S3O_E_BAD_FRAME;
      if (n > dcap - op) return S3O_E_CAPACITY;
      memcpy(dst + op, src + ip, (size_t)n);
      ip += n;
      op += n;
    } else {
      int len = (int)(ctrl >> 5);
      if (len == 7) {
        if (ip >= slen) return S3O_E_BAD_FRAME;
        len += src[ip++];
      }
      if (ip >= slen) return S3O_E_BAD_FRAME;
      const int off = (int)(((ctrl & 0x1f) << 8) | src[ip++]) + 1;
      len += 2;
      if (off > op) return S3O_E_BAD_FRAME;
      if (len > dcap - op) return S3O_E_CAPACITY;
      for (int i = 0; i < len; i++) dst[op + i] = dst[op + i - off]; /* (overlap = run) */
      op += len;
    }
  }
  return op;
}

/* greedy test encoder: 3-byte hash, matches of 3..264 bytes up to 8192 back, literal runs of up to 32 */
int s3o_lzf_compress_block(const uint8_t* src, int n, uint8_t* dst, int cap) {
  static _Thread_local int32_t table[1 << 14];
  for (int i = 0; i < (1 << 14); i++) table[i] = -1;
  int ip = 0, op = 0, lit_at = -1, lit = 0;
#define LZF_FLUSH_LIT()                 \
  do {                                  \
    if (lit) {                          \
      dst[lit_at] = (uint8_t)(lit - 1); \
      lit = 0;                          \
    }                                   \
  } while (0)
  while (ip < n) {
    int best = 0, ref = -1;
    if (ip + 3 <= n) {
      const uint32_t v = (uint32_t)src[ip] | ((uint32_t)src[ip + 1] << 8) | ((uint32_t)src[ip + 2] << 16);
      const uint32_t h = (v * 2654435761u) >> 18;
      ref = table[h];
      table[h] = ip;
      if (ref >= 0 && ip - ref <= LZF_MAX_OFF && src[ref] == src[ip] && src[ref + 1] == src[ip + 1] && src[ref + 2] == src[ip + 2]) {
        best = 3;
        while (best < LZF_MAX_REF && ip + best < n && src[ref + best] == src[ip + best]) best++;
      }
    }
    if (best >= 3) {
      LZF_FLUSH_LIT();
      const int off = ip - ref - 1, len = best - 2;
      if (op + 3 > cap) return S3O_E_CAPACITY;
      if (len < 7) {
        dst[op++] = (uint8_t)((len << 5) | (off >> 8));
      } else {
        dst[op++] = (uint8_t)((7 << 5) | (off >> 8));
        dst[op++] = (uint8_t)(len - 7);
      }
      dst[op++] = (uint8_t)off;
      ip += best;
    } else {
      if (lit == 0) {
        if (op + 2 > cap) return S3O_E_CAPACITY;
        lit_at = op++;
      }
      if (op + 1 > cap) return S3O_E_CAPACITY;
      dst[op++] = src[ip++];
      if (++lit == LZF_MAX_LIT) LZF_FLUSH_LIT();
    }
  }
  LZF_FLUSH_LIT();
#undef LZF_FLUSH_LIT
  return op;
}

int64_t s3o_lzf_max_stream_size(int64_t ulen) {
  if (ulen <= 0) return 0;
  const int64_t chunks = (ulen + LZF_MAX_CHUNK - 1) / LZF_MAX_CHUNK;
  return ulen + 7 * chunks;
}

/* LZFOutputStream: chunks of up to 65535 bytes; a chunk that does not shrink by at least 2 bytes is stored */
int64_t s3o_lzf_compress_stream(const uint8_t* src, int64_t ulen, uint8_t* dst, int64_t cap) {
  if (ulen < 0) return S3O_E_INVALID;
  if (cap < s3o_lzf_max_stream_size(ulen)) return S3O_E_CAPACITY;
  static _Thread_local uint8_t tmp[LZF_MAX_CHUNK + LZF_MAX_CHUNK / 16 + 64];
  int64_t op = 0;
  for (int64_t pos = 0; pos < ulen; pos += LZF_MAX_CHUNK) {
    const int n = (int)(ulen - pos < LZF_MAX_CHUNK ? ulen - pos : LZF_MAX_CHUNK);
    const int c = s3o_lzf_compress_block(src + pos, n, tmp, (int)sizeof tmp);
    dst[op++] = 'Z';
    dst[op++] = 'V';
    if (c <= 0 || c >= n - 2) {
      dst[op++] = 0;
      dst[op++] = (uint8_t)(n >> 8);
      dst[op++] = (uint8_t)n;
      memcpy(dst + op, src + pos, (size_t)n);
      op += n;
    } else {
      dst[op++] = 1;
      dst[op++] = (uint8_t)(c >> 8);
      dst[op++] = (uint8_t)c;
      dst[op++] = (uint8_t)(n >> 8);
      dst[op++] = (uint8_t)n;
      memcpy(dst + op, tmp, (size_t)c);
      op += c;
    }
  }
  return op;
}

/* LZFInputStream: chunks until the end of the input */
int64_t s3o_lzf_decompress_stream(const uint8_t* src, int64_t clen, uint8_t* dst, int64_t cap) {
  int64_t ip = 0, op = 0;
  while (ip < clen) {
    if (clen - ip < 5 || src[ip] != 'Z' || src[ip + 1] != 'V' || src[ip + 2] > 1) return S3O_E_BAD_FRAME;
    const int type = src[ip + 2];
    const int len = (src[ip + 3] << 8) | src[ip + 4];
    ip += 5;
    if (type == 0) {
      if (len > clen - ip) return S3O_E_BAD_FRAME;
      if (len > cap - op) return S3O_E_CAPACITY;
      memcpy(dst + op, src + ip, (size_t)len);
      ip += len;
      op += len;
    } else {
      if (clen - ip < 2) return S3O_E_BAD_FRAME;
      const int ulen = (src[ip] << 8) | src[ip + 1];
      ip += 2;
      if (len > clen - ip) return S3O_E_BAD_FRAME;
      if (ulen > cap - op) return S3O_E_CAPACITY;
      const int got = s3o_lzf_decompress_block(src + ip, len, dst + op, ulen);
      if (got != ulen) return S3O_E_BAD_FRAME; /* (a block that wants more room than its header says is corrupt too) */
      ip += len;
      op += ulen;
    }
  }
  return op;
}
