/*
 * s3s_oracle_mt.c — multi-threaded CPU baseline driver (bench.py cpu_baseline leg only).
 *
 * TEST / BENCH INFRASTRUCTURE ONLY (see s3s_oracle.h).
 *
 * Mirrors how Spark runs the reference path: one map task per executor core, each task
 * compressing + checksumming its own map output independently.  Each thread
 * runs s3o_compress_map_output on its own task; when liblz4.so.1 (1.9.3 — the same native
 * code lz4-java reaches through JNI) can be dlopen'ed, `use_liblz4=1` swaps the block
 * compressor for the library's LZ4_compress_default so the baseline is not sandbagged by
 * the restatement's plainer inner loops.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "s3s_oracle.h"

typedef int (*lz4_fn)(const char*, char*, int, int);
static lz4_fn g_lz4 = NULL;

/* libsnappy (1.1.8 in this image; snappy-java binds the same C++ code through JNI, in version 1.1.10): the Snappy legs of
 * the baseline run the LIBRARY's compressor / decompressor, as the LZ4 legs run liblz4 (the
 * restatement's plainer loops made the host look slower than it is).  snappy-c.h API. */
typedef int (*snz_fn)(const char*, size_t, char*, size_t*);
static snz_fn g_sn_compress = NULL, g_sn_uncompress = NULL;
static size_t (*g_sn_bound)(size_t) = NULL;

int s3o_mt_have_libsnappy(void) {
  if (g_sn_compress && g_sn_uncompress && g_sn_bound) return 1;
  static const char* names[] = {"libsnappy.so.1", "/opt/conda/lib/libsnappy.so.1", "/usr/lib/x86_64-linux-gnu/libsnappy.so.1"};
  for (unsigned i = 0; i < sizeof names / sizeof names[0]; i++) {
    void* h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    if (!h) continue;
    g_sn_compress = (snz_fn)dlsym(h, "snappy_compress");
    g_sn_uncompress = (snz_fn)dlsym(h, "snappy_uncompress");
    g_sn_bound = (size_t(*)(size_t))dlsym(h, "snappy_max_compressed_length");
    if (g_sn_compress && g_sn_uncompress && g_sn_bound) return 1;
  }
  return 0;
}

int s3o_mt_have_liblz4(void) {
  if (g_lz4) return 1;
  void* h = dlopen("liblz4.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) return 0;
  g_lz4 = (lz4_fn)dlsym(h, "LZ4_compress_default");
  return g_lz4 != NULL;
}

/* LZ4Block stream using liblz4 for the block compressor (same framing as the oracle). */
static void put_header(uint8_t* h, int token, uint32_t c, uint32_t o, uint32_t check) {
  static const uint8_t magic[8] = {'L', 'Z', '4', 'B', 'l', 'o', 'c', 'k'};
  memcpy(h, magic, 8);
  h[8] = (uint8_t)token;
  for (int i = 0; i < 4; i++) {
    h[9 + i] = (uint8_t)(c >> (8 * i));
    h[13 + i] = (uint8_t)(o >> (8 * i));
    h[17 + i] = (uint8_t)(check >> (8 * i));
  }
}

static int64_t stream_liblz4(const uint8_t* src, int64_t ulen, int bs, uint8_t* dst) {
  int level = 0;
  while ((1 << level) < bs) level++;
  level = level > 10 ? level - 10 : 0;
  int64_t op = 0;
  if (ulen == 0) return 0;
  for (int64_t pos = 0; pos < ulen; pos += bs) {
    const int o = (int)(ulen - pos < bs ? ulen - pos : bs);
    uint8_t* h = dst + op;
    const uint32_t check = s3o_xxh32(src + pos, (size_t)o, 0x9747b28cu) & 0x0FFFFFFFu;
    int r = g_lz4((const char*)src + pos, (char*)h + 21, o, o + o / 255 + 16);
    int method = 0x20;
    if (r >= o || r <= 0) {
      memcpy(h + 21, src + pos, (size_t)o);
      r = o;
      method = 0x10;
    }
    put_header(h, method | level, (uint32_t)r, (uint32_t)o, check);
    op += 21 + r;
  }
  put_header(dst + op, 0x10 | level, 0, 0, 0);
  return op + 21;
}

/* SnappyOutputStream framing (s3s_oracle_snappy.c:s3o_snappy_compress_stream) around libsnappy's block compressor */
static int64_t stream_libsnappy(const uint8_t* src, int64_t ulen, int bs, uint8_t* dst) {
  static const uint8_t hdr[16] = {0x82, 'S', 'N', 'A', 'P', 'P', 'Y', 0, 0, 0, 0, 1, 0, 0, 0, 1};
  if (ulen == 0) return 0;
  if (bs < 1024) bs = 1024;
  memcpy(dst, hdr, 16);
  int64_t op = 16;
  for (int64_t pos = 0; pos < ulen; pos += bs) {
    const size_t o = (size_t)(ulen - pos < bs ? ulen - pos : bs);
    size_t c = g_sn_bound(o);
    if (g_sn_compress((const char*)src + pos, o, (char*)dst + op + 4, &c) != 0) return S3O_E_INVALID;
    dst[op] = (uint8_t)(c >> 24);
    dst[op + 1] = (uint8_t)(c >> 16);
    dst[op + 2] = (uint8_t)(c >> 8);
    dst[op + 3] = (uint8_t)c;
    op += 4 + (int64_t)c;
  }
  return op;
}

/* SnappyInputStream over concatenated streams (s3o_snappy_decompress_stream) with libsnappy's decompressor */
static int64_t decode_libsnappy(const uint8_t* src, int64_t clen, uint8_t* dst, int64_t cap) {
  int64_t ip = 0, op = 0;
  if (clen == 0) return 0;
  if (clen < 16 || memcmp(src, "\x82SNAPPY", 8) != 0) return S3O_E_BAD_FRAME;
  ip = 16;
  while (ip < clen) {
    if (clen - ip < 4) return S3O_E_BAD_FRAME;
    const uint32_t c = ((uint32_t)src[ip] << 24) | ((uint32_t)src[ip + 1] << 16) | ((uint32_t)src[ip + 2] << 8) | src[ip + 3];
    if (c == 0x82534e41u) { /* the header of the next concatenated stream */
      if (clen - ip < 16) return S3O_E_BAD_FRAME;
      ip += 16;
      continue;
    }
    ip += 4;
    if ((int64_t)c > clen - ip) return S3O_E_BAD_FRAME;
    size_t got = (size_t)(cap - op);
    if (g_sn_uncompress((const char*)src + ip, c, (char*)dst + op, &got) != 0) return S3O_E_BAD_FRAME;
    ip += c;
    op += (int64_t)got;
  }
  return op;
}

typedef struct {
  int codec, checksum, block_size, use_liblz4;
  const uint8_t* src;
  const int64_t* offs;
  int32_t nparts;
  uint8_t* dst;
  int64_t cap;
  int64_t* index;
  int64_t* sums;
  int64_t total;
  int rc;
  int reps;
} task_t;

static void* run_task(void* arg) {
  task_t* t = (task_t*)arg;
  for (int r = 0; r < t->reps; r++) {
    const int lib_lz4 = t->use_liblz4 && t->codec == S3O_CODEC_LZ4 && g_lz4;
    const int lib_sn = t->use_liblz4 && t->codec == S3O_CODEC_SNAPPY && g_sn_compress;
    if (lib_lz4 || lib_sn) {
      int64_t op = 0;
      t->index[0] = 0;
      t->rc = 0;
      for (int32_t p = 0; p < t->nparts; p++) {
        int64_t u = t->offs[p + 1] - t->offs[p];
        int64_t w = lib_lz4 ? stream_liblz4(t->src + t->offs[p], u, t->block_size, t->dst + op)
                            : stream_libsnappy(t->src + t->offs[p], u, t->block_size, t->dst + op);
        if (w < 0) {
          t->rc = (int)w;
          break;
        }
        if (t->checksum != S3O_CHECKSUM_NONE) /* java.util.zip runs these as intrinsics: s3s_oracle_simd.c */
          t->sums[p] = s3o_checksum_fast(t->checksum, t->dst + op, (size_t)w);
        op += w;
        t->index[p + 1] = op;
      }
      t->total = op;
    } else {
      t->rc = s3o_compress_map_output(t->codec, t->checksum, t->block_size, t->src, t->offs,
                                      t->nparts, t->dst, t->cap, t->index, t->sums, &t->total);
    }
  }
  return NULL;
}

/* Runs `ntasks` identical-shape map tasks on `nthreads` threads (task i -> thread i %
 * nthreads is NOT used: exactly one task per thread, ntasks == nthreads), `reps` times
 * each, and returns wall seconds.  All tasks read the same src (read-only) but write
 * private outputs.  out_total receives one task's compressed size. */
double s3o_mt_compress_bench(int codec, int checksum, int block_size, int use_liblz4,
                             const uint8_t* src, const int64_t* offs, int32_t nparts,
                             int nthreads, int reps, int64_t* out_total) {
  if (use_liblz4 && codec == S3O_CODEC_LZ4 && !s3o_mt_have_liblz4()) use_liblz4 = 0;
  if (use_liblz4 && codec == S3O_CODEC_SNAPPY && !s3o_mt_have_libsnappy()) use_liblz4 = 0;
  int64_t cap = s3o_max_compressed_size(codec, block_size, offs, nparts);
  task_t* ts = (task_t*)calloc((size_t)nthreads, sizeof(task_t));
  pthread_t* th = (pthread_t*)calloc((size_t)nthreads, sizeof(pthread_t));
  for (int i = 0; i < nthreads; i++) {
    ts[i] = (task_t){codec, checksum, block_size, use_liblz4, src, offs, nparts,
                     (uint8_t*)malloc((size_t)cap + 64), cap,
                     (int64_t*)malloc(sizeof(int64_t) * (size_t)(nparts + 1)),
                     (int64_t*)malloc(sizeof(int64_t) * (size_t)(nparts + 1)), 0, 0, reps};
    memset(ts[i].dst, 0, (size_t)cap); /* fault the pages in before timing */
  }
  struct timespec a, b;
  clock_gettime(CLOCK_MONOTONIC, &a);
  for (int i = 0; i < nthreads; i++) pthread_create(&th[i], NULL, run_task, &ts[i]);
  for (int i = 0; i < nthreads; i++) pthread_join(th[i], NULL);
  clock_gettime(CLOCK_MONOTONIC, &b);
  if (out_total) *out_total = ts[0].total;
  int rc = 0;
  for (int i = 0; i < nthreads; i++) {
    rc |= ts[i].rc;
    free(ts[i].dst);
    free(ts[i].index);
    free(ts[i].sums);
  }
  free(ts);
  free(th);
  double s = (double)(b.tv_sec - a.tv_sec) + 1e-9 * (double)(b.tv_nsec - a.tv_nsec);
  return rc ? -1.0 : s;
}

/* For tests: the libsnappy-backed stream must equal the restatement byte for byte (snappy 1.1.8), and decode back. */
int64_t s3o_mt_stream_libsnappy(const uint8_t* src, int64_t ulen, int block_size, uint8_t* dst) {
  if (!s3o_mt_have_libsnappy()) return S3O_E_UNSUPPORTED;
  return stream_libsnappy(src, ulen, block_size, dst);
}
int64_t s3o_mt_decode_libsnappy(const uint8_t* src, int64_t clen, uint8_t* dst, int64_t cap) {
  if (!s3o_mt_have_libsnappy()) return S3O_E_UNSUPPORTED;
  return decode_libsnappy(src, clen, dst, cap);
}

/* For tests: the liblz4-backed stream must equal the restatement byte for byte. */
int64_t s3o_mt_stream_liblz4(const uint8_t* src, int64_t ulen, int block_size, uint8_t* dst) {
  if (!s3o_mt_have_liblz4()) return S3O_E_UNSUPPORTED;
  return stream_liblz4(src, ulen, block_size, dst);
}

/* ---- reduce side: verify + decompress, one fetched block range per thread ------------------------------- */
typedef int (*lz4d_fn)(const char*, char*, int, int);
static lz4d_fn g_lz4d = NULL;

static int have_liblz4_decoder(void) {
  if (g_lz4d) return 1;
  void* h = dlopen("liblz4.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) return 0;
  g_lz4d = (lz4d_fn)dlsym(h, "LZ4_decompress_safe");
  return g_lz4d != NULL;
}

/* LZ4BlockInputStream over concatenated streams with liblz4's decoder (what lz4-java's JNI instance calls) */
static int64_t decode_liblz4(const uint8_t* c, int64_t n, uint8_t* dst, int64_t cap) {
  int64_t ip = 0, op = 0;
  while (ip < n) {
    if (n - ip < 21 || memcmp(c + ip, "LZ4Block", 8) != 0) return S3O_E_BAD_FRAME;
    const int token = c[ip + 8];
    uint32_t clen = 0, olen = 0, check = 0;
    for (int i = 0; i < 4; i++) {
      clen |= (uint32_t)c[ip + 9 + i] << (8 * i);
      olen |= (uint32_t)c[ip + 13 + i] << (8 * i);
      check |= (uint32_t)c[ip + 17 + i] << (8 * i);
    }
    ip += 21;
    if (olen == 0) continue; /* end frame of one stream: the next stream follows */
    if (ip + clen > n || op + olen > cap) return S3O_E_BAD_FRAME;
    if ((token & 0xF0) == 0x10) {
      memcpy(dst + op, c + ip, olen);
    } else if (g_lz4d((const char*)c + ip, (char*)dst + op, (int)clen, (int)olen) != (int)olen) {
      return S3O_E_BAD_FRAME;
    }
    if ((s3o_xxh32(dst + op, olen, 0x9747b28cu) & 0x0FFFFFFFu) != check) return S3O_E_BAD_FRAME;
    ip += clen;
    op += olen;
  }
  return op;
}

typedef struct {
  int codec, checksum, use_liblz4, reps, rc;
  const uint8_t* comp;
  int64_t comp_len;
  const int64_t* offs;
  const int64_t* sums;
  int32_t nparts;
  uint8_t* dst;
  int64_t cap, out_len;
} dtask_t;

static void* run_dtask(void* arg) {
  dtask_t* t = (dtask_t*)arg;
  for (int r = 0; r < t->reps; r++) {
    if (t->use_liblz4 && (t->codec == S3O_CODEC_LZ4 || t->codec == S3O_CODEC_SNAPPY)) {
      t->rc = 0;
      for (int32_t p = 0; p < t->nparts && t->checksum != S3O_CHECKSUM_NONE; p++) /* S3ChecksumValidationStream */
        if (s3o_checksum_fast(t->checksum, t->comp + t->offs[p], (size_t)(t->offs[p + 1] - t->offs[p])) != t->sums[p])
          t->rc = S3O_E_CHECKSUM;
      if (t->codec == S3O_CODEC_LZ4) {
        t->out_len = decode_liblz4(t->comp, t->comp_len, t->dst, t->cap);
      } else { /* a batch range = the partitions' streams back to back: SnappyInputStream reads across them */
        t->out_len = decode_libsnappy(t->comp, t->comp_len, t->dst, t->cap);
      }
      if (t->out_len < 0) t->rc = (int)t->out_len;
    } else {
      int32_t bad = -1;
      t->rc = s3o_decompress_range(t->codec, t->checksum, t->comp, t->comp_len, t->offs, t->sums, t->nparts, t->dst,
                                   t->cap, &t->out_len, &bad);
    }
  }
  return NULL;
}

/* `nthreads` reduce tasks in parallel, each verifying + decoding the same fetched range `reps` times into a
 * private buffer; returns wall seconds (< 0 on error), *out_len = decoded bytes of one task. */
double s3o_mt_decompress_bench(int codec, int checksum, int use_liblz4, const uint8_t* comp, int64_t comp_len,
                               const int64_t* part_offsets, const int64_t* ref_checksums, int32_t nparts,
                               int64_t dst_capacity, int nthreads, int reps, int64_t* out_len) {
  if (use_liblz4 && codec == S3O_CODEC_LZ4 && !have_liblz4_decoder()) use_liblz4 = 0;
  if (use_liblz4 && codec == S3O_CODEC_SNAPPY && !s3o_mt_have_libsnappy()) use_liblz4 = 0;
  dtask_t* ts = (dtask_t*)calloc((size_t)nthreads, sizeof(dtask_t));
  pthread_t* th = (pthread_t*)calloc((size_t)nthreads, sizeof(pthread_t));
  for (int i = 0; i < nthreads; i++) {
    ts[i] = (dtask_t){codec, checksum, use_liblz4, reps, 0, comp, comp_len, part_offsets, ref_checksums, nparts,
                      (uint8_t*)malloc((size_t)dst_capacity + 64), dst_capacity, 0};
    memset(ts[i].dst, 0, (size_t)dst_capacity);
  }
  struct timespec a, b;
  clock_gettime(CLOCK_MONOTONIC, &a);
  for (int i = 0; i < nthreads; i++) pthread_create(&th[i], NULL, run_dtask, &ts[i]);
  for (int i = 0; i < nthreads; i++) pthread_join(th[i], NULL);
  clock_gettime(CLOCK_MONOTONIC, &b);
  if (out_len) *out_len = ts[0].out_len;
  int rc = 0;
  for (int i = 0; i < nthreads; i++) {
    rc |= ts[i].rc;
    free(ts[i].dst);
  }
  free(ts);
  free(th);
  const double s = (double)(b.tv_sec - a.tv_sec) + 1e-9 * (double)(b.tv_nsec - a.tv_nsec);
  return rc ? -1.0 : s;
}
