/* A stand-in for libs3shuffle_codec on a box without a GPU (TEST INFRASTRUCTURE, linked only by tests/test_jni_exec.py's
 * CPU case): the entry points jni/s3s_jni.c calls, with a toy "codec" — a non-empty stream is an 8-byte little-endian
 * length followed by the bytes xor 0x5A, an empty one is 0 bytes — and a toy checksum.  It exists so that the JNI
 * translation unit can be executed against the mock JNIEnv here; the same harness runs against the real library on
 * the GPU box.  Nothing of the product links or loads this file. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "s3shuffle_codec.h"

struct s3s_ctx {
  char err[256];
  int64_t opt[8];
};

static int fail(s3s_ctx* c, int rc, const char* why) {
  if (c) snprintf(c->err, sizeof c->err, "%s", why);
  return rc;
}
static int64_t toy_sum(int algo, const uint8_t* p, int64_t n) {
  uint32_t s = (uint32_t)algo * 1000003u;
  for (int64_t i = 0; i < n; i++) s += (uint32_t)p[i] * (uint32_t)(i + 1);
  return (int64_t)s;
}
static int64_t stream_size(int64_t n) { return n > 0 ? n + 8 : 0; }
static int64_t put_stream(uint8_t* d, const uint8_t* s, int64_t n) {
  if (n <= 0) return 0;
  for (int b = 0; b < 8; b++) d[b] = (uint8_t)((uint64_t)n >> (8 * b));
  for (int64_t i = 0; i < n; i++) d[8 + i] = s[i] ^ 0x5A;
  return n + 8;
}

const char* s3s_version(void) { return "fake"; }
int s3s_abi_version(void) { return S3S_ABI_VERSION; }
int s3s_device_count(void) { return 1; }
s3s_ctx* s3s_create(int dev, int64_t scratch) {
  (void)scratch;
  if (dev != 0) return NULL;
  s3s_ctx* c = (s3s_ctx*)calloc(1, sizeof *c);
  c->opt[S3S_OPT_LZ4_BLOCK_SIZE] = c->opt[S3S_OPT_SNAPPY_BLOCK_SIZE] = 32768;
  return c;
}
void s3s_destroy(s3s_ctx* c) { free(c); }
const char* s3s_last_error(const s3s_ctx* c) { return c ? c->err : "null context"; }
int s3s_set_option(s3s_ctx* c, int key, int64_t v) {
  if (!c || key < 1 || key > 7) return S3S_E_INVALID;
  if ((key == S3S_OPT_LZ4_BLOCK_SIZE || key == S3S_OPT_SNAPPY_BLOCK_SIZE) && (v < 1024 || v > 32768))
    return fail(c, S3S_E_UNSUPPORTED, "block size outside the supported range");
  c->opt[key] = v;
  return S3S_OK;
}
int64_t s3s_get_option(const s3s_ctx* c, int key) { return c && key >= 1 && key <= 7 ? c->opt[key] : -1; }
void* s3s_host_alloc(int64_t n) { return n >= 0 ? malloc((size_t)(n > 0 ? n : 1)) : NULL; }
void s3s_host_free(void* p) { free(p); }

int64_t s3s_max_compressed_size(const s3s_ctx* c, int codec, const int64_t* o, int32_t n) {
  (void)c;
  if (codec < 0 || codec > 2 || n < 0 || (n > 0 && !o)) return -1;
  int64_t t = 0;
  for (int32_t p = 0; p < n; p++) {
    if (o[p + 1] < o[p]) return -1;
    t += stream_size(o[p + 1] - o[p]);
  }
  return t;
}
int64_t s3s_max_compressed_size_segments(const s3s_ctx* c, int codec, const int64_t* so, int32_t ns) {
  return s3s_max_compressed_size(c, codec, so, ns); /* (one stream per segment) */
}
int s3s_compress_map_output_segments(s3s_ctx* c, int codec, int algo, const uint8_t* src, const int64_t* so, int32_t ns,
                                     const int32_t* first, int32_t n, uint8_t* dst, int64_t cap, int64_t* index,
                                     int64_t* sums, int64_t* total) {
  if (!c) return S3S_E_INVALID;
  if (codec < 0 || codec > 2 || n < 0 || ns < 0 || !so || !first || !index || !total || (algo != 0 && !sums))
    return fail(c, S3S_E_INVALID, "bad argument");
  int64_t at = 0;
  index[0] = 0;
  for (int32_t p = 0; p < n; p++) {
    for (int32_t s = first[p]; s < first[p + 1]; s++) {
      const int64_t len = so[s + 1] - so[s];
      if (len < 0) return fail(c, S3S_E_INVALID, "offsets not monotonic");
      if (at + stream_size(len) > cap) return fail(c, S3S_E_CAPACITY, "destination too small");
      at += put_stream(dst + at, src + so[s], len);
    }
    index[p + 1] = at;
    if (algo != 0) sums[p] = toy_sum(algo, dst + index[p], at - index[p]);
  }
  *total = at;
  return S3S_OK;
}
int s3s_compress_map_output(s3s_ctx* c, int codec, int algo, const uint8_t* src, const int64_t* o, int32_t n, uint8_t* dst,
                            int64_t cap, int64_t* index, int64_t* sums, int64_t* total) {
  if (!c) return S3S_E_INVALID;
  if (n < 0) return fail(c, S3S_E_INVALID, "negative partition count");
  int32_t* first = (int32_t*)malloc(((size_t)n + 1) * sizeof *first);
  for (int32_t p = 0; p <= n; p++) first[p] = p;
  const int rc = s3s_compress_map_output_segments(c, codec, algo, src, o, n, first, n, dst, cap, index, sums, total);
  free(first);
  return rc;
}
int s3s_compress_map_outputs_batch(s3s_ctx* c, int codec, int algo, s3s_map_task* t, int32_t n) {
  if (!c || n < 0 || (n > 0 && !t)) return S3S_E_INVALID;
  int worst = S3S_OK;
  for (int32_t i = 0; i < n; i++) {
    if (t[i].num_partitions < 0 || !t[i].src_offsets || !t[i].out_index) return fail(c, S3S_E_INVALID, "bad task");
    t[i].status = s3s_compress_map_output(c, codec, algo, t[i].d_src, t[i].src_offsets, t[i].num_partitions, t[i].d_dst,
                                          t[i].dst_capacity, t[i].out_index, t[i].out_checksums, &t[i].out_total);
    if (t[i].status != S3S_OK && worst == S3S_OK) worst = t[i].status;
  }
  return worst;
}
int s3s_checksum_ranges(s3s_ctx* c, int algo, const uint8_t* d, const int64_t* o, int32_t n, int64_t* out) {
  if (!c || n < 0 || !o || !out) return S3S_E_INVALID;
  for (int32_t p = 0; p < n; p++) out[p] = toy_sum(algo, d + o[p], o[p + 1] - o[p]);
  return S3S_OK;
}
static int walk(const uint8_t* comp, int64_t from, int64_t to, uint8_t* dst, int64_t cap, int64_t* at) {
  while (from < to) {
    if (to - from < 8) return S3S_E_BAD_FRAME;
    int64_t len = 0;
    for (int b = 0; b < 8; b++) len |= (int64_t)comp[from + b] << (8 * b);
    if (len <= 0 || len > to - from - 8) return S3S_E_BAD_FRAME;
    if (dst) {
      if (*at + len > cap) return S3S_E_CAPACITY;
      for (int64_t i = 0; i < len; i++) dst[*at + i] = comp[from + 8 + i] ^ 0x5A;
    }
    *at += len;
    from += 8 + len;
  }
  return S3S_OK;
}
int s3s_decompressed_size(s3s_ctx* c, int codec, const uint8_t* comp, int64_t n, int64_t* out) {
  (void)codec;
  if (!c || n < 0 || !out) return S3S_E_INVALID;
  *out = 0;
  return walk(comp, 0, n, NULL, 0, out);
}
int s3s_decompress_range(s3s_ctx* c, int codec, int algo, const uint8_t* comp, int64_t n, const int64_t* po,
                         const int64_t* refs, int32_t np, uint8_t* dst, int64_t cap, int64_t* out_len, int32_t* bad) {
  (void)codec;
  if (!c || n < 0 || np < 0 || !po || !out_len || (algo != 0 && !refs)) return fail(c, S3S_E_INVALID, "bad argument");
  if (bad) *bad = -1;
  *out_len = 0;
  for (int32_t p = 0; p < np; p++)
    if (algo != 0 && toy_sum(algo, comp + po[p], po[p + 1] - po[p]) != refs[p]) {
      if (bad) *bad = p;
      return fail(c, S3S_E_CHECKSUM, "checksum");
    }
  return walk(comp, 0, n, dst, cap, out_len);
}
int s3s_decompress_ranges_batch(s3s_ctx* c, int codec, int algo, s3s_fetch_range* r, int32_t n) {
  if (!c || n < 0 || (n > 0 && !r)) return S3S_E_INVALID;
  int worst = S3S_OK;
  for (int32_t i = 0; i < n; i++) {
    if (r[i].num_partitions < 0 || !r[i].part_offsets) return fail(c, S3S_E_INVALID, "bad range");
    r[i].status = s3s_decompress_range(c, codec, algo, r[i].d_comp, r[i].comp_len, r[i].part_offsets, r[i].ref_checksums,
                                       r[i].num_partitions, r[i].d_dst, r[i].dst_capacity, &r[i].out_len, &r[i].bad_partition);
    if (r[i].status != S3S_OK && worst == S3S_OK) worst = r[i].status;
  }
  return worst;
}
