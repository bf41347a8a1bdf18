/* Executes jni/s3s_jni.c against the mock JNIEnv of mock_jvm.h (TEST INFRASTRUCTURE; tests/test_jni_exec.py builds and
 * runs it).  Every native of S3SCodec.scala is called the way the Scala shim calls it — direct buffers from hostAlloc,
 * long[] / int[] / Object[] arguments — and its results are compared with the same C-ABI entry point called directly
 * on plain C arrays: the shim has to be transparent.  Then the reduce-side natives take the bytes back to the source.
 * After every native call no array may still be pinned and no local reference may be left.
 *
 * Linked against the real libs3shuffle_codec on the GPU box (codec = LZ4, Adler32) and against fake_codec.c here. */
#include <stdio.h>

#include "mock_jvm.h"
#include "s3shuffle_codec.h"

#define FN(name) Java_org_apache_spark_shuffle_gpu_S3SCodec_00024_##name /* (natives of the Scala object's module class S3SCodec$) */
#define CHECK(x)                                                         \
  do {                                                                   \
    if (!(x)) {                                                          \
      printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #x);              \
      fflush(stdout);                                                    \
      return 1;                                                          \
    }                                                                    \
  } while (0)
#define CLEAN() CHECK(mj_outstanding() == 0)

jint FN(abiVersion)(JNIEnv*, jclass);
jint FN(deviceCount)(JNIEnv*, jclass);
jlong FN(create)(JNIEnv*, jclass, jint, jlong);
void FN(destroy)(JNIEnv*, jclass, jlong);
jint FN(setOption)(JNIEnv*, jclass, jlong, jint, jlong);
jlong FN(getOption)(JNIEnv*, jclass, jlong, jint);
jstring FN(lastError)(JNIEnv*, jclass, jlong);
jobject FN(hostAlloc)(JNIEnv*, jclass, jlong);
void FN(hostFree)(JNIEnv*, jclass, jobject);
jlong FN(maxCompressedSize)(JNIEnv*, jclass, jlong, jint, jlongArray, jint);
jint FN(decompressedSize)(JNIEnv*, jclass, jlong, jint, jobject, jlong, jlongArray);
jint FN(compressMapOutput)(JNIEnv*, jclass, jlong, jint, jint, jobject, jlongArray, jint, jobject, jlong, jlongArray,
                           jlongArray, jlongArray);
jint FN(compressMapOutputSegments)(JNIEnv*, jclass, jlong, jint, jint, jobject, jlongArray, jint, jintArray, jint, jobject,
                                   jlong, jlongArray, jlongArray, jlongArray);
jint FN(checksumRanges)(JNIEnv*, jclass, jlong, jint, jobject, jlongArray, jint, jlongArray);
jint FN(decompressRange)(JNIEnv*, jclass, jlong, jint, jint, jobject, jlong, jlongArray, jlongArray, jint, jobject, jlong,
                         jlongArray, jintArray);
jint FN(compressMapOutputsBatch)(JNIEnv*, jclass, jlong, jint, jint, jobjectArray, jobjectArray, jobjectArray, jlongArray,
                                 jobjectArray, jobjectArray, jlongArray, jintArray);
jint FN(decompressRangesBatch)(JNIEnv*, jclass, jlong, jint, jint, jobjectArray, jlongArray, jobjectArray, jobjectArray,
                               jobjectArray, jlongArray, jlongArray, jintArray, jintArray);

enum { CODEC = S3S_CODEC_LZ4, ALGO = S3S_CHECKSUM_ADLER32 };

/* shuffle-like bytes: records of a 10-byte key and a payload drawn from a small vocabulary (compressible) */
static void fill(uint8_t* p, int64_t n, uint32_t seed) {
  static const char* words[] = {"shuffle ", "partition ", "map_output ", "reduce ", "0000000000", "spark ", "s3a://bucket/"};
  uint32_t s = seed * 2654435761u + 1;
  int64_t i = 0;
  while (i < n) {
    s = s * 1664525u + 1013904223u;
    if ((s >> 28) < 5) {
      for (int k = 0; k < 10 && i < n; k++, s = s * 1664525u + 1013904223u) p[i++] = (uint8_t)(s >> 24);
    } else {
      const char* w = words[(s >> 20) % 7];
      for (int k = 0; w[k] && i < n; k++) p[i++] = (uint8_t)w[k];
    }
  }
}

#define NP 6
static const int64_t kParts[3][NP] = {{40000, 0, 100000, 13, 250000, 70000}, {1, 2, 3, 0, 0, 900000}, {300000, 300000, 0, 5, 64, 1000}};

int main(void) {
  JNIEnv* e = &mj_env;
  CHECK(FN(abiVersion)(e, NULL) == S3S_ABI_VERSION);
  CHECK(FN(deviceCount)(e, NULL) >= 1);
  const jlong h = FN(create)(e, NULL, 0, 0);
  CHECK(h != 0);
  s3s_ctx* ctx = (s3s_ctx*)(intptr_t)h;
  CHECK(FN(setOption)(e, NULL, h, S3S_OPT_LZ4_BLOCK_SIZE, 32768) == S3S_OK);
  CHECK(FN(getOption)(e, NULL, h, S3S_OPT_LZ4_BLOCK_SIZE) == 32768);
  CHECK(FN(setOption)(e, NULL, h, S3S_OPT_LZ4_BLOCK_SIZE, 7) != S3S_OK); /* refused, and the reason is a Java string */
  jstring why = FN(lastError)(e, NULL, h);
  CHECK(why && why->kind == MJ_STRING && why->len > 0);
  mj_free(why);
  CLEAN();

  /* ---- three map tasks: sources in direct buffers from hostAlloc ------------------------------------------------ */
  jobject src[3], dst[3], back[3];
  jlongArray offs[3], index[3], sums[3];
  int64_t usize[3], cap[3], want_total[3];
  uint8_t* want_img[3];
  int64_t want_index[3][NP + 1], want_sums[3][NP];
  for (int t = 0; t < 3; t++) {
    offs[t] = mj_longs(NP + 1);
    for (int p = 0; p < NP; p++) mj_l(offs[t])[p + 1] = mj_l(offs[t])[p] + kParts[t][p];
    usize[t] = mj_l(offs[t])[NP];
    src[t] = FN(hostAlloc)(e, NULL, usize[t]);
    CHECK(src[t] && src[t]->kind == MJ_DIRECT && src[t]->len == usize[t] && src[t]->data);
    fill((uint8_t*)src[t]->data, usize[t], 11u + (uint32_t)t);
    cap[t] = FN(maxCompressedSize)(e, NULL, h, CODEC, offs[t], NP);
    CHECK(cap[t] == s3s_max_compressed_size(ctx, CODEC, (const int64_t*)mj_l(offs[t]), NP) && cap[t] > 0);
    dst[t] = FN(hostAlloc)(e, NULL, cap[t]);
    back[t] = FN(hostAlloc)(e, NULL, usize[t]);
    CHECK(dst[t] && back[t]);
    index[t] = mj_longs(NP + 1);
    sums[t] = mj_longs(NP);
    /* what the C-ABI answers when called directly (plain malloc'd destination) */
    want_img[t] = (uint8_t*)malloc((size_t)cap[t]);
    CHECK(s3s_compress_map_output(ctx, CODEC, ALGO, (const uint8_t*)src[t]->data, (const int64_t*)mj_l(offs[t]), NP, want_img[t],
                                  cap[t], want_index[t], want_sums[t], &want_total[t]) == S3S_OK);
    CHECK(want_total[t] > 0 && want_total[t] <= cap[t] && want_index[t][NP] == want_total[t]);
  }
  CLEAN();

  /* ---- compressMapOutput: same image, index, checksums as the direct call; the input array is not written ------- */
  jlongArray total = mj_longs(1);
  {
    jlong before[NP + 1];
    memcpy(before, mj_l(offs[0]), sizeof before);
    CHECK(FN(compressMapOutput)(e, NULL, h, CODEC, ALGO, src[0], offs[0], NP, dst[0], cap[0], index[0], sums[0], total) == S3S_OK);
    CLEAN();
    CHECK(memcmp(before, mj_l(offs[0]), sizeof before) == 0);
    CHECK(mj_l(total)[0] == want_total[0]);
    CHECK(memcmp(mj_l(index[0]), want_index[0], sizeof want_index[0]) == 0);
    CHECK(memcmp(mj_l(sums[0]), want_sums[0], sizeof want_sums[0]) == 0);
    CHECK(memcmp(dst[0]->data, want_img[0], (size_t)want_total[0]) == 0);
    /* a destination that is too small is the library's S3S_E_CAPACITY, passed through */
    CHECK(FN(compressMapOutput)(e, NULL, h, CODEC, ALGO, src[0], offs[0], NP, dst[0], 100, index[0], sums[0], total) == S3S_E_CAPACITY);
    CLEAN();
    CHECK(FN(compressMapOutput)(e, NULL, h, CODEC, ALGO, src[0], offs[0], NP, dst[0], cap[0], index[0], sums[0], total) == S3S_OK);
  }

  /* ---- checksumRanges over the image = the checksums of the compress call ----------------------------------------- */
  {
    jlongArray out = mj_longs(NP);
    CHECK(FN(checksumRanges)(e, NULL, h, ALGO, dst[0], index[0], NP, out) == S3S_OK);
    CLEAN();
    CHECK(memcmp(mj_l(out), want_sums[0], sizeof want_sums[0]) == 0);
    mj_free(out);
  }

  /* ---- decompressedSize / decompressRange: back to the source; a wrong reference checksum names its partition ----- */
  {
    jlongArray n = mj_longs(1);
    jintArray bad = mj_ints(1);
    CHECK(FN(decompressedSize)(e, NULL, h, CODEC, dst[0], want_total[0], n) == S3S_OK);
    CHECK(mj_l(n)[0] == usize[0]);
    mj_l(n)[0] = 0;
    mj_i(bad)[0] = 77;
    memset(back[0]->data, 0xA5, (size_t)usize[0]);
    CHECK(FN(decompressRange)(e, NULL, h, CODEC, ALGO, dst[0], want_total[0], index[0], sums[0], NP, back[0], usize[0], n, bad) == S3S_OK);
    CLEAN();
    CHECK(mj_l(n)[0] == usize[0] && mj_i(bad)[0] == -1);
    CHECK(memcmp(back[0]->data, src[0]->data, (size_t)usize[0]) == 0);
    mj_l(sums[0])[4] ^= 0x10;
    CHECK(FN(decompressRange)(e, NULL, h, CODEC, ALGO, dst[0], want_total[0], index[0], sums[0], NP, back[0], usize[0], n, bad) == S3S_E_CHECKSUM);
    CLEAN();
    CHECK(mj_i(bad)[0] == 4);
    mj_l(sums[0])[4] ^= 0x10;
    /* without checksums the reference array may be null */
    CHECK(FN(decompressRange)(e, NULL, h, CODEC, S3S_CHECKSUM_NONE, dst[0], want_total[0], index[0], NULL, NP, back[0], usize[0], n, bad) == S3S_OK);
    CLEAN();
    mj_free(n);
    mj_free(bad);
  }

  /* ---- compressMapOutputSegments: every partition in two pieces (two streams); decodes to the same bytes ---------- */
  {
    jlongArray so = mj_longs(2 * NP + 1);
    jintArray first = mj_ints(NP + 1);
    for (int p = 0; p < NP; p++) {
      const jlong a = mj_l(offs[1])[p], b = mj_l(offs[1])[p + 1];
      mj_l(so)[2 * p] = a;
      mj_l(so)[2 * p + 1] = a + (b - a) / 2;
      mj_i(first)[p] = 2 * p;
    }
    mj_l(so)[2 * NP] = usize[1];
    mj_i(first)[NP] = 2 * NP;
    const int64_t scap = s3s_max_compressed_size_segments(ctx, CODEC, (const int64_t*)mj_l(so), 2 * NP);
    CHECK(scap > 0);
    jobject sdst = FN(hostAlloc)(e, NULL, scap);
    jlongArray n = mj_longs(1);
    jintArray bad = mj_ints(1);
    CHECK(FN(compressMapOutputSegments)(e, NULL, h, CODEC, ALGO, src[1], so, 2 * NP, first, NP, sdst, scap, index[1], sums[1], total) == S3S_OK);
    CLEAN();
    CHECK(mj_l(total)[0] == mj_l(index[1])[NP] && mj_l(total)[0] <= scap);
    CHECK(FN(decompressRange)(e, NULL, h, CODEC, ALGO, sdst, mj_l(total)[0], index[1], sums[1], NP, back[1], usize[1], n, bad) == S3S_OK);
    CHECK(mj_l(n)[0] == usize[1] && memcmp(back[1]->data, src[1]->data, (size_t)usize[1]) == 0);
    FN(hostFree)(e, NULL, sdst);
    mj_free(sdst);
    mj_free(so);
    mj_free(first);
    mj_free(n);
    mj_free(bad);
  }

  /* ---- compressMapOutputsBatch: three tasks in one call, each equal to its direct single-task result -------------- */
  jobjectArray a_src = mj_objs(3), a_offs = mj_objs(3), a_dst = mj_objs(3), a_index = mj_objs(3), a_sums = mj_objs(3);
  jlongArray a_cap = mj_longs(3), a_total = mj_longs(3);
  jintArray a_status = mj_ints(3);
  for (int t = 0; t < 3; t++) {
    mj_o(a_src)[t] = src[t];
    mj_o(a_offs)[t] = offs[t];
    mj_o(a_dst)[t] = dst[t];
    mj_o(a_index)[t] = index[t];
    mj_o(a_sums)[t] = sums[t];
    mj_l(a_cap)[t] = cap[t];
    memset(dst[t]->data, 0, (size_t)cap[t]);
    memset(mj_l(index[t]), 0xFF, sizeof(jlong) * (NP + 1));
    mj_i(a_status)[t] = 55;
  }
  CHECK(FN(compressMapOutputsBatch)(e, NULL, h, CODEC, ALGO, a_src, a_offs, a_dst, a_cap, a_index, a_sums, a_total, a_status) == S3S_OK);
  CLEAN();
  for (int t = 0; t < 3; t++) {
    CHECK(mj_i(a_status)[t] == S3S_OK && mj_l(a_total)[t] == want_total[t]);
    CHECK(memcmp(mj_l(index[t]), want_index[t], sizeof want_index[t]) == 0);
    CHECK(memcmp(mj_l(sums[t]), want_sums[t], sizeof want_sums[t]) == 0);
    CHECK(memcmp(dst[t]->data, want_img[t], (size_t)want_total[t]) == 0);
  }
  /* the second task's destination too small: its status says so, the others are complete; the call answers the code */
  mj_l(a_cap)[1] = 100;
  CHECK(FN(compressMapOutputsBatch)(e, NULL, h, CODEC, ALGO, a_src, a_offs, a_dst, a_cap, a_index, a_sums, a_total, a_status) == S3S_E_CAPACITY);
  CLEAN();
  CHECK(mj_i(a_status)[0] == S3S_OK && mj_i(a_status)[1] == S3S_E_CAPACITY && mj_i(a_status)[2] == S3S_OK);
  CHECK(mj_l(a_total)[2] == want_total[2] && memcmp(dst[2]->data, want_img[2], (size_t)want_total[2]) == 0);
  mj_l(a_cap)[1] = cap[1];
  /* arrays of different lengths are refused before anything is pinned */
  {
    jlongArray short_total = mj_longs(2);
    CHECK(FN(compressMapOutputsBatch)(e, NULL, h, CODEC, ALGO, a_src, a_offs, a_dst, a_cap, a_index, a_sums, short_total, a_status) == S3S_E_INVALID);
    CLEAN();
    mj_free(short_total);
  }
  /* an INNER array shorter than the partition count its task's offsets imply (advisor r3: the library would write past the
     pinned JVM array): refused before anything is pinned, nothing written */
  {
    jlongArray short_index = mj_longs(NP), short_sums = mj_longs(NP - 1), keep_i = (jlongArray)mj_o(a_index)[1], keep_s = (jlongArray)mj_o(a_sums)[2];
    mj_o(a_index)[1] = short_index;
    mj_i(a_status)[0] = 77;
    CHECK(FN(compressMapOutputsBatch)(e, NULL, h, CODEC, ALGO, a_src, a_offs, a_dst, a_cap, a_index, a_sums, a_total, a_status) == S3S_E_INVALID);
    CLEAN();
    /* (round 5, advisor r4: a refused call stamps every status with S3S_STATUS_NOT_RUN — a fresh JVM int[] is all zeros = OK,
       and an empty map output would be committed; nothing ELSE is written) */
    CHECK(mj_i(a_status)[0] == S3S_STATUS_NOT_RUN && mj_i(a_status)[1] == S3S_STATUS_NOT_RUN && mj_i(a_status)[2] == S3S_STATUS_NOT_RUN);
    mj_o(a_index)[1] = keep_i;
    mj_o(a_sums)[2] = short_sums;
    CHECK(FN(compressMapOutputsBatch)(e, NULL, h, CODEC, ALGO, a_src, a_offs, a_dst, a_cap, a_index, a_sums, a_total, a_status) == S3S_E_INVALID);
    CLEAN();
    mj_o(a_sums)[2] = keep_s;
    mj_free(short_index);
    mj_free(short_sums);
  }
  CHECK(FN(compressMapOutputsBatch)(e, NULL, h, CODEC, ALGO, a_src, a_offs, a_dst, a_cap, a_index, a_sums, a_total, a_status) == S3S_OK);

  /* ---- decompressRangesBatch: the three images back to their sources; one damaged checksum names range + partition -- */
  {
    jobjectArray a_back = mj_objs(3);
    jlongArray a_clen = mj_longs(3), a_bcap = mj_longs(3), a_olen = mj_longs(3);
    jintArray a_bad = mj_ints(3);
    for (int t = 0; t < 3; t++) {
      mj_o(a_back)[t] = back[t];
      mj_l(a_clen)[t] = want_total[t];
      mj_l(a_bcap)[t] = usize[t];
      memset(back[t]->data, 0xA5, (size_t)usize[t]);
    }
    CHECK(FN(decompressRangesBatch)(e, NULL, h, CODEC, ALGO, a_dst, a_clen, a_index, a_sums, a_back, a_bcap, a_olen, a_bad, a_status) == S3S_OK);
    CLEAN();
    for (int t = 0; t < 3; t++) {
      CHECK(mj_i(a_status)[t] == S3S_OK && mj_i(a_bad)[t] == -1 && mj_l(a_olen)[t] == usize[t]);
      CHECK(memcmp(back[t]->data, src[t]->data, (size_t)usize[t]) == 0);
    }
    mj_l(sums[2])[0] += 1;
    CHECK(FN(decompressRangesBatch)(e, NULL, h, CODEC, ALGO, a_dst, a_clen, a_index, a_sums, a_back, a_bcap, a_olen, a_bad, a_status) == S3S_E_CHECKSUM);
    CLEAN();
    CHECK(mj_i(a_status)[0] == S3S_OK && mj_i(a_status)[1] == S3S_OK && mj_i(a_status)[2] == S3S_E_CHECKSUM && mj_i(a_bad)[2] == 0);
    mj_l(sums[2])[0] -= 1;
    { /* reference checksums shorter than the range's partitions: refused */
      jlongArray short_sums = mj_longs(NP - 1), keep_s = (jlongArray)mj_o(a_sums)[0];
      mj_o(a_sums)[0] = short_sums;
      CHECK(FN(decompressRangesBatch)(e, NULL, h, CODEC, ALGO, a_dst, a_clen, a_index, a_sums, a_back, a_bcap, a_olen, a_bad, a_status) == S3S_E_INVALID);
      CLEAN();
      mj_o(a_sums)[0] = keep_s;
      mj_free(short_sums);
    }
    /* no checksums: the reference array of arrays may be null */
    CHECK(FN(decompressRangesBatch)(e, NULL, h, CODEC, S3S_CHECKSUM_NONE, a_dst, a_clen, a_index, NULL, a_back, a_bcap, a_olen, a_bad, a_status) == S3S_OK);
    CLEAN();
    mj_free(a_back);
    mj_free(a_clen);
    mj_free(a_bcap);
    mj_free(a_olen);
    mj_free(a_bad);
  }

  for (int t = 0; t < 3; t++) {
    FN(hostFree)(e, NULL, src[t]);
    FN(hostFree)(e, NULL, dst[t]);
    FN(hostFree)(e, NULL, back[t]);
    mj_free(src[t]);
    mj_free(dst[t]);
    mj_free(back[t]);
    mj_free(offs[t]);
    mj_free(index[t]);
    mj_free(sums[t]);
    free(want_img[t]);
  }
  mj_free(a_src); mj_free(a_offs); mj_free(a_dst); mj_free(a_index); mj_free(a_sums);
  mj_free(a_cap); mj_free(a_total); mj_free(a_status); mj_free(total);
  FN(destroy)(e, NULL, h);
  CLEAN();
  printf("jni_exec ok\n");
  return 0;
}
