/* A JNIEnv that runs (TEST INFRASTRUCTURE): the function table of tests/mock_jni/jni.h implemented over plain C
 * structs, so that jni/s3s_jni.c can be EXECUTED without a JVM (tests/mock_jni/jni_exec.c).  It models the parts of the
 * JNI contract the shim can get wrong:
 *   - Get<Type>ArrayElements hands out a COPY; Release with mode 0 copies back, JNI_ABORT discards — an output array
 *     released with JNI_ABORT, or an input array written through, shows up as a wrong result;
 *   - every pin and every local reference is counted: mj_outstanding() must be 0 after each native call;
 *   - GetDirectBufferAddress answers NULL for anything that is not a direct buffer. */
#ifndef MOCK_JVM_H
#define MOCK_JVM_H
#include <jni.h>
#include <stdlib.h>
#include <string.h>

enum { MJ_LONGS = 1, MJ_INTS, MJ_OBJS, MJ_DIRECT, MJ_STRING };
struct _jobject {
  int kind;
  jlong len;  /* elements (arrays), bytes (direct buffers, strings) */
  void* data; /* jlong* / jint* / jobject* / the buffer's address / char* */
};

static int mj_pins, mj_locals;
static int mj_outstanding(void) { return mj_pins + mj_locals; }

static jobject mj_new(int kind, jlong len, void* data) {
  jobject o = (jobject)calloc(1, sizeof *o);
  o->kind = kind;
  o->len = len;
  o->data = data;
  return o;
}
static jlongArray mj_longs(jlong n) { return mj_new(MJ_LONGS, n, calloc((size_t)(n > 0 ? n : 1), sizeof(jlong))); }
static jintArray mj_ints(jlong n) { return mj_new(MJ_INTS, n, calloc((size_t)(n > 0 ? n : 1), sizeof(jint))); }
static jobjectArray mj_objs(jlong n) { return mj_new(MJ_OBJS, n, calloc((size_t)(n > 0 ? n : 1), sizeof(jobject))); }
static jlong* mj_l(jobject a) { return (jlong*)a->data; }
static jint* mj_i(jobject a) { return (jint*)a->data; }
static jobject* mj_o(jobject a) { return (jobject*)a->data; }
static void mj_free(jobject o) {
  if (!o) return;
  if (o->kind != MJ_DIRECT) free(o->data);
  free(o);
}

static jstring mj_NewStringUTF(JNIEnv* e, const char* s) {
  (void)e;
  return mj_new(MJ_STRING, (jlong)strlen(s), memcpy(calloc(strlen(s) + 1, 1), s, strlen(s)));
}
static jlong* mj_GetLongArrayElements(JNIEnv* e, jlongArray a, jboolean* c) {
  (void)e;
  if (!a || a->kind != MJ_LONGS) abort();
  if (c) *c = 1;
  jlong* p = (jlong*)malloc((size_t)(a->len > 0 ? a->len : 1) * sizeof(jlong));
  memcpy(p, a->data, (size_t)a->len * sizeof(jlong));
  mj_pins++;
  return p;
}
static void mj_ReleaseLongArrayElements(JNIEnv* e, jlongArray a, jlong* p, jint mode) {
  (void)e;
  if (!a || a->kind != MJ_LONGS || !p) abort();
  if (mode == 0) memcpy(a->data, p, (size_t)a->len * sizeof(jlong));
  free(p);
  mj_pins--;
}
static jint* mj_GetIntArrayElements(JNIEnv* e, jintArray a, jboolean* c) {
  (void)e;
  if (!a || a->kind != MJ_INTS) abort();
  if (c) *c = 1;
  jint* p = (jint*)malloc((size_t)(a->len > 0 ? a->len : 1) * sizeof(jint));
  memcpy(p, a->data, (size_t)a->len * sizeof(jint));
  mj_pins++;
  return p;
}
static void mj_ReleaseIntArrayElements(JNIEnv* e, jintArray a, jint* p, jint mode) {
  (void)e;
  if (!a || a->kind != MJ_INTS || !p) abort();
  if (mode == 0) memcpy(a->data, p, (size_t)a->len * sizeof(jint));
  free(p);
  mj_pins--;
}
static void* mj_GetDirectBufferAddress(JNIEnv* e, jobject b) {
  (void)e;
  return b && b->kind == MJ_DIRECT ? b->data : NULL;
}
static jlong mj_GetDirectBufferCapacity(JNIEnv* e, jobject b) {
  (void)e;
  return b && b->kind == MJ_DIRECT ? b->len : -1;
}
static jobject mj_NewDirectByteBuffer(JNIEnv* e, void* p, jlong n) {
  (void)e;
  return mj_new(MJ_DIRECT, n, p);
}
static jsize mj_GetArrayLength(JNIEnv* e, jarray a) {
  (void)e;
  if (!a || a->kind < MJ_LONGS || a->kind > MJ_OBJS) abort();
  return (jsize)a->len;
}
static jobject mj_GetObjectArrayElement(JNIEnv* e, jobjectArray a, jsize i) {
  (void)e;
  if (!a || a->kind != MJ_OBJS || i < 0 || i >= a->len) abort();
  jobject o = mj_o(a)[i];
  if (o) mj_locals++; /* a new local reference */
  return o;
}
static void mj_DeleteLocalRef(JNIEnv* e, jobject o) {
  (void)e;
  if (o) mj_locals--;
}
static jint mj_EnsureLocalCapacity(JNIEnv* e, jint n) {
  (void)e;
  return n >= 0 ? 0 : -1;
}

static const struct JNINativeInterface_ mj_table = {
    mj_NewStringUTF,           mj_GetLongArrayElements,    mj_ReleaseLongArrayElements, mj_GetIntArrayElements,
    mj_ReleaseIntArrayElements, mj_GetDirectBufferAddress, mj_GetDirectBufferCapacity,  mj_NewDirectByteBuffer,
    mj_GetArrayLength,         mj_GetObjectArrayElement,   mj_DeleteLocalRef,           mj_EnsureLocalCapacity};
static JNIEnv mj_env = &mj_table;
#endif
