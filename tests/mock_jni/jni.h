/* Minimal stand-in for <jni.h> (TEST INFRASTRUCTURE): just enough of the JNI C interface for jni/s3s_jni.c to be
 * type-checked without a JDK (tests/test_jni_shim.py compiles the shim against it with -Wall -Werror, so every
 * wrapper's argument list is checked against include/s3shuffle_codec.h).  Types follow the JNI specification. */
#ifndef MOCK_JNI_H
#define MOCK_JNI_H
#include <stdint.h>
typedef int32_t jint;
typedef int64_t jlong;
typedef int8_t jbyte;
typedef uint8_t jboolean;
typedef struct _jobject* jobject;
typedef jobject jclass, jstring, jarray, jlongArray, jintArray, jbyteArray, jobjectArray;
typedef jint jsize;
#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#define JNI_ABORT 2
#define JNI_FALSE 0
struct JNINativeInterface_;
typedef const struct JNINativeInterface_* JNIEnv;
struct JNINativeInterface_ {
  jstring (*NewStringUTF)(JNIEnv*, const char*);
  jlong* (*GetLongArrayElements)(JNIEnv*, jlongArray, jboolean*);
  void (*ReleaseLongArrayElements)(JNIEnv*, jlongArray, jlong*, jint);
  jint* (*GetIntArrayElements)(JNIEnv*, jintArray, jboolean*);
  void (*ReleaseIntArrayElements)(JNIEnv*, jintArray, jint*, jint);
  void* (*GetDirectBufferAddress)(JNIEnv*, jobject);
  jlong (*GetDirectBufferCapacity)(JNIEnv*, jobject);
  jobject (*NewDirectByteBuffer)(JNIEnv*, void*, jlong);
  jsize (*GetArrayLength)(JNIEnv*, jarray);
  jobject (*GetObjectArrayElement)(JNIEnv*, jobjectArray, jsize);
  void (*DeleteLocalRef)(JNIEnv*, jobject);
  jint (*EnsureLocalCapacity)(JNIEnv*, jint);
};
#endif
