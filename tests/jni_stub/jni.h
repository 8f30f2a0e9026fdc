/*
 * jni.h -- a STUB, written for this repository's tests.  NOT the JDK's header and not derived from it.
 *
 * The build image has no JDK, so jni/tgpu_jni.c could never be seen by a compiler.  This file declares the JNI types and exactly the
 * JNIEnv / JavaVM functions that shim uses, with the signatures the JNI specification gives them, so that every line of the shim is
 * type-checked (-Wall -Wextra -Werror) and linked against libtgpu.so -- and so that tests/jni_stub/fake_jvm.c, a mock of those
 * functions over plain C arrays, can EXECUTE the shim on the GPU box (tests/test_gpu_jni_shim.py).
 *
 * The member ORDER of the function tables below is this file's own: a shim compiled against it is binary-compatible with fake_jvm.c
 * only, never with a real JVM.  Production builds use $JAVA_HOME/include/jni.h (see the build line in jni/tgpu_jni.c).
 */
#ifndef TGPU_TEST_JNI_STUB_H
#define TGPU_TEST_JNI_STUB_H

#include <stdarg.h>
#include <stdint.h>

#define TGPU_JNI_STUB 1

typedef int32_t jint;
typedef int64_t jlong;
typedef int8_t jbyte;
typedef uint8_t jboolean;
typedef uint16_t jchar;
typedef int16_t jshort;
typedef float jfloat;
typedef double jdouble;
typedef jint jsize;

struct _jobject;
typedef struct _jobject *jobject;
typedef jobject jclass;
typedef jobject jthrowable;
typedef jobject jstring;
typedef jobject jarray;
typedef jarray jbooleanArray;
typedef jarray jbyteArray;
typedef jarray jintArray;
typedef jarray jlongArray;
typedef jarray jdoubleArray;
typedef jarray jobjectArray;
struct _jmethodID;
typedef struct _jmethodID *jmethodID;

#define JNI_FALSE 0
#define JNI_TRUE 1
#define JNI_OK 0
#define JNI_ERR (-1)
#define JNI_COMMIT 1
#define JNI_ABORT 2
#define JNI_VERSION_1_8 0x00010008

#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL

struct JNINativeInterface_;
struct JNIInvokeInterface_;
typedef const struct JNINativeInterface_ *JNIEnv;
typedef const struct JNIInvokeInterface_ *JavaVM;

struct JNINativeInterface_ {
    jclass (*FindClass)(JNIEnv *env, const char *name);
    jclass (*GetObjectClass)(JNIEnv *env, jobject obj);
    jmethodID (*GetMethodID)(JNIEnv *env, jclass cls, const char *name, const char *sig);
    jobject (*NewObject)(JNIEnv *env, jclass cls, jmethodID ctor, ...);
    jstring (*NewStringUTF)(JNIEnv *env, const char *utf);
    jint (*Throw)(JNIEnv *env, jthrowable obj);
    jboolean (*ExceptionCheck)(JNIEnv *env);
    jint (*PushLocalFrame)(JNIEnv *env, jint capacity);
    jobject (*PopLocalFrame)(JNIEnv *env, jobject result);
    void (*DeleteLocalRef)(JNIEnv *env, jobject obj);
    jobject (*NewGlobalRef)(JNIEnv *env, jobject obj);
    void (*DeleteGlobalRef)(JNIEnv *env, jobject obj);
    jint (*GetJavaVM)(JNIEnv *env, JavaVM **vm);
    jsize (*GetArrayLength)(JNIEnv *env, jarray array);
    jobject (*GetObjectArrayElement)(JNIEnv *env, jobjectArray array, jsize index);
    void *(*GetPrimitiveArrayCritical)(JNIEnv *env, jarray array, jboolean *isCopy);
    void (*ReleasePrimitiveArrayCritical)(JNIEnv *env, jarray array, void *carray, jint mode);
    jint *(*GetIntArrayElements)(JNIEnv *env, jintArray array, jboolean *isCopy);
    void (*ReleaseIntArrayElements)(JNIEnv *env, jintArray array, jint *elems, jint mode);
    jlong *(*GetLongArrayElements)(JNIEnv *env, jlongArray array, jboolean *isCopy);
    void (*ReleaseLongArrayElements)(JNIEnv *env, jlongArray array, jlong *elems, jint mode);
    jdouble *(*GetDoubleArrayElements)(JNIEnv *env, jdoubleArray array, jboolean *isCopy);
    void (*ReleaseDoubleArrayElements)(JNIEnv *env, jdoubleArray array, jdouble *elems, jint mode);
    jbyte *(*GetByteArrayElements)(JNIEnv *env, jbyteArray array, jboolean *isCopy);
    void (*ReleaseByteArrayElements)(JNIEnv *env, jbyteArray array, jbyte *elems, jint mode);
    void (*GetIntArrayRegion)(JNIEnv *env, jintArray array, jsize start, jsize len, jint *buf);
    void (*SetIntArrayRegion)(JNIEnv *env, jintArray array, jsize start, jsize len, const jint *buf);
    void (*SetLongArrayRegion)(JNIEnv *env, jlongArray array, jsize start, jsize len, const jlong *buf);
    void (*SetBooleanArrayRegion)(JNIEnv *env, jbooleanArray array, jsize start, jsize len, const jboolean *buf);
    void (*GetByteArrayRegion)(JNIEnv *env, jbyteArray array, jsize start, jsize len, jbyte *buf);
    void (*SetByteArrayRegion)(JNIEnv *env, jbyteArray array, jsize start, jsize len, const jbyte *buf);
    jlongArray (*NewLongArray)(JNIEnv *env, jsize len);
    jbyteArray (*NewByteArray)(JNIEnv *env, jsize len);
    jint (*CallIntMethod)(JNIEnv *env, jobject obj, jmethodID method, ...);
    jboolean (*CallBooleanMethod)(JNIEnv *env, jobject obj, jmethodID method, ...);
    jobject (*CallObjectMethod)(JNIEnv *env, jobject obj, jmethodID method, ...);
    void (*CallVoidMethod)(JNIEnv *env, jobject obj, jmethodID method, ...);
};

struct JNIInvokeInterface_ {
    jint (*GetEnv)(JavaVM *vm, void **penv, jint version);
};

#endif
