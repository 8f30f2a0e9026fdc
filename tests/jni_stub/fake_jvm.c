/*
 * fake_jvm.c -- a mock of the JNIEnv / JavaVM functions declared in tests/jni_stub/jni.h over plain C objects, so that the JNI shim
 * (jni/tgpu_jni.c) can be EXECUTED without a JVM: tests/test_jni_shim_cpu.py and tests/test_gpu_jni_shim.py build it together with the
 * shim into one shared object and drive the Java_io_trino_operator_gpu_GpuNative_* entry points through ctypes.  Test infrastructure only.
 *
 * What it models: primitive arrays and object arrays (malloc'ed), strings, the GpuNative$NativeError exception (code + message, kept
 * as the "pending exception"), local frames (a depth counter), critical sections (a counter of outstanding pins: the tests assert that
 * every GetPrimitiveArrayCritical was released and that no JNI call happened while an array was pinned -- the rule of the real JVM),
 * and a page-source adapter object whose methods are C function pointers (the upcalls of the scan operator).
 */
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "jni.h"

enum { K_ARRAY = 1, K_OBJECT_ARRAY, K_STRING, K_CLASS, K_EXCEPTION, K_ADAPTER };

struct _jobject {
    int kind;
    jsize n;
    size_t width;
    void *data;        /* arrays: elements; string / class: chars */
    int code;          /* exception */
    char *message;
    /* adapter */
    jint (*next_page)(void);
    jboolean (*is_finished)(void);
    jboolean (*is_blocked)(void);
    jobject (*load_block)(jint channel);
    void (*close_source)(void);
};
struct _jmethodID {
    char name[32];
};

static jobject g_pending = NULL;
static int g_pins = 0, g_frames = 0, g_calls_while_pinned = 0, g_global_refs = 0;
static struct _jmethodID g_methods[16];
static int g_method_count = 0;

static void touch(void)   /* a JNI call other than the critical get / release while an array is pinned is illegal on a real JVM */
{
    if (g_pins > 0) g_calls_while_pinned++;
}
static jobject new_object(int kind)
{
    jobject o = (jobject)calloc(1, sizeof(struct _jobject));
    o->kind = kind;
    return o;
}

/* ---- helpers exported to the tests ---- */
jobject fj_new_array(jsize n, size_t width, const void *data)
{
    jobject o = new_object(K_ARRAY);
    o->n = n;
    o->width = width;
    o->data = calloc((size_t)(n > 0 ? n : 1), width);
    if (data && n > 0) memcpy(o->data, data, (size_t)n * width);
    return o;
}
jobject fj_new_object_array(jsize n)
{
    jobject o = new_object(K_OBJECT_ARRAY);
    o->n = n;
    o->width = sizeof(jobject);
    o->data = calloc((size_t)(n > 0 ? n : 1), sizeof(jobject));
    return o;
}
void fj_set_object(jobject array, jsize i, jobject element) { ((jobject *)array->data)[i] = element; }
void *fj_array_data(jobject array) { return array->data; }
jsize fj_array_length(jobject array) { return array->n; }
const char *fj_string_chars(jobject s) { return s ? (const char *)s->data : NULL; }
void fj_free(jobject o)
{
    if (!o) return;
    free(o->data);
    free(o->message);
    free(o);
}
int fj_pending_code(void) { return g_pending ? g_pending->code : 0; }
const char *fj_pending_message(void) { return g_pending && g_pending->message ? g_pending->message : ""; }
void fj_clear_pending(void)
{
    fj_free(g_pending);
    g_pending = NULL;
}
int fj_outstanding_pins(void) { return g_pins; }
int fj_open_frames(void) { return g_frames; }
int fj_calls_while_pinned(void) { return g_calls_while_pinned; }
int fj_global_refs(void) { return g_global_refs; }
jobject fj_new_adapter(jint (*next_page)(void), jboolean (*is_finished)(void), jboolean (*is_blocked)(void), jobject (*load_block)(jint), void (*close_source)(void))
{
    jobject o = new_object(K_ADAPTER);
    o->next_page = next_page;
    o->is_finished = is_finished;
    o->is_blocked = is_blocked;
    o->load_block = load_block;
    o->close_source = close_source;
    return o;
}

/* ---- the function table ---- */
static jclass FindClass(JNIEnv *env, const char *name)
{
    (void)env;
    touch();
    jobject c = new_object(K_CLASS);
    c->data = strdup(name);
    return c;
}
static jclass GetObjectClass(JNIEnv *env, jobject obj)
{
    (void)obj;
    return FindClass(env, "io/trino/operator/gpu/GpuPageSource");
}
static jmethodID GetMethodID(JNIEnv *env, jclass cls, const char *name, const char *sig)
{
    (void)env; (void)cls; (void)sig;
    touch();
    for (int i = 0; i < g_method_count; i++)
        if (!strcmp(g_methods[i].name, name)) return &g_methods[i];
    if (g_method_count >= 16) return NULL;
    strncpy(g_methods[g_method_count].name, name, sizeof(g_methods[0].name) - 1);
    return &g_methods[g_method_count++];
}
static jobject NewObject(JNIEnv *env, jclass cls, jmethodID ctor, ...)
{
    (void)env; (void)cls; (void)ctor;
    touch();
    va_list ap;
    va_start(ap, ctor);
    const jint code = va_arg(ap, jint);
    jstring msg = va_arg(ap, jstring);
    va_end(ap);
    jobject e = new_object(K_EXCEPTION);
    e->code = code;
    e->message = strdup(msg && msg->data ? (const char *)msg->data : "");
    return e;
}
static jstring NewStringUTF(JNIEnv *env, const char *utf)
{
    (void)env;
    touch();
    jobject s = new_object(K_STRING);
    s->data = strdup(utf ? utf : "");
    return s;
}
static jint Throw(JNIEnv *env, jthrowable obj)
{
    (void)env;
    touch();
    if (g_pending && g_pending != obj) fj_free(g_pending);
    g_pending = obj;
    return 0;
}
static jboolean ExceptionCheck(JNIEnv *env)
{
    (void)env;
    return g_pending != NULL;
}
static jint PushLocalFrame(JNIEnv *env, jint capacity)
{
    (void)env; (void)capacity;
    touch();
    g_frames++;
    return 0;
}
static jobject PopLocalFrame(JNIEnv *env, jobject result)
{
    (void)env;
    touch();
    g_frames--;
    return result;
}
static void DeleteLocalRef(JNIEnv *env, jobject obj) { (void)env; (void)obj; touch(); }
static jobject NewGlobalRef(JNIEnv *env, jobject obj)
{
    (void)env;
    touch();
    g_global_refs++;
    return obj;
}
static void DeleteGlobalRef(JNIEnv *env, jobject obj)
{
    (void)env; (void)obj;
    touch();
    g_global_refs--;
}
static jsize GetArrayLength(JNIEnv *env, jarray array)
{
    (void)env;
    touch();
    return array->n;
}
static jobject GetObjectArrayElement(JNIEnv *env, jobjectArray array, jsize index)
{
    (void)env;
    touch();
    if (!array || index < 0 || index >= array->n) return NULL;
    return ((jobject *)array->data)[index];
}
static void *GetPrimitiveArrayCritical(JNIEnv *env, jarray array, jboolean *isCopy)
{
    (void)env;
    if (isCopy) *isCopy = JNI_FALSE;
    g_pins++;
    return array->data;
}
static void ReleasePrimitiveArrayCritical(JNIEnv *env, jarray array, void *carray, jint mode)
{
    (void)env; (void)array; (void)carray; (void)mode;
    g_pins--;
}
#define ELEMENTS(T, Name)                                                                                                      \
    static T *Get##Name##ArrayElements(JNIEnv *env, jarray array, jboolean *isCopy)                                            \
    {                                                                                                                          \
        (void)env;                                                                                                             \
        touch();                                                                                                               \
        if (isCopy) *isCopy = JNI_FALSE;                                                                                       \
        return (T *)array->data;                                                                                               \
    }                                                                                                                          \
    static void Release##Name##ArrayElements(JNIEnv *env, jarray array, T *elems, jint mode)                                   \
    {                                                                                                                          \
        (void)env; (void)array; (void)elems; (void)mode;                                                                       \
        touch();                                                                                                               \
    }
ELEMENTS(jint, Int)
ELEMENTS(jlong, Long)
ELEMENTS(jdouble, Double)
ELEMENTS(jbyte, Byte)
static void region(jarray array, jsize start, jsize len, void *out, const void *in)
{
    touch();
    if (!array || start < 0 || len < 0 || start + len > array->n) {
        fprintf(stderr, "fake_jvm: array region out of bounds (ArrayIndexOutOfBoundsException)\n");
        abort();
    }
    if (out) memcpy(out, (const char *)array->data + (size_t)start * array->width, (size_t)len * array->width);
    else memcpy((char *)array->data + (size_t)start * array->width, in, (size_t)len * array->width);
}
static void GetIntArrayRegion(JNIEnv *env, jintArray a, jsize s, jsize l, jint *buf) { (void)env; region(a, s, l, buf, NULL); }
static void SetIntArrayRegion(JNIEnv *env, jintArray a, jsize s, jsize l, const jint *buf) { (void)env; region(a, s, l, NULL, buf); }
static void SetLongArrayRegion(JNIEnv *env, jlongArray a, jsize s, jsize l, const jlong *buf) { (void)env; region(a, s, l, NULL, buf); }
static void SetBooleanArrayRegion(JNIEnv *env, jbooleanArray a, jsize s, jsize l, const jboolean *buf) { (void)env; region(a, s, l, NULL, buf); }
static void GetByteArrayRegion(JNIEnv *env, jbyteArray a, jsize s, jsize l, jbyte *buf) { (void)env; region(a, s, l, buf, NULL); }
static void SetByteArrayRegion(JNIEnv *env, jbyteArray a, jsize s, jsize l, const jbyte *buf) { (void)env; region(a, s, l, NULL, buf); }
static jlongArray NewLongArray(JNIEnv *env, jsize len) { (void)env; touch(); return fj_new_array(len, 8, NULL); }
static jbyteArray NewByteArray(JNIEnv *env, jsize len) { (void)env; touch(); return fj_new_array(len, 1, NULL); }

static jint CallIntMethod(JNIEnv *env, jobject obj, jmethodID m, ...)
{
    (void)env;
    touch();
    return !strcmp(m->name, "nextPage") && obj->next_page ? obj->next_page() : -1;
}
static jboolean CallBooleanMethod(JNIEnv *env, jobject obj, jmethodID m, ...)
{
    (void)env;
    touch();
    if (!strcmp(m->name, "isFinished") && obj->is_finished) return obj->is_finished();
    if (!strcmp(m->name, "isBlocked") && obj->is_blocked) return obj->is_blocked();
    return JNI_FALSE;
}
static jobject CallObjectMethod(JNIEnv *env, jobject obj, jmethodID m, ...)
{
    (void)env;
    touch();
    va_list ap;
    va_start(ap, m);
    const jint channel = va_arg(ap, jint);
    va_end(ap);
    return !strcmp(m->name, "loadBlock") && obj->load_block ? obj->load_block(channel) : NULL;
}
static void CallVoidMethod(JNIEnv *env, jobject obj, jmethodID m, ...)
{
    (void)env;
    touch();
    if (!strcmp(m->name, "close") && obj->close_source) obj->close_source();
}

static jint GetEnv(JavaVM *vm, void **penv, jint version);
static const struct JNIInvokeInterface_ g_invoke = {GetEnv};
static JavaVM g_vm = &g_invoke;
static jint GetJavaVM(JNIEnv *env, JavaVM **vm)
{
    (void)env;
    touch();
    *vm = &g_vm;
    return JNI_OK;
}

static const struct JNINativeInterface_ g_functions = {
    FindClass, GetObjectClass, GetMethodID, NewObject, NewStringUTF, Throw, ExceptionCheck, PushLocalFrame, PopLocalFrame, DeleteLocalRef, NewGlobalRef,
    DeleteGlobalRef, GetJavaVM, GetArrayLength, GetObjectArrayElement, GetPrimitiveArrayCritical, ReleasePrimitiveArrayCritical, GetIntArrayElements,
    ReleaseIntArrayElements, GetLongArrayElements, ReleaseLongArrayElements, GetDoubleArrayElements, ReleaseDoubleArrayElements, GetByteArrayElements,
    ReleaseByteArrayElements, GetIntArrayRegion, SetIntArrayRegion, SetLongArrayRegion, SetBooleanArrayRegion, GetByteArrayRegion, SetByteArrayRegion, NewLongArray,
    NewByteArray, CallIntMethod, CallBooleanMethod, CallObjectMethod, CallVoidMethod};
static JNIEnv g_env = &g_functions;

static jint GetEnv(JavaVM *vm, void **penv, jint version)
{
    (void)vm; (void)version;
    *penv = (void *)&g_env;
    return JNI_OK;
}

JNIEnv *fj_env(void) { return &g_env; }
