/* NOT a JDK header: the few JNI declarations jni/rb_jni.c uses, so that tests/test_capi_symbols.py can ask gcc to
 * type-check the shim in an image without a JDK (gcc -fsyntax-only -Itests/jni_stub).  Types and member signatures follow
 * the JNI specification (jni.h of any JDK); nothing is ever linked or run against this file. */
#ifndef RB_TEST_JNI_STUB_H
#define RB_TEST_JNI_STUB_H
#include <stdint.h>
typedef int32_t jint; typedef int64_t jlong; typedef int8_t jbyte; typedef uint8_t jboolean; typedef float jfloat; typedef jint jsize;
struct _jobject; typedef struct _jobject *jobject;
typedef jobject jclass; typedef jobject jstring; typedef jobject jarray; typedef jarray jlongArray; typedef jarray jbyteArray; typedef jarray jfloatArray; typedef jarray jintArray;
#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#define JNI_ABORT 2
struct JNINativeInterface_;
typedef const struct JNINativeInterface_ *JNIEnv;
struct JNINativeInterface_ {
    jclass (*FindClass)(JNIEnv *, const char *);
    jint (*ThrowNew)(JNIEnv *, jclass, const char *);
    void *(*GetDirectBufferAddress)(JNIEnv *, jobject);
    jobject (*NewDirectByteBuffer)(JNIEnv *, void *, jlong);
    const char *(*GetStringUTFChars)(JNIEnv *, jstring, jboolean *);
    void (*ReleaseStringUTFChars)(JNIEnv *, jstring, const char *);
    jsize (*GetArrayLength)(JNIEnv *, jarray);
    jlongArray (*NewLongArray)(JNIEnv *, jsize);
    void (*SetLongArrayRegion)(JNIEnv *, jlongArray, jsize, jsize, const jlong *);
    jlong *(*GetLongArrayElements)(JNIEnv *, jlongArray, jboolean *);
    jbyte *(*GetByteArrayElements)(JNIEnv *, jbyteArray, jboolean *);
    jfloat *(*GetFloatArrayElements)(JNIEnv *, jfloatArray, jboolean *);
    jint *(*GetIntArrayElements)(JNIEnv *, jintArray, jboolean *);
    void (*ReleaseLongArrayElements)(JNIEnv *, jlongArray, jlong *, jint);
    void (*ReleaseByteArrayElements)(JNIEnv *, jbyteArray, jbyte *, jint);
    void (*ReleaseFloatArrayElements)(JNIEnv *, jfloatArray, jfloat *, jint);
    void (*ReleaseIntArrayElements)(JNIEnv *, jintArray, jint *, jint);
};
#endif
