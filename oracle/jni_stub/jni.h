/* oracle/jni_stub/jni.h — NOT the JDK's jni.h: the image has no JDK.  A minimal stand-in written for this
 * repository that declares only the types and the JNIEnv / JavaVM function-table slots the reference's JNI shim
 * (JavaGUI/jni/TSDRLibraryNDK.c) uses, so that the shim can be compiled here — from its source where it lies —
 * and its pixel loop (TSDRLibraryNDK.c:222-276) pinned as the oracle of the frame -> RGB conversion.
 * oracle/ref_shim_jni.c supplies the function tables.  TEST INFRASTRUCTURE ONLY. */
#ifndef ORACLE_JNI_STUB_H_
#define ORACLE_JNI_STUB_H_
#include <stdarg.h>
#include <stdint.h>

typedef int32_t jint;
typedef int64_t jlong;
typedef float jfloat;
typedef double jdouble;
typedef uint8_t jboolean;
typedef jint jsize;
typedef void *jobject;
typedef jobject jclass;
typedef jobject jstring;
typedef jobject jarray;
typedef jarray jintArray;
typedef jarray jdoubleArray;
typedef struct jni_stub_method *jmethodID;
typedef struct jni_stub_field *jfieldID;
typedef enum { JNIInvalidRefType = 0, JNILocalRefType = 1, JNIGlobalRefType = 2, JNIWeakGlobalRefType = 3 } jobjectRefType;

#define JNI_OK 0
#define JNI_EDETACHED (-2)
#define JNI_FALSE 0
#define JNI_TRUE 1
#define JNIEXPORT
#define JNICALL

struct JNINativeInterface_;
struct JNIInvokeInterface_;
typedef const struct JNINativeInterface_ *JNIEnv;
typedef const struct JNIInvokeInterface_ *JavaVM;

struct JNINativeInterface_ {
    jint (*GetVersion)(JNIEnv *);
    jclass (*FindClass)(JNIEnv *, const char *);
    jint (*ThrowNew)(JNIEnv *, jclass, const char *);
    jboolean (*ExceptionCheck)(JNIEnv *);
    jobject (*NewGlobalRef)(JNIEnv *, jobject);
    void (*DeleteGlobalRef)(JNIEnv *, jobject);
    void (*DeleteLocalRef)(JNIEnv *, jobject);
    jobjectRefType (*GetObjectRefType)(JNIEnv *, jobject);
    jclass (*GetObjectClass)(JNIEnv *, jobject);
    jmethodID (*GetMethodID)(JNIEnv *, jclass, const char *, const char *);
    jfieldID (*GetFieldID)(JNIEnv *, jclass, const char *, const char *);
    jobject (*GetObjectField)(JNIEnv *, jobject, jfieldID);
    void (*CallVoidMethod)(JNIEnv *, jobject, jmethodID, ...);
    jint (*CallIntMethod)(JNIEnv *, jobject, jmethodID, ...);
    jobject (*CallObjectMethod)(JNIEnv *, jobject, jmethodID, ...);
    const char *(*GetStringUTFChars)(JNIEnv *, jstring, jboolean *);
    void (*ReleaseStringUTFChars)(JNIEnv *, jstring, const char *);
    jsize (*GetArrayLength)(JNIEnv *, jarray);
    void (*SetIntArrayRegion)(JNIEnv *, jintArray, jsize, jsize, const jint *);
    void (*SetDoubleArrayRegion)(JNIEnv *, jdoubleArray, jsize, jsize, const jdouble *);
    jint (*GetJavaVM)(JNIEnv *, JavaVM **);
};

struct JNIInvokeInterface_ {
    jint (*GetEnv)(JavaVM *, void **, jint);
    jint (*AttachCurrentThread)(JavaVM *, void **, void *);
    jint (*DetachCurrentThread)(JavaVM *);
};
#endif
