/* oracle/ref_shim_jni.c — compiles the reference's JNI shim (JavaGUI/jni/TSDRLibraryNDK.c) FROM ITS SOURCE WHERE IT
 * LIES (the Makefile passes its path as REF_NDK_C; nothing is copied) against oracle/jni_stub/jni.h, and drives its
 * frame callback read_async() — the float -> packed RGB pixel loop of TSDRLibraryNDK.c:222-276 — with a fake JNIEnv
 * whose SetIntArrayRegion hands the pixels back.  This is what pins orc_frame_to_rgb (oracle/tsdr_oracle.c).
 * TEST INFRASTRUCTURE ONLY. */
#include <jni.h>
#include <stdlib.h>
#include <string.h>

#include REF_NDK_C /* the reference's file itself: defines jvm, inverted, java_context_t, read_async, ... */

static jint *g_out;     /* where SetIntArrayRegion delivers */
static jsize g_out_cap;

static jint s_GetVersion(JNIEnv *e) { (void)e; return 0x00010008; }
static jclass s_FindClass(JNIEnv *e, const char *n) { (void)e; (void)n; return (jclass)1; }
static jint s_ThrowNew(JNIEnv *e, jclass c, const char *m) { (void)e; (void)c; (void)m; return 0; }
static jboolean s_ExceptionCheck(JNIEnv *e) { (void)e; return 0; }
static jobject s_NewGlobalRef(JNIEnv *e, jobject o) { (void)e; return o ? o : (jobject)1; }
static void s_DeleteGlobalRef(JNIEnv *e, jobject o) { (void)e; (void)o; }
static void s_DeleteLocalRef(JNIEnv *e, jobject o) { (void)e; (void)o; }
static jobjectRefType s_GetObjectRefType(JNIEnv *e, jobject o) { (void)e; (void)o; return JNIGlobalRefType; }
static jclass s_GetObjectClass(JNIEnv *e, jobject o) { (void)e; (void)o; return (jclass)1; }
static jmethodID s_GetMethodID(JNIEnv *e, jclass c, const char *n, const char *s) { (void)e; (void)c; (void)n; (void)s; return (jmethodID)1; }
static jfieldID s_GetFieldID(JNIEnv *e, jclass c, const char *n, const char *s) { (void)e; (void)c; (void)n; (void)s; return (jfieldID)1; }
static jobject s_GetObjectField(JNIEnv *e, jobject o, jfieldID f) { (void)e; (void)o; (void)f; return (jobject)1; }
static void s_CallVoidMethod(JNIEnv *e, jobject o, jmethodID m, ...) { (void)e; (void)o; (void)m; }
static jint s_CallIntMethod(JNIEnv *e, jobject o, jmethodID m, ...) { (void)e; (void)o; (void)m; return 0; }
static jobject s_CallObjectMethod(JNIEnv *e, jobject o, jmethodID m, ...) { (void)e; (void)o; (void)m; return (jobject)1; }
static const char *s_GetStringUTFChars(JNIEnv *e, jstring s, jboolean *c) { (void)e; (void)s; if (c) *c = 0; return ""; }
static void s_ReleaseStringUTFChars(JNIEnv *e, jstring s, const char *c) { (void)e; (void)s; (void)c; }
static jsize s_GetArrayLength(JNIEnv *e, jarray a) { (void)e; (void)a; return 0x7fffffff; }
static void s_SetIntArrayRegion(JNIEnv *e, jintArray a, jsize start, jsize len, const jint *buf)
{
    (void)e; (void)a;
    if (g_out && start + len <= g_out_cap) memcpy(g_out + start, buf, sizeof(jint) * (size_t)len);
}
static void s_SetDoubleArrayRegion(JNIEnv *e, jdoubleArray a, jsize s, jsize l, const jdouble *b) { (void)e; (void)a; (void)s; (void)l; (void)b; }
static jint s_GetJavaVM(JNIEnv *e, JavaVM **vm);

static const struct JNINativeInterface_ g_env_table = {
    s_GetVersion, s_FindClass, s_ThrowNew, s_ExceptionCheck, s_NewGlobalRef, s_DeleteGlobalRef, s_DeleteLocalRef, s_GetObjectRefType,
    s_GetObjectClass, s_GetMethodID, s_GetFieldID, s_GetObjectField, s_CallVoidMethod, s_CallIntMethod, s_CallObjectMethod,
    s_GetStringUTFChars, s_ReleaseStringUTFChars, s_GetArrayLength, s_SetIntArrayRegion, s_SetDoubleArrayRegion, s_GetJavaVM};
static JNIEnv g_env = &g_env_table;

static jint v_GetEnv(JavaVM *vm, void **penv, jint ver) { (void)vm; (void)ver; *penv = (void *)&g_env; return JNI_OK; }
static jint v_Attach(JavaVM *vm, void **penv, void *args) { (void)vm; (void)args; *penv = (void *)&g_env; return JNI_OK; }
static jint v_Detach(JavaVM *vm) { (void)vm; return JNI_OK; }
static const struct JNIInvokeInterface_ g_vm_table = {v_GetEnv, v_Attach, v_Detach};
static JavaVM g_vm = &g_vm_table;
static jint s_GetJavaVM(JNIEnv *e, JavaVM **vm) { (void)e; *vm = &g_vm; return JNI_OK; }

/* A viewer context that persists between calls, like the GUI's: PIXEL_SPECIAL_VALUE_TRANSPARENT keeps what the
 * previous frame left in the pixel buffer. */
static java_context_t g_ctx;

/* One frame through the reference's read_async(): out receives width*height packed 0x00RRGGBB pixels. */
void ref_jni_frame_to_rgb(const float *frame, int width, int height, int inverted_colours, int32_t *out)
{
    jvm = &g_vm;
    javaversion = 0x00010008;
    inverted = inverted_colours;
    if (!g_ctx.obj) g_ctx.obj = (jobject)1;
    g_out = out;
    g_out_cap = width * height;
    float *copy = (float *)malloc(sizeof(float) * (size_t)width * height); /* read_async takes a non-const buffer */
    memcpy(copy, frame, sizeof(float) * (size_t)width * height);
    read_async(copy, width, height, &g_ctx);
    free(copy);
    g_out = NULL;
}

void ref_jni_reset(void)
{
    free(g_ctx.pixels);
    memset(&g_ctx, 0, sizeof(g_ctx));
}
