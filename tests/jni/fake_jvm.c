/* tests/jni/fake_jvm.c — a stand-in for the JVM that hosts the reference's JNI shim.  TEST INFRASTRUCTURE ONLY.
 *
 * The Java GUI reaches the library through JavaGUI/jni/TSDRLibraryNDK.c, which the reference builds into libTSDRLibraryNDK.so by
 * linking the static libTSDRLibrary.a (JavaGUI/jni/makefile:122).  oracle/Makefile builds that very shim — from its source where it
 * lies, against oracle/jni_stub/jni.h (the image has no JDK) — on top of OUR libTSDRLibrary.a + libtsdrgpu.so, the way a
 * maintainer who swaps the library would.  This program plays the JVM's part: it dlopens the shim, looks its natives up by their
 * JNI names (Java_martin_tempest_core_TSDRLibrary_*), hands them a JNIEnv whose function table implements what the shim calls
 * (strings, enum ordinals / names, the `pixels` int[] and `double_array` double[] fields, fixSize / notifyCallbacks /
 * onIncomingArray(Notify) / onValueChanged), and replays the call sequence of martin.tempest.core.TSDRLibrary:
 *
 *     init -> loadPlugin -> setBaseFreq -> setGain -> setResolution -> setMotionBlur -> setParam... -> setInvertedColors
 *          -> nativeStart (its own thread: blocks) ... frames arrive through SetIntArrayRegion + notifyCallbacks ...
 *          -> [sync] -> stop -> unloadPlugin -> free
 *
 * Everything the shim delivers is written to a dump file the Python test reads (tests/test_gpu_jni_shim.py).
 *
 * usage: fake_jvm <shim.so> <plugin.so> <plugin params> <height> <refresh> <nframes> <dump> [key=value ...]
 *        keys: blur=<float> inverted=<0|1> param<ordinal>=<int> (PARAM enum ordinal, TSDRLibrary.java) timeout=<s> sync=<pixels>:<DIR>
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <jni.h> /* oracle/jni_stub/jni.h */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

/* ---- the "Java heap" ---------------------------------------------------------------------------------------------------- */
typedef struct { int kind; jsize len; void *data; } jv_array;       /* kind 1: int[], 2: double[] */
typedef struct { int ordinal; const char *name; } jv_enum;
typedef struct { jv_array pixels, doubles; int width, height; } jv_library_object; /* martin.tempest.core.TSDRLibrary */

static jv_library_object g_obj;
static FILE *g_dump;
static pthread_mutex_t g_lock = PTHREAD_MUTEX_INITIALIZER;
static int g_frames, g_frames_wanted, g_plots, g_values, g_exceptions;
static int g_plots_dumped[2];

enum { M_FIXSIZE = 1, M_NOTIFY, M_ONARRAY, M_ONARRAYNOTIFY, M_ONVALUE, M_NAME, M_ORDINAL };
enum { F_PIXELS = 1, F_DOUBLES };

static void put(const void *p, size_t n) { fwrite(p, 1, n, g_dump); }
static void put_i32(int32_t v) { put(&v, 4); }

/* ---- JNIEnv ------------------------------------------------------------------------------------------------------------- */
static jint e_GetVersion(JNIEnv *e) { (void)e; return 0x00010008; }
static jclass e_FindClass(JNIEnv *e, const char *n) { (void)e; return (jclass)strdup(n); } /* a class is its name */
static jint e_ThrowNew(JNIEnv *e, jclass c, const char *m)
{
    (void)e;
    pthread_mutex_lock(&g_lock);
    g_exceptions++;
    const char *cn = c ? (const char *)c : "?";
    const char *msg = m ? m : "";
    put_i32('E'); put_i32((int32_t)strlen(cn)); put(cn, strlen(cn)); put_i32((int32_t)strlen(msg)); put(msg, strlen(msg));
    pthread_mutex_unlock(&g_lock);
    fprintf(stderr, "fake_jvm: exception %s: %s\n", cn, msg);
    return 0;
}
static jboolean e_ExceptionCheck(JNIEnv *e) { (void)e; return 0; }
static jobject e_NewGlobalRef(JNIEnv *e, jobject o) { (void)e; return o; }
static void e_DeleteGlobalRef(JNIEnv *e, jobject o) { (void)e; (void)o; }
static void e_DeleteLocalRef(JNIEnv *e, jobject o) { (void)e; (void)o; }
static jobjectRefType e_GetObjectRefType(JNIEnv *e, jobject o) { (void)e; (void)o; return JNIGlobalRefType; }
static jclass e_GetObjectClass(JNIEnv *e, jobject o) { (void)e; (void)o; return (jclass)"martin/tempest/core/TSDRLibrary"; }
static jmethodID e_GetMethodID(JNIEnv *e, jclass c, const char *n, const char *s)
{
    (void)e; (void)c; (void)s;
    static const struct { const char *n; int id; } tab[] = {{"fixSize", M_FIXSIZE}, {"notifyCallbacks", M_NOTIFY}, {"onIncomingArray", M_ONARRAY},
        {"onIncomingArrayNotify", M_ONARRAYNOTIFY}, {"onValueChanged", M_ONVALUE}, {"name", M_NAME}, {"ordinal", M_ORDINAL}};
    for (size_t i = 0; i < sizeof(tab) / sizeof(tab[0]); i++)
        if (!strcmp(tab[i].n, n)) return (jmethodID)(intptr_t)tab[i].id;
    fprintf(stderr, "fake_jvm: GetMethodID(%s): no such method\n", n);
    abort();
}
static jfieldID e_GetFieldID(JNIEnv *e, jclass c, const char *n, const char *s)
{
    (void)e; (void)c; (void)s;
    if (!strcmp(n, "pixels")) return (jfieldID)(intptr_t)F_PIXELS;
    if (!strcmp(n, "double_array")) return (jfieldID)(intptr_t)F_DOUBLES;
    fprintf(stderr, "fake_jvm: GetFieldID(%s): no such field\n", n);
    abort();
}
static jobject e_GetObjectField(JNIEnv *e, jobject o, jfieldID f)
{
    (void)e;
    jv_library_object *obj = (jv_library_object *)o;
    return (intptr_t)f == F_PIXELS ? (jobject)&obj->pixels : (jobject)&obj->doubles;
}
static void array_fit(jv_array *a, int kind, jsize len)
{
    if (a->len < len || !a->data) {
        a->data = realloc(a->data, (size_t)len * (kind == 1 ? sizeof(jint) : sizeof(jdouble)));
        a->len = len;
    }
    a->kind = kind;
}
static void e_CallVoidMethod(JNIEnv *e, jobject o, jmethodID m, ...)
{
    (void)e;
    jv_library_object *obj = (jv_library_object *)o;
    va_list ap;
    va_start(ap, m);
    switch ((int)(intptr_t)m) {
    case M_FIXSIZE: { /* TSDRLibrary.fixSize(int, int): (re)allocates the BufferedImage's int[] */
        const int w = va_arg(ap, int), h = va_arg(ap, int);
        pthread_mutex_lock(&g_lock);
        obj->width = w; obj->height = h;
        free(obj->pixels.data); obj->pixels.data = NULL; obj->pixels.len = 0; /* a NEW array, like `new BufferedImage` */
        array_fit(&obj->pixels, 1, (jsize)w * h);
        memset(obj->pixels.data, 0, sizeof(jint) * (size_t)w * h);
        pthread_mutex_unlock(&g_lock);
        break;
    }
    case M_NOTIFY: /* TSDRLibrary.notifyCallbacks(): the frame in `pixels` goes to the listeners */
        pthread_mutex_lock(&g_lock);
        if (g_frames < g_frames_wanted) {
            put_i32('F'); put_i32(obj->width); put_i32(obj->height);
            put(obj->pixels.data, sizeof(jint) * (size_t)obj->width * obj->height);
        }
        g_frames++;
        pthread_mutex_unlock(&g_lock);
        break;
    case M_ONARRAY: { /* onIncomingArray(int size): makes double_array large enough */
        const int size = va_arg(ap, int);
        pthread_mutex_lock(&g_lock);
        array_fit(&obj->doubles, 2, size);
        pthread_mutex_unlock(&g_lock);
        break;
    }
    case M_ONARRAYNOTIFY: { /* onIncomingArrayNotify(int plot_id, int offset, int size, long samplerate) */
        const int id = va_arg(ap, int), off = va_arg(ap, int), size = va_arg(ap, int);
        const jlong rate = va_arg(ap, jlong);
        pthread_mutex_lock(&g_lock);
        g_plots++;
        if (id >= 0 && id < 2 && g_plots_dumped[id] < 2) { /* the first two updates of each plot, whole */
            g_plots_dumped[id]++;
            put_i32('P'); put_i32(id); put_i32(off); put_i32(size); put(&rate, 8);
            put(obj->doubles.data, sizeof(jdouble) * (size_t)size);
        }
        pthread_mutex_unlock(&g_lock);
        break;
    }
    case M_ONVALUE: { /* onValueChanged(int, double, double) */
        const int id = va_arg(ap, int);
        const double a0 = va_arg(ap, double), a1 = va_arg(ap, double);
        pthread_mutex_lock(&g_lock);
        g_values++;
        put_i32('V'); put_i32(id); put(&a0, 8); put(&a1, 8);
        pthread_mutex_unlock(&g_lock);
        break;
    }
    default:
        fprintf(stderr, "fake_jvm: CallVoidMethod on method %d\n", (int)(intptr_t)m);
        abort();
    }
    va_end(ap);
}
static jint e_CallIntMethod(JNIEnv *e, jobject o, jmethodID m, ...)
{
    (void)e;
    if ((int)(intptr_t)m != M_ORDINAL) abort();
    return ((jv_enum *)o)->ordinal;
}
static jobject e_CallObjectMethod(JNIEnv *e, jobject o, jmethodID m, ...)
{
    (void)e;
    if ((int)(intptr_t)m != M_NAME) abort();
    return (jobject)((jv_enum *)o)->name; /* a String is its chars */
}
static const char *e_GetStringUTFChars(JNIEnv *e, jstring s, jboolean *c) { (void)e; if (c) *c = 0; return (const char *)s; }
static void e_ReleaseStringUTFChars(JNIEnv *e, jstring s, const char *c) { (void)e; (void)s; (void)c; }
static jsize e_GetArrayLength(JNIEnv *e, jarray a) { (void)e; return ((jv_array *)a)->len; }
static void e_SetIntArrayRegion(JNIEnv *e, jintArray a, jsize start, jsize len, const jint *buf)
{
    (void)e;
    jv_array *arr = (jv_array *)a;
    if (arr->kind != 1 || start < 0 || start + len > arr->len) { fprintf(stderr, "fake_jvm: ArrayIndexOutOfBoundsException (int[])\n"); abort(); }
    memcpy((jint *)arr->data + start, buf, sizeof(jint) * (size_t)len);
}
static void e_SetDoubleArrayRegion(JNIEnv *e, jdoubleArray a, jsize start, jsize len, const jdouble *buf)
{
    (void)e;
    jv_array *arr = (jv_array *)a;
    if (arr->kind != 2 || start < 0 || start + len > arr->len) { fprintf(stderr, "fake_jvm: ArrayIndexOutOfBoundsException (double[])\n"); abort(); }
    memcpy((jdouble *)arr->data + start, buf, sizeof(jdouble) * (size_t)len);
}
static jint e_GetJavaVM(JNIEnv *e, JavaVM **vm);

static const struct JNINativeInterface_ g_env_table = {
    e_GetVersion, e_FindClass, e_ThrowNew, e_ExceptionCheck, e_NewGlobalRef, e_DeleteGlobalRef, e_DeleteLocalRef, e_GetObjectRefType,
    e_GetObjectClass, e_GetMethodID, e_GetFieldID, e_GetObjectField, e_CallVoidMethod, e_CallIntMethod, e_CallObjectMethod,
    e_GetStringUTFChars, e_ReleaseStringUTFChars, e_GetArrayLength, e_SetIntArrayRegion, e_SetDoubleArrayRegion, e_GetJavaVM};
static JNIEnv g_env = &g_env_table;

static jint v_GetEnv(JavaVM *vm, void **penv, jint ver) { (void)vm; (void)ver; *penv = (void *)&g_env; return JNI_OK; }
static jint v_Attach(JavaVM *vm, void **penv, void *args) { (void)vm; (void)args; *penv = (void *)&g_env; return JNI_OK; }
static jint v_Detach(JavaVM *vm) { (void)vm; return JNI_OK; }
static const struct JNIInvokeInterface_ g_vm_table = {v_GetEnv, v_Attach, v_Detach};
static JavaVM g_vm = &g_vm_table;
static jint e_GetJavaVM(JNIEnv *e, JavaVM **vm) { (void)e; *vm = &g_vm; return JNI_OK; }

/* ---- the natives, by their JNI names -------------------------------------------------------------------------------------- */
#define NATIVE(ret, name, ...) typedef ret (*fn_##name)(JNIEnv *, jobject, ##__VA_ARGS__); static fn_##name n_##name
NATIVE(void, init);
NATIVE(void, loadPlugin, jstring, jstring);
NATIVE(void, setBaseFreq, jlong);
NATIVE(void, setGain, jfloat);
NATIVE(void, setResolution, jint, jdouble);
NATIVE(void, setMotionBlur, jfloat);
NATIVE(void, setParam, jobject, jlong);
NATIVE(void, setInvertedColors, jboolean);
NATIVE(void, nativeStart);
NATIVE(jboolean, isRunning);
NATIVE(void, sync, jint, jobject);
NATIVE(void, stop);
NATIVE(void, unloadPlugin);
NATIVE(void, free);

static void *sym(void *dl, const char *name)
{
    char full[160];
    snprintf(full, sizeof(full), "Java_martin_tempest_core_TSDRLibrary_%s", name);
    void *p = dlsym(dl, full);
    if (!p) { fprintf(stderr, "fake_jvm: UnsatisfiedLinkError: %s\n", full); exit(3); }
    return p;
}

static void *start_thread(void *arg)
{
    (void)arg;
    n_nativeStart(&g_env, (jobject)&g_obj); /* blocks until stop() */
    return NULL;
}

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + ts.tv_nsec * 1e-9;
}

int main(int argc, char **argv)
{
    if (argc < 8) {
        fprintf(stderr, "usage: %s <shim.so> <plugin.so> <params> <height> <refresh> <nframes> <dump> [blur=.. inverted=.. paramN=.. timeout=.. sync=px:DIR]\n", argv[0]);
        return 2;
    }
    void *dl = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL); /* System.loadLibrary("TSDRLibraryNDK") */
    if (!dl) { fprintf(stderr, "fake_jvm: %s\n", dlerror()); return 3; }
    char *plugin = argv[2], *params = argv[3];
    const int height = atoi(argv[4]);
    const double refresh = atof(argv[5]);
    g_frames_wanted = atoi(argv[6]);
    g_dump = fopen(argv[7], "wb");
    if (!g_dump) { perror(argv[7]); return 3; }
    float blur = 0.f;
    int inverted = 0, sync_px = 0;
    double timeout = 60.0;
    const char *sync_dir = NULL;
    jv_enum prm[16];
    long long prm_val[16];
    int nprm = 0;
    for (int i = 8; i < argc; i++) {
        if (!strncmp(argv[i], "blur=", 5)) blur = (float)atof(argv[i] + 5);
        else if (!strncmp(argv[i], "inverted=", 9)) inverted = atoi(argv[i] + 9);
        else if (!strncmp(argv[i], "timeout=", 8)) timeout = atof(argv[i] + 8);
        else if (!strncmp(argv[i], "sync=", 5)) { sync_px = atoi(argv[i] + 5); sync_dir = strchr(argv[i], ':') ? strchr(argv[i], ':') + 1 : "ANY"; }
        else if (!strncmp(argv[i], "param", 5) && nprm < 16) {
            prm[nprm].ordinal = atoi(argv[i] + 5);
            prm[nprm].name = "PARAM";
            prm_val[nprm] = strchr(argv[i], '=') ? atoll(strchr(argv[i], '=') + 1) : 0;
            nprm++;
        } else { fprintf(stderr, "fake_jvm: unknown argument %s\n", argv[i]); return 2; }
    }
#define BIND(name) n_##name = (fn_##name)sym(dl, #name)
    BIND(init); BIND(loadPlugin); BIND(setBaseFreq); BIND(setGain); BIND(setResolution); BIND(setMotionBlur); BIND(setParam);
    BIND(setInvertedColors); BIND(nativeStart); BIND(isRunning); BIND(sync); BIND(stop); BIND(unloadPlugin); BIND(free);

    jobject self = (jobject)&g_obj;
    n_init(&g_env, self);
    n_loadPlugin(&g_env, self, (jstring)plugin, (jstring)params);
    if (g_exceptions) { fclose(g_dump); return 4; }
    n_setBaseFreq(&g_env, self, 400000000LL);
    n_setGain(&g_env, self, 0.5f);
    n_setResolution(&g_env, self, height, refresh);
    n_setMotionBlur(&g_env, self, blur);
    for (int i = 0; i < nprm; i++) n_setParam(&g_env, self, (jobject)&prm[i], (jlong)prm_val[i]);
    n_setInvertedColors(&g_env, self, inverted ? JNI_TRUE : JNI_FALSE);

    pthread_t th;
    pthread_create(&th, NULL, start_thread, NULL);
    const double t0 = now_s();
    int running_seen = 0, synced = 0;
    for (;;) {
        pthread_mutex_lock(&g_lock);
        const int have = g_frames, exc = g_exceptions;
        pthread_mutex_unlock(&g_lock);
        if (n_isRunning(&g_env, self)) running_seen = 1;
        if (sync_dir && !synced && have >= g_frames_wanted / 2) {
            jv_enum dir = {0, sync_dir};
            n_sync(&g_env, self, sync_px, (jobject)&dir);
            synced = 1;
        }
        if (have >= g_frames_wanted || exc || now_s() - t0 > timeout) break;
        if (running_seen && !n_isRunning(&g_env, self)) break; /* the session ended on its own */
        usleep(2000);
    }
    const double t1 = now_s();
    n_stop(&g_env, self);
    pthread_join(th, NULL);
    const int still = n_isRunning(&g_env, self);
    n_unloadPlugin(&g_env, self);
    n_free(&g_env, self);
    pthread_mutex_lock(&g_lock);
    put_i32('Z'); put_i32(g_frames); put_i32(g_plots); put_i32(g_values); put_i32(g_exceptions);
    pthread_mutex_unlock(&g_lock);
    fclose(g_dump);
    printf("fake_jvm: frames %d plots %d values %d exceptions %d running_seen %d running_after_stop %d seconds %.3f\n", g_frames, g_plots, g_values,
           g_exceptions, running_seen, still, t1 - t0);
    return g_exceptions ? 4 : (g_frames >= g_frames_wanted ? 0 : 5);
}
