/* tsdr_host.h — internals of libTSDRLibrary (the tsdr_* drop-in, host side in C).
 * The reference keeps the equivalent state in struct tsdr_lib
 * (TempestSDR/src/internaldefinitions.h:30-65). */
#ifndef TSDR_HOST_H_
#define TSDR_HOST_H_

#include <pthread.h>
#include <stdint.h>

#include "TSDRCodes.h"
/* the library is built with -fvisibility=hidden: only the tsdr_* API is exported */
#pragma GCC visibility push(default)
#include "TSDRLibrary.h"
#include "TSDRLibraryExt.h"
#pragma GCC visibility pop
#include "tsdrgpu.h"

/* plugin entry points are reached through dlsym only */
typedef void (*tsdrplugin_readasync_function)(float *buf, uint64_t items_count, void *ctx, int64_t samples_dropped);

/* the ten tsdrplugin_* entry points of a loaded source plugin */
typedef struct plugin_host {
    void *dl;
    int loaded;
    void (*getName)(char *);
    int (*init)(const char *);
    uint32_t (*setsamplerate)(uint32_t);
    uint32_t (*getsamplerate)(void);
    int (*setbasefreq)(uint32_t);
    int (*stop)(void);
    int (*setgain)(float);
    char *(*getlasterrortext)(void);
    int (*readasync)(tsdrplugin_readasync_function, void *);
    void (*cleanup)(void);
    int (*readasync_raw)(tsdrplugin_readasync_raw_function, void *); /* optional extension, NULL when absent */
    int (*memory_stable)(void);                                      /* optional extension (TSDRLibraryExt.h), NULL when absent */
} plugin_host_t;

int plugin_host_load(plugin_host_t *p, const char *path); /* TSDR_OK / TSDR_INCOMPATIBLE_PLUGIN / TSDR_ERR_PLUGIN */
void plugin_host_close(plugin_host_t *p);

struct engine;
void engine_stats(struct engine *e, tsdrx_stats_t *out); /* counters of a running session */

/* Fields that several threads read and write WITHOUT a lock (the reference marks the same ones `volatile`,
 * internaldefinitions.h:30-65) go through these: acquire loads and release stores — the plain loads and stores `volatile`
 * gave on x86, the right barriers on other hosts — and read-modify-writes through __atomic_exchange_n / __atomic_fetch_add,
 * so that a shift or a reset request that arrives between a worker's read and its clear is not lost.  ThreadSanitizer
 * (tests/sanitize) can then tell them from accidents: everything else is owned by one thread or guarded by a mutex. */
#define A_LD(x) __atomic_load_n(&(x), __ATOMIC_ACQUIRE)
#define A_ST(x, v) __atomic_store_n(&(x), (v), __ATOMIC_RELEASE)

struct tsdr_lib {
    plugin_host_t plugin;

    /* geometry — set_internal_samplerate, TSDRLibrary.c:540-550; under `lock` */
    uint32_t samplerate;      /* the rate the pipeline runs at (4x in super-bandwidth mode) */
    uint32_t samplerate_real; /* what the plugin reports (A_LD / A_ST) */
    int width, height;
    double pixelrate, refreshrate, pixeltimeoversampletime;
    float motionblur;         /* under `lock` as well (a float: the frame path reads it together with the geometry) */

    /* A_LD / A_ST */
    int running;       /* workers should keep going */
    int stop_sent;     /* tsdrplugin_stop was called for this run (tsdr_plugin_stop_once): tsdr_stop and a failing
                          engine worker may both want to, the plugin hears it once (under `lock`) */
    int stop_status;   /* what that call returned (under `lock`) */
    int nativerunning; /* tsdr_readasync is on some thread's stack */
    uint32_t centfreq;
    float gain;        /* the API threads' own (never read by a worker) */
    int syncoffset;    /* tsdr_sync adds, the resampler's thread takes it with an exchange */

    /* the last error: any thread that calls into the API sets it, so it has a lock of its own */
    pthread_mutex_t errlock;
    char *errormsg;
    int errormsg_code;

    uint32_t params_int[COUNT_PARAM_INT]; /* A_LD / A_ST */
    double params_double[COUNT_PARAM_DOUBLE]; /* under `lock` (validated and never read, like the reference's) */

    tsdr_value_changed_callback callback;      /* A_LD / A_ST: tsdr_free clears them while a worker may still announce */
    tsdr_on_plot_ready_callback plotready_callback;
    void *callbackctx;

    pthread_mutex_t lock; /* guards geometry + the running/stop handshake */
    pthread_cond_t stopped;

    int detector_purge; /* frameratedetector_flushcachedestimation (A_LD / A_ST) */
    struct engine *eng; /* under `lock` */

    /* TSDRLibraryExt.h: frames as packed RGB for this run (NULL: float frames through the tsdr_readasync callback); set before
     * the workers exist */
    tsdrx_readasync_rgb_function rgb_cb;
    tsdrx_stats_t last_stats; /* of the last session that ended (tsdrx_get_stats; under `lock`) */
    int rgb_inverted;
};

void tsdr_geometry_update(tsdr_lib_t *t, uint32_t samplerate); /* set_internal_samplerate */
void tsdr_announce_value(tsdr_lib_t *t, int id, double a0, double a1);
int tsdr_set_error(tsdr_lib_t *t, int status, const char *msg);
int tsdr_plugin_stop_once(tsdr_lib_t *t); /* tsdrplugin_stop for the current run, at most once, from any thread */

/* engine.c — the streaming pipeline behind tsdr_readasync */
int engine_run(tsdr_lib_t *t, tsdr_readasync_function cb, void *ctx); /* blocks inside the plugin's readasync */

/* dropped-sample bookkeeping (dsp.c:313-368), integer only */
int64_t drop_shift_with(int64_t difference, uint32_t block, int64_t syncoffset);

#endif
