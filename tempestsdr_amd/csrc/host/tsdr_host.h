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

struct tsdr_lib {
    plugin_host_t plugin;

    /* geometry — set_internal_samplerate, TSDRLibrary.c:540-550 */
    uint32_t samplerate;      /* the rate the pipeline runs at (4x in super-bandwidth mode) */
    uint32_t samplerate_real; /* what the plugin reports */
    int width, height;
    double pixelrate, refreshrate, pixeltimeoversampletime;

    volatile int running;       /* workers should keep going */
    volatile int stop_sent;     /* tsdrplugin_stop was called for this run (tsdr_plugin_stop_once): tsdr_stop and a failing
                                   engine worker may both want to, the plugin hears it once */
    int stop_status;            /* what that call returned */
    volatile int nativerunning; /* tsdr_readasync is on some thread's stack */
    uint32_t centfreq;
    float gain, motionblur;
    volatile int syncoffset;

    char *errormsg;
    int errormsg_code;

    volatile uint32_t params_int[COUNT_PARAM_INT];
    double params_double[COUNT_PARAM_DOUBLE];

    tsdr_value_changed_callback callback;
    tsdr_on_plot_ready_callback plotready_callback;
    void *callbackctx;

    pthread_mutex_t lock; /* guards geometry + the running/stop handshake */
    pthread_cond_t stopped;

    volatile int detector_purge; /* frameratedetector_flushcachedestimation */
    struct engine *eng;

    /* TSDRLibraryExt.h: frames as packed RGB for this run (NULL: float frames through the tsdr_readasync callback) */
    tsdrx_readasync_rgb_function rgb_cb;
    tsdrx_stats_t last_stats; /* of the last session that ended (tsdrx_get_stats) */
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
