/* tsdr_api.c — the tsdr_* entry points (include/TSDRLibrary.h).  Behaviour
 * follows TempestSDR/src/TSDRLibrary.c call for call (line references inline);
 * what differs is what sits behind tsdr_readasync: engine.c streams the plugin's
 * IQ blocks to an MI355X instead of four CPU worker threads.
 *
 * Known defects of the reference are NOT reproduced: every field
 * is initialised by tsdr_init, the error text is owned and freed safely, the
 * teardown does not touch freed memory, and the plugin receives a writable copy
 * of its parameter string. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "tsdr_host.h"

#define MAX_ARR_SIZE (4000 * 4000) /* TSDRLibrary.c:31 */
#define MAX_SAMP_RATE (500e6)      /* TSDRLibrary.c:32 */

/* The last error belongs to whichever call set it last — the host's setters on its GUI thread and tsdr_readasync on its
 * own thread both do (the reference's announceexception, TSDRLibrary.c:622-653, reallocs one shared buffer from all of
 * them) — so the text changes hands under a lock of its own: two threads failing at once cannot free it twice. */
int tsdr_set_error(tsdr_lib_t *t, int status, const char *msg)
{
    if (status == TSDR_OK) {
        pthread_mutex_lock(&t->errlock);
        t->errormsg_code = TSDR_OK;
        pthread_mutex_unlock(&t->errlock);
        return status;
    }
    if (!msg)
        msg = "An exception with no detailed explanation cause has occurred. This could as well be a bug in the "
              "TSDRlibrary or in one of its plugins.";
    char *copy = strdup(msg);
    pthread_mutex_lock(&t->errlock);
    t->errormsg_code = status;
    if (copy) {
        free(t->errormsg);
        t->errormsg = copy;
    }
    pthread_mutex_unlock(&t->errlock);
    return status;
}

static int ok(tsdr_lib_t *t) { return tsdr_set_error(t, TSDR_OK, NULL); }

static int plugin_result(tsdr_lib_t *t, int status)
{
    if (status == TSDR_OK) return ok(t);
    return tsdr_set_error(t, status, t->plugin.getlasterrortext ? t->plugin.getlasterrortext() : NULL);
}

void tsdr_announce_value(tsdr_lib_t *t, int id, double a0, double a1)
{
    tsdr_value_changed_callback cb = A_LD(t->callback);
    if (cb) cb(id, a0, a1, t->callbackctx);
}

/* set_internal_samplerate, TSDRLibrary.c:540-550 (expression for expression) */
void tsdr_geometry_update(tsdr_lib_t *t, uint32_t samplerate)
{
    t->samplerate = samplerate;
    if (t->height <= 0 || !(t->refreshrate > 0)) return; /* tsdr_setresolution not called yet */
    const double real_width = samplerate / (t->refreshrate * t->height);
    t->width = (int)2 * real_width;
    t->pixelrate = t->width * t->height * t->refreshrate;
    if (t->samplerate != 0 && t->pixelrate != 0) t->pixeltimeoversampletime = ((double)t->samplerate) / t->pixelrate;
}

void tsdr_init(tsdr_lib_t **out, tsdr_value_changed_callback callback, tsdr_on_plot_ready_callback plotready_callback, void *ctx)
{
    tsdr_lib_t *t = (tsdr_lib_t *)calloc(1, sizeof(*t));
    *out = t;
    if (!t) return;
    {   /* Opt-in: TSDR_GPU_HW_QUEUES=n is forwarded to the HIP runtime as GPU_MAX_HW_QUEUES (read at the process's
         * first HIP call; 2 is the streaming optimum, tsdrgpu_core.hip) unless the host chose a value itself.  Nothing
         * is changed without it: a library must not reconfigure its host process behind its back. */
        const char *q = getenv("TSDR_GPU_HW_QUEUES");
        if (q && q[0] >= '1' && q[0] <= '9') setenv("GPU_MAX_HW_QUEUES", q, 0);
    }
    t->callback = callback;
    t->plotready_callback = plotready_callback;
    t->callbackctx = ctx;
    t->errormsg_code = TSDR_OK;
    pthread_mutex_init(&t->lock, NULL);
    pthread_mutex_init(&t->errlock, NULL);
    pthread_cond_init(&t->stopped, NULL);
}

void tsdr_free(tsdr_lib_t **pt)
{
    if (!pt || !*pt) return;
    tsdr_lib_t *t = *pt;
    A_ST(t->callback, NULL);
    A_ST(t->plotready_callback, NULL);
    tsdr_stop(t); /* also waits while tsdr_readasync is still tearing the pipeline down on its own */
    plugin_host_close(&t->plugin);
    free(t->errormsg);
    pthread_cond_destroy(&t->stopped);
    pthread_mutex_destroy(&t->lock);
    pthread_mutex_destroy(&t->errlock);
    free(t);
    *pt = NULL;
}

void tsdr_reset(tsdr_lib_t *t) /* TSDRLibrary.c:118-133: the DSP state itself lives in the engine and is rebuilt per run */
{
    A_ST(t->syncoffset, 0);
}

void *tsdr_getctx(tsdr_lib_t *t) { return t->callbackctx; }
int tsdr_isrunning(tsdr_lib_t *t) { return A_LD(t->nativerunning); }

char *tsdr_getlasterrortext(tsdr_lib_t *t)
{
    pthread_mutex_lock(&t->errlock);
    char *text = (t->errormsg_code == TSDR_OK) ? NULL : t->errormsg; /* (valid until the next failing call, like the reference's) */
    pthread_mutex_unlock(&t->errlock);
    return text;
}

int tsdr_getsamplerate(tsdr_lib_t *t) /* TSDRLibrary.c:181-193 */
{
    if (!t->plugin.loaded) return tsdr_set_error(t, TSDR_ERR_PLUGIN, "Cannot change sample rate. Plugin not loaded yet.");
    const uint32_t real = t->plugin.getsamplerate();
    A_ST(t->samplerate_real, real);
    if (real == 0 || real > MAX_SAMP_RATE)
        return tsdr_set_error(t, TSDR_SAMPLE_RATE_WRONG, "Invalid/unsupported value for sample rate.");
    pthread_mutex_lock(&t->lock);
    tsdr_geometry_update(t, real);
    pthread_mutex_unlock(&t->lock);
    return ok(t);
}

int tsdr_setbasefreq(tsdr_lib_t *t, uint32_t freq) /* TSDRLibrary.c:195-205 */
{
    A_ST(t->centfreq, freq);
    if (!t->plugin.loaded) return ok(t);
    /* frameratedetector_flushcachedestimation, frameratedetector.c:197-201 */
    A_ST(t->detector_purge, 1);
    A_ST(t->params_int[PARAM_AUTOCORR_PLOTS_RESET], 2);
    return plugin_result(t, t->plugin.setbasefreq(freq));
}

int tsdr_setgain(tsdr_lib_t *t, float gain) /* TSDRLibrary.c:226-237 */
{
    pthread_mutex_lock(&t->lock);
    t->gain = gain;
    pthread_mutex_unlock(&t->lock);
    if (!t->plugin.loaded) return ok(t);
    return plugin_result(t, t->plugin.setgain(gain));
}

int tsdr_unloadplugin(tsdr_lib_t *t) /* TSDRLibrary.c:425-435 */
{
    if (!t->plugin.loaded) return tsdr_set_error(t, TSDR_ERR_PLUGIN, "No plugin has been loaded so it can't be unloaded");
    if (A_LD(t->nativerunning) || A_LD(t->running))
        return tsdr_set_error(t, TSDR_ALREADY_RUNNING, "The library is already running in async mode. Stop it first!");
    plugin_host_close(&t->plugin);
    return ok(t);
}

int tsdr_loadplugin(tsdr_lib_t *t, const char *path, const char *params) /* TSDRLibrary.c:437-465 */
{
    if (A_LD(t->nativerunning) || A_LD(t->running))
        return tsdr_set_error(t, TSDR_ALREADY_RUNNING, "The library is already running in async mode. Stop it first!");
    plugin_host_close(&t->plugin);
    int status = plugin_host_load(&t->plugin, path);
    if (status == TSDR_INCOMPATIBLE_PLUGIN)
        return tsdr_set_error(t, status, "The plugin cannot be loaded. It is incompatible or there are depending libraries "
                                         "missing. Please check the readme file that comes with the plugin.");
    if (status != TSDR_OK) return tsdr_set_error(t, status, "The selected library is not a valid TSDR plugin!");

    char name[256];
    t->plugin.getName(name);
    char *writable = strdup(params ? params : ""); /* RawFile tokenises its argument in place */
    status = writable ? t->plugin.init(writable) : TSDR_ERR_PLUGIN;
    free(writable);
    if (status != TSDR_OK) {
        tsdr_set_error(t, status, t->plugin.getlasterrortext());
        plugin_host_close(&t->plugin);
        return status;
    }
    return ok(t);
}

int tsdr_setresolution(tsdr_lib_t *t, int height, double refreshrate) /* TSDRLibrary.c:552-565 */
{
    if (height <= 0 || refreshrate <= 0)
        return tsdr_set_error(t, TSDR_WRONG_VIDEOPARAMS, "The supplied height is invalid or refreshrate is negative!");
    pthread_mutex_lock(&t->lock);
    t->height = height;
    t->refreshrate = refreshrate;
    if (t->plugin.loaded) tsdr_geometry_update(t, t->samplerate);
    pthread_mutex_unlock(&t->lock);
    return ok(t);
}

int tsdr_motionblur(tsdr_lib_t *t, float coeff) /* TSDRLibrary.c:568-574 */
{
    if (coeff < 0.0f || coeff > 1.0f) return TSDR_WRONG_VIDEOPARAMS;
    pthread_mutex_lock(&t->lock);
    t->motionblur = coeff;
    pthread_mutex_unlock(&t->lock);
    return ok(t);
}

int tsdr_sync(tsdr_lib_t *t, int pixels, int direction) /* TSDRLibrary.c:576-602 */
{
    if (pixels == 0) return TSDR_OK;
    pthread_mutex_lock(&t->lock);
    const int width = t->width, height = t->height;
    pthread_mutex_unlock(&t->lock);
    int shift = 0;
    switch (direction) {
        case DIRECTION_CUSTOM:
            shift = pixels;
            break;
        case DIRECTION_UP:
            if (pixels > height || pixels < 0)
                return tsdr_set_error(t, TSDR_WRONG_VIDEOPARAMS, "Cannot shift up with more pixels than the height of the image or shift is negative!");
            shift = pixels * width;
            break;
        case DIRECTION_DOWN:
            if (pixels > height || pixels < 0)
                return tsdr_set_error(t, TSDR_WRONG_VIDEOPARAMS, "Cannot shift down with more pixels than the height of the image or shift is negative!");
            shift = -pixels * width;
            break;
        case DIRECTION_LEFT:
            if (pixels > width || pixels < 0)
                return tsdr_set_error(t, TSDR_WRONG_VIDEOPARAMS, "Cannot shift to the left with more pixels than the width of the image or shift is negative!");
            shift = pixels;
            break;
        case DIRECTION_RIGHT:
            if (pixels > width || pixels < 0)
                return tsdr_set_error(t, TSDR_WRONG_VIDEOPARAMS, "Cannot shift to the right with more pixels than the width of the image or shift is negative!");
            shift = -pixels;
            break;
    }
    /* the resampler's thread takes the pending shift with an exchange (engine.c): added here in one step, nothing is lost
     * between the two (the reference's `syncoffset +=` against `syncoffset = 0`, TSDRLibrary.c:345-346, can lose one) */
    __atomic_fetch_add(&t->syncoffset, shift, __ATOMIC_ACQ_REL);
    return ok(t);
}

int tsdr_setparameter_int(tsdr_lib_t *t, int parameter, uint32_t value) /* TSDRLibrary.c:604-611 */
{
    if (parameter < 0 || parameter >= COUNT_PARAM_INT) return tsdr_set_error(t, TSDR_INVALID_PARAMETER, "Invalid integer parameter id");
    A_ST(t->params_int[parameter], value);
    return ok(t);
}

int tsdr_setparameter_double(tsdr_lib_t *t, int parameter, double value) /* TSDRLibrary.c:613-620 */
{
    if (parameter < 0 || parameter >= COUNT_PARAM_DOUBLE)
        return tsdr_set_error(t, TSDR_INVALID_PARAMETER, "Invalid double floating point parameter id");
    pthread_mutex_lock(&t->lock);
    t->params_double[parameter] = value; /* the reference validates the id and discards the value */
    pthread_mutex_unlock(&t->lock);
    return ok(t);
}

/* tsdr_stop (host thread) and an engine worker that saw a device call fail may decide to stop the plugin at the same
 * moment: whoever comes first makes the call, the other one takes its result. */
int tsdr_plugin_stop_once(tsdr_lib_t *t)
{
    if (!t->plugin.loaded || !t->plugin.stop) return TSDR_OK;
    pthread_mutex_lock(&t->lock);
    const int first = !t->stop_sent;
    if (first) {
        t->stop_sent = 1; /* 1: the call is being made, 2: its status is known */
        pthread_mutex_unlock(&t->lock);
        const int status = t->plugin.stop();
        pthread_mutex_lock(&t->lock);
        t->stop_status = status;
        t->stop_sent = 2;
        pthread_cond_broadcast(&t->stopped);
    } else {
        /* the second caller reports what the first one's call RETURNED, not what stop_status held before it did */
        while (t->stop_sent == 1) pthread_cond_wait(&t->stopped, &t->lock);
    }
    const int status = t->stop_status;
    pthread_mutex_unlock(&t->lock);
    return status;
}

int tsdr_stop(tsdr_lib_t *t) /* TSDRLibrary.c:213-224 */
{
    pthread_mutex_lock(&t->lock);
    const int was_running = A_LD(t->running);
    pthread_mutex_unlock(&t->lock);
    const int status = (was_running && t->plugin.loaded) ? tsdr_plugin_stop_once(t) : TSDR_OK;
    /* Wait until tsdr_readasync has torn the pipeline down — also when the plugin's readasync returned on its
     * own and the teardown is merely still in progress (running already 0, nativerunning still 1): nobody may
     * unload the plugin or free the library under it. */
    pthread_mutex_lock(&t->lock);
    A_ST(t->running, 0);
    while (A_LD(t->nativerunning)) pthread_cond_wait(&t->stopped, &t->lock);
    pthread_mutex_unlock(&t->lock);
    if (!was_running) return ok(t);
    return plugin_result(t, status);
}

static int readasync_common(tsdr_lib_t *t, tsdr_readasync_function cb, tsdrx_readasync_rgb_function rgb_cb, int inverted, void *ctx)
{
    pthread_mutex_lock(&t->lock);
    if (A_LD(t->nativerunning) || A_LD(t->running)) {
        pthread_mutex_unlock(&t->lock);
        return tsdr_set_error(t, TSDR_ALREADY_RUNNING, "The library is already running in async mode. Stop it first!");
    }
    if (!t->plugin.loaded) {
        pthread_mutex_unlock(&t->lock);
        return tsdr_set_error(t, TSDR_ERR_PLUGIN, "Please load a working plugin first!");
    }
    tsdr_reset(t);
    t->rgb_cb = rgb_cb;
    t->rgb_inverted = inverted;
    A_ST(t->nativerunning, 1);
    A_ST(t->running, 1);
    t->stop_sent = 0;
    t->stop_status = TSDR_OK;
    pthread_mutex_unlock(&t->lock);

    int status = tsdr_getsamplerate(t);
    double ptost = 0.0;
    if (status == TSDR_OK) {
        pthread_mutex_lock(&t->lock);
        const int width = t->width, height = t->height;
        ptost = t->pixeltimeoversampletime;
        pthread_mutex_unlock(&t->lock);
        const long long size = (long long)width * height;
        if (width <= 0 || height <= 0 || size > MAX_ARR_SIZE)
            status = tsdr_set_error(t, TSDR_WRONG_VIDEOPARAMS, "The supplied height and the width are invalid!");
    }
    if (status == TSDR_OK) status = tsdr_setbasefreq(t, A_LD(t->centfreq));
    if (status == TSDR_OK) {
        pthread_mutex_lock(&t->lock);
        const float gain = t->gain;
        pthread_mutex_unlock(&t->lock);
        status = tsdr_setgain(t, gain);
    }
    if (status == TSDR_OK && ptost > 0) status = engine_run(t, cb, ctx);

    pthread_mutex_lock(&t->lock);
    A_ST(t->running, 0);
    A_ST(t->nativerunning, 0);
    pthread_cond_broadcast(&t->stopped);
    pthread_mutex_unlock(&t->lock);
    return status;
}

int tsdr_readasync(tsdr_lib_t *t, tsdr_readasync_function cb, void *ctx) /* TSDRLibrary.c:467-536 */
{
    return readasync_common(t, cb, NULL, 0, ctx);
}

#pragma GCC visibility push(default)
int tsdrx_get_stats(tsdr_lib_t *t, tsdrx_stats_t *out) /* TSDRLibraryExt.h */
{
    if (!t || !out) return TSDR_INVALID_PARAMETER;
    pthread_mutex_lock(&t->lock);
    struct engine *e = (struct engine *)t->eng;
    if (e) engine_stats(e, out); /* (counters of other threads, read without their locks: a snapshot, not a barrier) */
    else *out = t->last_stats;
    pthread_mutex_unlock(&t->lock);
    return TSDR_OK;
}

int tsdrx_readasync_rgb(tsdr_lib_t *t, tsdrx_readasync_rgb_function cb, void *ctx, int inverted) /* TSDRLibraryExt.h */
{
    if (!cb) return tsdr_set_error(t, TSDR_WRONG_VIDEOPARAMS, "tsdrx_readasync_rgb needs a callback");
    return readasync_common(t, NULL, cb, inverted ? 1 : 0, ctx);
}
#pragma GCC visibility pop

/* dsp.c:321-324,354-368 */
static uint64_t drop_comp(const int block, const int dropped)
{
    const uint64_t frames = dropped / block;
    return ((frames + 1) * block - dropped) % block;
}

int64_t drop_shift_with(int64_t difference, uint32_t block, int64_t syncoffset)
{
    if (block == 0) return difference;
    if (syncoffset >= 0) difference -= syncoffset % block;
    else difference -= block + syncoffset % block;
    if (difference < 0) difference = (int64_t)drop_comp((int)block, (int)-difference);
    return difference;
}
