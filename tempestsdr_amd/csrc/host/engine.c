/* engine.c — the streaming pipeline behind tsdr_readasync.
 *
 * The reference moves data plugin-thread -> decimatingthread ->
 * postprocessingthread -> videodecodingthread through three float ring buffers,
 * with frameratedetector_thread and super_thread on the side
 * (TempestSDR/src/TSDRLibrary.c:264-418, frameratedetector.c:128-187,
 * superbandwidth.c:154-254).  Here:
 *
 *   plugin thread   on_block(): copy the IQ block into a pinned slot, return
 *   device thread   upload, then queue fused demod+resample, batched frame
 *                   post-processing and the autocorrelation on the GPU stream
 *   video thread    frame callback          (reference: videodecodingthread)
 *   plot thread     plot + value callbacks  (reference: frameratedetector_thread)
 *
 * Sample / pixel skipping after drops follows dsp_dropped_compensation_*
 * (dsp.c:313-368) so frames stay aligned, and back-pressure is lossy in whole
 * blocks like the reference's circular buffers.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "tsdr_host.h"

#define NSLOT 8          /* pinned input blocks in flight */
#define NFRAMEQ 4        /* frames waiting for the video callback */
#define MAX_FRAME_BATCH 8
#define NORMALISATION_LOWPASS_COEFF (0.1f) /* TSDRLibrary.c:37 */
#define FRAMES_TO_POLL (0.1)               /* TSDRLibrary.c:41 */
#define AUTOGAIN_REPORT_EVERY_FRAMES (5)   /* dsp.c:20 */

/* super-bandwidth state machine, superbandwidth.c:22-33 */
enum { SUPER_STOPPED, SUPER_STARTING, SUPER_GATHERING, SUPER_PAUSE };
#define SUPER_HOPS 4
#define SUPER_FRAMES_TO_RECORD 10
#define SUPER_SECS_TO_PAUSE 0.5

typedef struct {
    float *d;          /* device buffer, `cap` floats */
    size_t cap, rd, wr; /* valid data: [rd, wr) */
    float *d_alt;      /* compaction target (ping-pong) */
} devstream_t;

typedef struct {
    float *h; /* pinned */
    size_t cap;
    int width, height;
    int ready;
} frame_slot_t;

struct engine {
    tsdr_lib_t *t;
    tsdr_readasync_function cb;
    void *cbctx;

    tsdrgpu_t *g;
    tsdrgpu_resampler_t *rs;
    tsdrgpu_postproc_t *pp;
    tsdrgpu_autocorr_t *ac;
    uint32_t ac_rate;

    /* input queue */
    struct { float *h; size_t cap, nfloats; int64_t dropped; } slot[NSLOT];
    int q_head, q_count;
    int64_t pending_drop;
    pthread_mutex_t qm;
    pthread_cond_t q_nonempty;

    float *d_block; size_t block_cap;
    devstream_t iq;   /* samples for the resampler (interleaved IQ; magnitude in super mode) */
    int iq_is_mag;
    devstream_t det;  /* samples for the frame-rate detector */
    devstream_t pix;  /* resampled pixel stream */
    float *d_rs; size_t rs_cap;
    float *d_out; size_t out_cap;

    int64_t dev_difference; /* samples still to skip (process(), TSDRLibrary.c:284-295) */
    int64_t pix_difference; /* pixels still to skip (decimatingthread, TSDRLibrary.c:342-346) */
    int pp_runs;
    int last_w, last_h;

    /* video delivery */
    frame_slot_t fq[NFRAMEQ];
    int fq_head, fq_count;
    pthread_mutex_t fm;
    pthread_cond_t f_nonempty;

    /* plot delivery */
    double *h_frameplot, *h_lineplot;
    int32_t flo, flen, llo, llen;
    uint64_t plot_calls;
    int plot_pending, plot_reset_announce, plot_dumped_announce;
    pthread_mutex_t pm;
    pthread_cond_t p_nonempty;

    volatile int alive; /* delivery threads keep going */

    /* super-bandwidth */
    int super_state, super_hop, super_gathered, super_to_gather, super_frame, super_to_pause;
    uint32_t super_rate;
    float *d_hops[SUPER_HOPS];
    float *d_super_out; size_t super_out_cap;
};

/* ---- small helpers ----------------------------------------------------------- */
/* Contract-exact modes are the default; `name`=0 (or TSDR_GPU_EXACT=0 for all of them) switches one off. */
static int exact_wanted(const char *name)
{
    const char *one = getenv(name), *all = getenv("TSDR_GPU_EXACT");
    if (one && one[0]) return one[0] != '0';
    if (all && all[0]) return all[0] != '0';
    return 1;
}

static int gpu_ok(struct engine *e, int rc, const char *what)
{
    if (rc == 0) return 1;
    fprintf(stderr, "tsdr: %s failed (%d): %s\n", what, rc, tsdrgpu_last_error(e->g));
    return 0;
}

static int stream_reserve(struct engine *e, devstream_t *s, size_t extra)
{
    if (s->wr + extra <= s->cap) return 1;
    const size_t live = s->wr - s->rd;
    size_t need = live + extra;
    if (need <= s->cap && s->d_alt) { /* compact into the twin buffer */
        if (live && !gpu_ok(e, tsdrgpu_copy(e->g, s->d_alt, s->d + s->rd, live * sizeof(float)), "compact")) return 0;
        float *tmp = s->d; s->d = s->d_alt; s->d_alt = tmp;
        s->rd = 0; s->wr = live;
        return 1;
    }
    size_t cap = need * 2 + 4096;
    float *n1 = NULL, *n2 = NULL;
    if (tsdrgpu_alloc(e->g, (void **)&n1, cap * sizeof(float)) || tsdrgpu_alloc(e->g, (void **)&n2, cap * sizeof(float))) return 0;
    if (live && !gpu_ok(e, tsdrgpu_copy(e->g, n1, s->d + s->rd, live * sizeof(float)), "grow")) return 0;
    tsdrgpu_sync(e->g);
    tsdrgpu_free(e->g, s->d);
    tsdrgpu_free(e->g, s->d_alt);
    s->d = n1; s->d_alt = n2; s->cap = cap; s->rd = 0; s->wr = live;
    return 1;
}

static int stream_append(struct engine *e, devstream_t *s, const float *d_src, size_t n)
{
    if (!n) return 1;
    if (!stream_reserve(e, s, n)) return 0;
    if (!gpu_ok(e, tsdrgpu_copy(e->g, s->d + s->wr, d_src, n * sizeof(float)), "append")) return 0;
    s->wr += n;
    return 1;
}

static void stream_free(struct engine *e, devstream_t *s)
{
    tsdrgpu_free(e->g, s->d);
    tsdrgpu_free(e->g, s->d_alt);
    memset(s, 0, sizeof(*s));
}

static int ensure_dev(struct engine *e, float **buf, size_t *cap, size_t need)
{
    if (*cap >= need) return 1;
    tsdrgpu_sync(e->g);
    tsdrgpu_free(e->g, *buf);
    *buf = NULL; *cap = 0;
    if (tsdrgpu_alloc(e->g, (void **)buf, (need + need / 4) * sizeof(float))) return 0;
    *cap = need + need / 4;
    return 1;
}

/* ---- plugin thread -------------------------------------------------------------- */
static void on_block(float *buf, uint64_t items, void *ctx, int64_t dropped)
{
    struct engine *e = (struct engine *)ctx;
    if (!e->t->running || (items & 1)) return;
    pthread_mutex_lock(&e->qm);
    if (e->q_count == NSLOT) { /* device thread is behind: lose the whole block */
        e->pending_drop += (int64_t)(items / 2) + dropped;
        pthread_mutex_unlock(&e->qm);
        return;
    }
    const int s = (e->q_head + e->q_count) % NSLOT;
    if (e->slot[s].cap < items) {
        pthread_mutex_unlock(&e->qm); /* allocation outside the lock; only this thread produces */
        float *h = NULL;
        if (tsdrgpu_alloc_host(e->g, (void **)&h, (size_t)items * sizeof(float))) return;
        pthread_mutex_lock(&e->qm);
        tsdrgpu_free_host(e->g, e->slot[s].h);
        e->slot[s].h = h;
        e->slot[s].cap = items;
    }
    if (items) memcpy(e->slot[s].h, buf, (size_t)items * sizeof(float));
    e->slot[s].nfloats = items;
    e->slot[s].dropped = dropped + e->pending_drop;
    e->pending_drop = 0;
    e->q_count++;
    pthread_cond_signal(&e->q_nonempty);
    pthread_mutex_unlock(&e->qm);
}

/* ---- video thread ---------------------------------------------------------------- */
static void *video_thread(void *arg)
{
    struct engine *e = (struct engine *)arg;
    pthread_mutex_lock(&e->fm);
    while (e->alive || e->fq_count) {
        if (!e->fq_count) {
            struct timespec ts;
            clock_gettime(CLOCK_REALTIME, &ts);
            ts.tv_nsec += 30 * 1000000L;
            if (ts.tv_nsec >= 1000000000L) { ts.tv_sec++; ts.tv_nsec -= 1000000000L; }
            pthread_cond_timedwait(&e->f_nonempty, &e->fm, &ts);
            continue;
        }
        frame_slot_t *f = &e->fq[e->fq_head];
        pthread_mutex_unlock(&e->fm);
        if (e->t->running) e->cb(f->h, f->width, f->height, e->cbctx);
        pthread_mutex_lock(&e->fm);
        e->fq_head = (e->fq_head + 1) % NFRAMEQ;
        e->fq_count--;
    }
    pthread_mutex_unlock(&e->fm);
    return NULL;
}

/* ---- plot thread ------------------------------------------------------------------- */
static void *plot_thread(void *arg)
{
    struct engine *e = (struct engine *)arg;
    tsdr_lib_t *t = e->t;
    pthread_mutex_lock(&e->pm);
    while (e->alive) {
        if (!e->plot_pending && !e->plot_reset_announce && !e->plot_dumped_announce) {
            struct timespec ts;
            clock_gettime(CLOCK_REALTIME, &ts);
            ts.tv_nsec += 30 * 1000000L;
            if (ts.tv_nsec >= 1000000000L) { ts.tv_sec++; ts.tv_nsec -= 1000000000L; }
            pthread_cond_timedwait(&e->p_nonempty, &e->pm, &ts);
            continue;
        }
        const int reset = e->plot_reset_announce, dumped = e->plot_dumped_announce, plots = e->plot_pending;
        e->plot_reset_announce = e->plot_dumped_announce = 0;
        /* the arrays stay untouched by the device thread while plot_pending is set */
        pthread_mutex_unlock(&e->pm);
        if (reset) tsdr_announce_value(t, VALUE_ID_AUTOCORRECT_RESET, 0, 0);
        if (dumped) tsdr_announce_value(t, VALUE_ID_AUTOCORRECT_DUMPED, 0, 0);
        if (plots) {
            tsdr_on_plot_ready_callback pcb = t->plotready_callback;
            if (pcb) { /* frameratedetector.c:121-124 */
                pcb(PLOT_ID_FRAME, e->flo, e->h_frameplot, e->flen, e->ac_rate, t->callbackctx);
                pcb(PLOT_ID_LINE, e->llo, e->h_lineplot, e->llen, e->ac_rate, t->callbackctx);
            }
            tsdr_announce_value(t, VALUE_ID_AUTOCORRECT_FRAMES_COUNT, 0, (double)e->plot_calls);
        }
        pthread_mutex_lock(&e->pm);
        if (plots) e->plot_pending = 0;
    }
    pthread_mutex_unlock(&e->pm);
    return NULL;
}

/* ---- device thread: stages ------------------------------------------------------------ */
static void dump_autocorr(struct engine *e) /* dump_autocorrect, frameratedetector.c:64-85 */
{
    const float *d_corr = NULL;
    uint32_t n = 0;
    if (tsdrgpu_autocorr_last_corr(e->ac, &d_corr, &n)) return;
    const uint32_t maxels = n / 2; /* fft_getrealsize(size)/2 floats */
    float *h = (float *)malloc(sizeof(float) * (maxels + 2));
    if (!h) return;
    if (tsdrgpu_download(e->g, h, d_corr, sizeof(float) * (maxels + 2)) == 0 && tsdrgpu_sync(e->g) == 0) {
        FILE *f = fopen("autocorr.csv", "w");
        if (f) {
            fprintf(f, "%s, %s\n", "ms", "dB");
            for (uint32_t i = 0; i < maxels; i += 2) {
                const double re = h[i], im = h[i + 1];
                fprintf(f, "%f, %f\n", 1000.0 * (i / 2) / (double)e->ac_rate, 10.0 * log10(sqrt(re * re + im * im)));
            }
            fclose(f);
        }
    }
    free(h);
}

static void run_detector(struct engine *e, uint32_t fs)
{
    tsdr_lib_t *t = e->t;
    if (t->params_int[PARAM_AUTOCORR_PLOTS_OFF]) { e->det.rd = e->det.wr = 0; return; }
    if (!e->ac || e->ac_rate != fs) {
        if (e->ac) tsdrgpu_autocorr_destroy(e->ac);
        e->ac = NULL;
        if (tsdrgpu_autocorr_create(e->g, &e->ac, fs)) return; /* rate too low for the lag windows */
        e->ac_rate = fs;
        /* The frame-rate detector runs in the reference's own FFT arithmetic by default: plots, their argmax and so
         * the detected mode are bit-identical to the CPU library's (0.24 ms per 100 MS/s window against 56 ms of
         * signal).  TSDR_GPU_EXACT_AUTOCORR=0 / TSDR_GPU_EXACT=0 select the fast transform (plots within 1e-4*max). */
        if (exact_wanted("TSDR_GPU_EXACT_AUTOCORR")) (void)tsdrgpu_autocorr_set_exact(e->ac, 1);
        uint32_t cap, n;
        tsdrgpu_autocorr_geometry(e->ac, &e->flo, &e->flen, &e->llo, &e->llen, &cap, &n);
        pthread_mutex_lock(&e->pm);
        while (e->plot_pending && e->alive) { pthread_mutex_unlock(&e->pm); struct timespec ts = {0, 1000000}; nanosleep(&ts, NULL); pthread_mutex_lock(&e->pm); }
        tsdrgpu_free_host(e->g, e->h_frameplot);
        tsdrgpu_free_host(e->g, e->h_lineplot);
        e->h_frameplot = e->h_lineplot = NULL;
        tsdrgpu_alloc_host(e->g, (void **)&e->h_frameplot, sizeof(double) * e->flen);
        tsdrgpu_alloc_host(e->g, (void **)&e->h_lineplot, sizeof(double) * e->llen);
        pthread_mutex_unlock(&e->pm);
    }
    uint32_t capture = 0;
    tsdrgpu_autocorr_geometry(e->ac, NULL, NULL, NULL, NULL, &capture, NULL);
    const size_t per = e->iq_is_mag ? 1 : 2;
    while ((e->det.wr - e->det.rd) / per >= capture && t->running) {
        if (t->detector_purge) { /* frameratedetector.c:171-176 */
            t->detector_purge = 0;
            tsdrgpu_autocorr_reset(e->ac);
        }
        if (t->params_int[PARAM_AUTOCORR_PLOTS_RESET]) { /* frameratedetector.c:97-104 */
            const uint32_t orig = t->params_int[PARAM_AUTOCORR_PLOTS_RESET];
            t->params_int[PARAM_AUTOCORR_PLOTS_RESET] = 0;
            tsdrgpu_autocorr_reset(e->ac);
            if (orig == 1) {
                pthread_mutex_lock(&e->pm);
                e->plot_reset_announce = 1;
                pthread_cond_signal(&e->p_nonempty);
                pthread_mutex_unlock(&e->pm);
            }
        }
        if (!gpu_ok(e, tsdrgpu_autocorr_run(e->ac, e->det.d + e->det.rd, !e->iq_is_mag, capture, 1, 0), "autocorr")) return;
        e->det.rd += (size_t)capture * per;
        if (t->params_int[PARAM_AUTOCORR_DUMP]) {
            t->params_int[PARAM_AUTOCORR_DUMP] = 0;
            dump_autocorr(e);
            pthread_mutex_lock(&e->pm);
            e->plot_dumped_announce = 1;
            pthread_mutex_unlock(&e->pm);
        }
        pthread_mutex_lock(&e->pm);
        const int busy = e->plot_pending; /* host still showing the previous plot: skip this update */
        pthread_mutex_unlock(&e->pm);
        if (!busy && e->h_frameplot && e->h_lineplot) {
            uint64_t calls = 0;
            if (tsdrgpu_autocorr_plots(e->ac, e->h_frameplot, e->h_lineplot, &calls) == 0) {
                pthread_mutex_lock(&e->pm);
                e->plot_calls = calls;
                e->plot_pending = 1;
                pthread_cond_signal(&e->p_nonempty);
                pthread_mutex_unlock(&e->pm);
            }
        }
    }
}

static void deliver_frames(struct engine *e, int F, int W, int H, const tsdrgpu_pp_frameinfo_t *info)
{
    tsdr_lib_t *t = e->t;
    const size_t P = (size_t)W * H;
    for (int f = 0; f < F; f++) {
        /* dsp.c:231-235: autogain values every 7th frame */
        if (e->pp_runs++ > AUTOGAIN_REPORT_EVERY_FRAMES) {
            e->pp_runs = 0;
            tsdr_announce_value(t, VALUE_ID_AUTOGAIN_VALUES, info[f].lastmin, info[f].lastmax);
        }
        if (info[f].pll_fired) { /* syncdetector.c:149-151 */
            pthread_mutex_lock(&t->lock);
            t->refreshrate -= info[f].frameratediff;
            tsdr_geometry_update(t, t->samplerate);
            const double rate = t->refreshrate;
            pthread_mutex_unlock(&t->lock);
            tsdr_announce_value(t, VALUE_ID_PLL_FRAMERATE, rate, 0);
        }
        pthread_mutex_lock(&e->fm);
        if (e->fq_count == NFRAMEQ) { /* viewer is slower than the stream: drop the frame */
            pthread_mutex_unlock(&e->fm);
            continue;
        }
        frame_slot_t *s = &e->fq[(e->fq_head + e->fq_count) % NFRAMEQ];
        pthread_mutex_unlock(&e->fm);
        if (s->cap < P) {
            tsdrgpu_free_host(e->g, s->h);
            s->h = NULL; s->cap = 0;
            if (tsdrgpu_alloc_host(e->g, (void **)&s->h, P * sizeof(float))) continue;
            s->cap = P;
        }
        if (tsdrgpu_download(e->g, s->h, e->d_out + (size_t)f * P, P * sizeof(float)) || tsdrgpu_sync(e->g)) continue;
        s->width = W; s->height = H;
        pthread_mutex_lock(&e->fm);
        e->fq_count++;
        pthread_cond_signal(&e->f_nonempty);
        pthread_mutex_unlock(&e->fm);
    }
}

static void run_frames(struct engine *e)
{
    tsdr_lib_t *t = e->t;
    while (t->running) {
        pthread_mutex_lock(&t->lock);
        const int W = t->width, H = t->height;
        pthread_mutex_unlock(&t->lock);
        if (W <= 0 || H <= 0) return;
        const size_t P = (size_t)W * H;
        size_t avail = e->pix.wr - e->pix.rd;
        if (avail < P) return;
        int F = (int)(avail / P);
        if (F > MAX_FRAME_BATCH) F = MAX_FRAME_BATCH;
        tsdrgpu_pp_params_t prm;
        prm.lowpass_before_sync = (int)t->params_int[PARAM_LOW_PASS_BEFORE_SYNC];
        prm.autogain_after_proc = (int)t->params_int[PARAM_AUTOGAIN_AFTER_PROCESSING];
        prm.autoshift = (int)t->params_int[PARAM_INT_AUTOSHIFT];
        prm.pll = (int)t->params_int[PARAM_INT_FRAMERATE_PLL];
        prm.superresolution = (int)t->params_int[PARAM_AUTOCORR_SUPERRESOLUTION];
        prm.motionblur = t->motionblur;
        prm.lowpasscoeff = NORMALISATION_LOWPASS_COEFF;
        if (prm.pll) F = 1; /* the PLL's nudge feeds back into the geometry between frames */
        if (!ensure_dev(e, &e->d_out, &e->out_cap, P * (size_t)F)) return;
        tsdrgpu_pp_frameinfo_t info[MAX_FRAME_BATCH];
        if (!gpu_ok(e, tsdrgpu_postproc_run(e->pp, e->pix.d + e->pix.rd, F, W, H, &prm, e->d_out, info), "postproc")) return;
        e->pix.rd += P * (size_t)F;
        deliver_frames(e, F, W, H, info);
    }
}

static void run_resampler(struct engine *e)
{
    tsdr_lib_t *t = e->t;
    const size_t per = e->iq_is_mag ? 1 : 2;
    for (;;) {
        /* the decimating thread re-reads the geometry for every chunk (TSDRLibrary.c:335-340): the frame-rate
         * PLL and tsdr_setresolution change it while the stream runs */
        pthread_mutex_lock(&t->lock);
        const int W = t->width, H = t->height;
        const double refresh = t->refreshrate;
        const uint32_t fs = t->samplerate;
        pthread_mutex_unlock(&t->lock);
        if (W <= 0 || H <= 0 || !(refresh > 0)) return;
        const int chunk = (int)(FRAMES_TO_POLL * fs / refresh); /* TSDRLibrary.c:335 */
        if (chunk <= 0) return;
        const double up = W * H * refresh, down = fs; /* TSDRLibrary.c:340 */
        const int totalpixels = W * H;
        const size_t have = (e->iq.wr - e->iq.rd) / per;
        int nchunks = (int)(have / (size_t)chunk);
        if (nchunks <= 0) break;
        /* while pixels are being skipped or a manual shift is pending go chunk by chunk like the reference;
         * with the PLL on as well, because a frame completed by this chunk may nudge the refresh rate, which
         * the next chunk's ratio must already see */
        if (e->pix_difference != 0 || t->syncoffset != 0 || t->params_int[PARAM_INT_FRAMERATE_PLL]) nchunks = 1;
        else if (nchunks > 40) nchunks = 40;
        const int64_t count = tsdrgpu_resample_count(e->rs, (uint32_t)chunk, nchunks, up, down);
        if (count < 0 || !ensure_dev(e, &e->d_rs, &e->rs_cap, (size_t)count + 16)) return;
        int64_t n = 0;
        if (!gpu_ok(e, tsdrgpu_resample(e->rs, e->iq.d + e->iq.rd, !e->iq_is_mag, (uint32_t)chunk, nchunks, up, down,
                                        (int)t->params_int[PARAM_NEAREST_NEIGHBOUR_RESAMPLING], e->d_rs, (int64_t)e->rs_cap, &n),
                    "resample"))
            return;
        e->iq.rd += (size_t)nchunks * chunk * per;
        /* dsp_dropped_compensation_add with a ring that always accepts (dsp.c:326-346) */
        if ((int64_t)n <= e->pix_difference) e->pix_difference -= n;
        else {
            if (!stream_append(e, &e->pix, e->d_rs + e->pix_difference, (size_t)(n - e->pix_difference))) return;
            e->pix_difference = 0;
        }
        /* manual sync, TSDRLibrary.c:345-346 */
        const int so = t->syncoffset;
        t->syncoffset = 0;
        e->pix_difference = drop_shift_with(e->pix_difference, (uint32_t)totalpixels, -(int64_t)so);
        run_frames(e);
    }
}

static void super_reset(struct engine *e)
{
    e->super_state = SUPER_STOPPED;
}

/* superb_run (superbandwidth.c:179-254) on the device thread.  Returns 1 when a
 * stitched buffer of *out_samples complex samples is ready in e->d_super_out. */
static int super_feed(struct engine *e, const float *d_blk, size_t nfloats, int64_t dropped, uint32_t *out_samples)
{
    tsdr_lib_t *t = e->t;
    if (e->super_state == SUPER_STOPPED) e->super_state = SUPER_STARTING;
    if (e->super_state == SUPER_STARTING) {
        e->super_hop = 0;
        e->super_gathered = 0;
        if (t->samplerate_real != e->super_rate || !e->d_hops[0]) {
            e->super_rate = t->samplerate_real;
            e->super_frame = (int)(t->samplerate_real / t->refreshrate);
            e->super_to_gather = SUPER_FRAMES_TO_RECORD * e->super_frame;
            e->super_to_pause = (int)(SUPER_SECS_TO_PAUSE * t->samplerate_real);
            tsdrgpu_sync(e->g);
            for (int i = 0; i < SUPER_HOPS; i++) {
                tsdrgpu_free(e->g, e->d_hops[i]);
                e->d_hops[i] = NULL;
                if (tsdrgpu_alloc(e->g, (void **)&e->d_hops[i], sizeof(float) * 2 * (size_t)e->super_to_gather)) return 0;
            }
        }
        e->super_state = SUPER_GATHERING;
    }
    if (e->super_state == SUPER_PAUSE) {
        e->super_gathered += (int)(nfloats / 2);
        if (e->super_gathered > e->super_to_pause) {
            e->super_gathered = 0;
            e->super_state = SUPER_GATHERING;
        }
    }
    if (e->super_state == SUPER_GATHERING) {
        if (dropped) { e->super_gathered = 0; return 0; }
        const int now = (int)(nfloats / 2);
        if (e->super_gathered + now < e->super_to_gather) {
            tsdrgpu_copy(e->g, e->d_hops[e->super_hop] + 2 * (size_t)e->super_gathered, d_blk, nfloats * sizeof(float));
            e->super_gathered += now;
        } else {
            const int remain = e->super_to_gather - e->super_gathered;
            tsdrgpu_copy(e->g, e->d_hops[e->super_hop] + 2 * (size_t)e->super_gathered, d_blk, (size_t)remain * 2 * sizeof(float));
            e->super_gathered += remain;
            e->super_hop++;
            const int gathered = e->super_gathered;
            e->super_gathered = 0;
            if (e->super_hop >= SUPER_HOPS) {
                /* superb_ondataready, superbandwidth.c:121-152 (the reference does this on super_thread) */
                uint32_t per = 1;
                while (per * 2 <= (uint32_t)gathered) per *= 2;
                const size_t need = (size_t)SUPER_HOPS * per * 2;
                if (!ensure_dev(e, &e->d_super_out, &e->super_out_cap, need)) return 0;
                int32_t offs[SUPER_HOPS];
                uint32_t total = 0;
                e->super_state = SUPER_STARTING;
                if (exact_wanted("TSDR_GPU_EXACT_AUTOCORR") /* covers the stitch too */
                        ? tsdrgpu_superb_stitch_exact(e->g, e->d_hops, SUPER_HOPS, gathered, e->super_frame, e->d_super_out, offs, &total)
                        : tsdrgpu_superb_stitch(e->g, e->d_hops, SUPER_HOPS, gathered, e->super_frame, e->d_super_out, offs, &total))
                    return 0;
                pthread_mutex_lock(&t->lock);
                tsdr_geometry_update(t, SUPER_HOPS * e->super_rate); /* superbandwidth.c:151 */
                pthread_mutex_unlock(&t->lock);
                *out_samples = total;
                return 1;
            }
            /* retune for the next hop, superbandwidth.c:241 */
            if (t->plugin.loaded) t->plugin.setbasefreq(t->centfreq + (uint32_t)((e->super_hop - SUPER_HOPS / 2) * (int64_t)e->super_rate));
            e->super_state = SUPER_PAUSE;
        }
    }
    return 0;
}

static void process_block(struct engine *e, const float *h, size_t nfloats, int64_t dropped)
{
    tsdr_lib_t *t = e->t;
    if (nfloats) {
        if (!ensure_dev(e, &e->d_block, &e->block_cap, nfloats)) return;
        if (!gpu_ok(e, tsdrgpu_upload(e->g, e->d_block, h, nfloats * sizeof(float)), "upload")) return;
    }
    const size_t size2 = nfloats / 2;

    if (t->params_int[PARAM_AUTOCORR_SUPERRESOLUTION]) { /* TSDRLibrary.c:271-279 */
        if (!e->iq_is_mag) { e->iq.rd = e->iq.wr = 0; e->det.rd = e->det.wr = 0; e->iq_is_mag = 1; }
        uint32_t total = 0;
        if (nfloats && super_feed(e, e->d_block, nfloats, dropped, &total)) {
            /* am_demod of the stitched buffer, then on to the resampler at 4x the rate */
            if (!stream_reserve(e, &e->iq, total)) return;
            if (!gpu_ok(e, tsdrgpu_am_demod(e->g, e->d_super_out, e->iq.d + e->iq.wr, total), "am_demod")) return;
            e->iq.wr += total;
        }
    } else {
        if (e->iq_is_mag || e->super_state != SUPER_STOPPED) { /* superb_stop, superbandwidth.c:256-264 */
            super_reset(e);
            if (t->plugin.loaded) t->plugin.setbasefreq(t->centfreq);
            pthread_mutex_lock(&t->lock);
            tsdr_geometry_update(t, t->samplerate_real);
            pthread_mutex_unlock(&t->lock);
            e->iq.rd = e->iq.wr = 0; e->det.rd = e->det.wr = 0; e->iq_is_mag = 0;
        }
        pthread_mutex_lock(&t->lock);
        const int block = (int)round(((t->width * t->height) << 1) * t->pixeltimeoversampletime); /* TSDRLibrary.c:284 */
        pthread_mutex_unlock(&t->lock);
        e->dev_difference = drop_shift_with(e->dev_difference, (uint32_t)block, dropped);
        const int drop_all = (int64_t)size2 <= e->dev_difference;
        const int plots_on = !t->params_int[PARAM_AUTOCORR_PLOTS_OFF];
        /* frameratedetector_run, frameratedetector.c:215-230 */
        if (plots_on) {
            if (dropped != 0) e->det.rd = e->det.wr = 0;
            else if (!drop_all && nfloats) stream_append(e, &e->det, e->d_block, nfloats);
        }
        /* dsp_dropped_compensation_add, dsp.c:326-346 */
        if (drop_all) e->dev_difference -= (int64_t)size2;
        else {
            stream_append(e, &e->iq, e->d_block + 2 * e->dev_difference, nfloats - 2 * (size_t)e->dev_difference);
            e->dev_difference = 0;
        }
    }
    if (nfloats) tsdrgpu_sync(e->g); /* the pinned slot is recycled as soon as we return */
    run_resampler(e);
    run_detector(e, t->samplerate);
}

static void *device_thread(void *arg)
{
    struct engine *e = (struct engine *)arg;
    tsdr_lib_t *t = e->t;
    while (t->running) {
        pthread_mutex_lock(&e->qm);
        if (!e->q_count) {
            struct timespec ts;
            clock_gettime(CLOCK_REALTIME, &ts);
            ts.tv_nsec += 30 * 1000000L;
            if (ts.tv_nsec >= 1000000000L) { ts.tv_sec++; ts.tv_nsec -= 1000000000L; }
            pthread_cond_timedwait(&e->q_nonempty, &e->qm, &ts);
            pthread_mutex_unlock(&e->qm);
            continue;
        }
        const int s = e->q_head;
        pthread_mutex_unlock(&e->qm);
        process_block(e, e->slot[s].h, e->slot[s].nfloats, e->slot[s].dropped);
        pthread_mutex_lock(&e->qm);
        e->q_head = (e->q_head + 1) % NSLOT;
        e->q_count--;
        pthread_mutex_unlock(&e->qm);
    }
    return NULL;
}

/* ---- entry ----------------------------------------------------------------------------- */
int engine_run(tsdr_lib_t *t, tsdr_readasync_function cb, void *ctx)
{
    struct engine *e = (struct engine *)calloc(1, sizeof(*e));
    if (!e) return tsdr_set_error(t, TSDR_ERR_PLUGIN, "out of memory");
    e->t = t;
    e->cb = cb;
    e->cbctx = ctx;
    int dev = 0;
    const char *env = getenv("TSDR_GPU_DEVICE");
    if (env) dev = atoi(env);
    if (tsdrgpu_create(&e->g, dev) || tsdrgpu_resampler_create(e->g, &e->rs) || tsdrgpu_postproc_create(e->g, &e->pp)) {
        if (e->g) tsdrgpu_destroy(e->g);
        free(e);
        return tsdr_set_error(t, TSDR_CANNOT_OPEN_DEVICE, "No usable MI355X/HIP device: this library has no CPU path.");
    }
    /* Sync-detector decisions that are toss-ups at the precision of the collapsed strips are detected and redone
     * with the reference's own strip arithmetic (tsdrgpu_postproc_set_exact_ties; on by default in the library as
     * well).  TSDR_GPU_EXACT_SYNC=0 / TSDR_GPU_EXACT=0 opt out. */
    (void)tsdrgpu_postproc_set_exact_ties(e->pp, exact_wanted("TSDR_GPU_EXACT_SYNC"));
    pthread_mutex_init(&e->qm, NULL); pthread_cond_init(&e->q_nonempty, NULL);
    pthread_mutex_init(&e->fm, NULL); pthread_cond_init(&e->f_nonempty, NULL);
    pthread_mutex_init(&e->pm, NULL); pthread_cond_init(&e->p_nonempty, NULL);
    e->alive = 1;
    t->eng = e;
    /* frameratedetector_startthread flushes the cached estimation, frameratedetector.c:203-209 */
    t->detector_purge = 1;
    t->params_int[PARAM_AUTOCORR_PLOTS_RESET] = 2;

    pthread_t th_dev, th_video, th_plot;
    pthread_create(&th_dev, NULL, device_thread, e);
    pthread_create(&th_video, NULL, video_thread, e);
    pthread_create(&th_plot, NULL, plot_thread, e);

    const int status = t->plugin.readasync(on_block, e); /* blocks until tsdr_stop / plugin failure */

    t->running = 0;
    pthread_join(th_dev, NULL);
    e->alive = 0;
    pthread_mutex_lock(&e->fm); pthread_cond_broadcast(&e->f_nonempty); pthread_mutex_unlock(&e->fm);
    pthread_mutex_lock(&e->pm); pthread_cond_broadcast(&e->p_nonempty); pthread_mutex_unlock(&e->pm);
    pthread_join(th_video, NULL);
    pthread_join(th_plot, NULL);

    tsdrgpu_sync(e->g);
    if (e->super_state != SUPER_STOPPED && t->plugin.loaded) t->plugin.setbasefreq(t->centfreq);
    for (int i = 0; i < NSLOT; i++) tsdrgpu_free_host(e->g, e->slot[i].h);
    for (int i = 0; i < NFRAMEQ; i++) tsdrgpu_free_host(e->g, e->fq[i].h);
    for (int i = 0; i < SUPER_HOPS; i++) tsdrgpu_free(e->g, e->d_hops[i]);
    tsdrgpu_free_host(e->g, e->h_frameplot);
    tsdrgpu_free_host(e->g, e->h_lineplot);
    tsdrgpu_free(e->g, e->d_block);
    tsdrgpu_free(e->g, e->d_rs);
    tsdrgpu_free(e->g, e->d_out);
    tsdrgpu_free(e->g, e->d_super_out);
    stream_free(e, &e->iq); stream_free(e, &e->det); stream_free(e, &e->pix);
    if (e->ac) tsdrgpu_autocorr_destroy(e->ac);
    tsdrgpu_postproc_destroy(e->pp);
    tsdrgpu_resampler_destroy(e->rs);
    tsdrgpu_destroy(e->g);
    pthread_mutex_destroy(&e->qm); pthread_cond_destroy(&e->q_nonempty);
    pthread_mutex_destroy(&e->fm); pthread_cond_destroy(&e->f_nonempty);
    pthread_mutex_destroy(&e->pm); pthread_cond_destroy(&e->p_nonempty);
    t->eng = NULL;
    free(e);
    if (status != TSDR_OK) return tsdr_set_error(t, status, t->plugin.getlasterrortext());
    t->errormsg_code = TSDR_OK;
    return TSDR_OK;
}
