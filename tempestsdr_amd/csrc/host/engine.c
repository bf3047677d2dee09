/* engine.c — the streaming pipeline behind tsdr_readasync.
 *
 * The reference moves data plugin-thread -> decimatingthread ->
 * postprocessingthread -> videodecodingthread through three float ring buffers,
 * with frameratedetector_thread and super_thread on the side
 * (TempestSDR/src/TSDRLibrary.c:264-418, frameratedetector.c:128-187,
 * superbandwidth.c:154-254).  Here four host threads feed five device queues
 * ("lanes", include/tsdrgpu.h) that only events order, so PCIe copies in both
 * directions overlap the kernels:
 *
 *   plugin thread   on_block(): DMA of the IQ block into a device slot on the UPLOAD
 *                   lane — straight out of the plugin's own buffer when that can be
 *                   page-locked, through a pinned bounce buffer otherwise —, then a
 *                   descriptor into the input queue
 *   device thread   queues, without ever waiting for the device in steady state:
 *                   slot -> sample streams, fused demod+resample straight into the pixel
 *                   stream, batched frame post-processing into a ring of output buffers
 *                   (COMPUTE lane), the frame-rate detector (BACKGROUND lane), the frames' and
 *                   plots' way back to pinned memory (DOWNLOAD / BACKGROUND lane)
 *   video thread    waits for a frame's download event, then the frame callback
 *                   (reference: videodecodingthread)
 *   plot thread     waits for the plots' event, then plot + value callbacks
 *                   (reference: frameratedetector_thread)
 *
 * Sample / pixel skipping after drops follows dsp_dropped_compensation_*
 * (dsp.c:313-368) so frames stay aligned, and back-pressure is lossy in whole
 * blocks / whole frames like the reference's circular buffers.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "tsdr_host.h"

#define NSLOT 64           /* input blocks in flight: when the host is slow the backlog turns into bigger batches */
#define NFRAMEQ 48         /* frames on their way to / waiting for the video callback */
#define NOUT 6             /* post-processed batches whose frames may still be downloading */
#define MAX_FRAME_BATCH 32
#define FUSE_MIN_FRAMES 8  /* a backlog of at least this many frames takes the fused run (tsdrgpu_postproc_begin_minmax) */
#define MAX_CHUNKS_PER_CALL 120 /* resampler chunks per call (0.1 frame each): a backlog of 12 frames in one launch group */
#define DET_REPLAY_STEP 16 /* windows of an exact replay per device-thread turn (2 ms at 2^22 samples): the frame path keeps flowing */
#define MM_CAP 4096        /* per-frame min/max values kept for frames that wait in the pixel stream */
#define MAX_HOSTREG 512    /* page-locked ranges of plugin memory */
#define MAX_HOSTREJ 16     /* ranges that could not be page-locked, remembered so that the call is not repeated */
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
    float *h;     /* pinned frame */
    size_t cap;
    int width, height;
    tsdrgpu_pp_frameinfo_t *h_info; /* pinned; the frame's post-processing record */
    int announce_autogain;
    tsdrgpu_event_t *ready; /* recorded on the DOWNLOAD lane behind the frame's copies */
    /* what the download thread needs: where the frame and its record live on the device, and the batch they belong to */
    const void *d_src;
    const tsdrgpu_pp_frameinfo_t *d_info_src;
    struct out_buf *ob;
} frame_slot_t;

typedef struct out_buf {
    float *d; size_t cap;              /* F frames */
    int32_t *d_rgb; size_t rgb_cap;    /* the same frames as packed RGB (tsdrx_readasync_rgb runs only) */
    tsdrgpu_pp_frameinfo_t *d_info;    /* F records */
    int info_cap;
    tsdrgpu_event_t *done, *last_dl;   /* batch computed (COMPUTE) / its latest download (DOWNLOAD) */
    int busy;
    int pending;                       /* frames of the batch whose download is not queued yet (under fm) */
} out_buf_t;

typedef struct {
    float *h; size_t hcap;   /* pinned bounce buffer (used when the plugin's memory cannot be page-locked) */
    float *d; size_t dcap;   /* device copy of the block */
    void *d_raw; size_t raw_cap; /* the block in the plugin's native sample format (tsdrplugin_readasync_raw), bytes */
    int raw_type;            /* TSDRX_SAMPLE_*: != FLOAT32 means d_raw still has to be decoded into d */
    size_t nfloats;
    int64_t dropped;
    tsdrgpu_event_t *consumed; /* COMPUTE lane is done reading d */
    int consumed_valid;
    tsdrgpu_event_t *uploaded; /* UPLOAD lane has written d (or d_raw): the device thread waits for it ON THE HOST */
    int uploaded_valid;
} in_slot_t;

typedef struct {
    int32_t flo, flen, llo, llen;
    uint32_t rate;
    double *h_frame, *h_line; /* pinned */
    const double *d_snapshot; /* the plots on the device (frame first), complete when plot_ready has fired */
    uint64_t calls;
    int certify;              /* an argmax + certificate was queued in front of the snapshot on `ac` */
    tsdrgpu_autocorr_t *ac;
} plot_msg_t;

struct engine {
    tsdr_lib_t *t;
    tsdr_readasync_function cb;
    void *cbctx;

    tsdrgpu_t *g;
    tsdrgpu_resampler_t *rs;
    tsdrgpu_postproc_t *pp;
    tsdrgpu_autocorr_t *ac;
    uint32_t ac_rate, ac_failed_rate;
    uint32_t ac_capture;
    tsdrgpu_event_t *det_read; /* the detector's lane (when it has its own) is done reading the detector's sample stream */
    int det_read_valid;
    int ac_certified;          /* the detector runs in its certified mode (default): plots leave only with a certificate */
    int det_promote;  /* (A_LD / A_ST) plot thread -> device thread: the last plot's argmax was not certified, replay the epoch exactly */
    int det_replaying;         /* an epoch is being replayed in the reference's arithmetic, DET_REPLAY_STEP windows per turn */
    long n_promotions, n_plots_held;

    /* input queue */
    in_slot_t slot[NSLOT];
    int q_head, q_count;
    int64_t pending_drop;
    pthread_mutex_t qm;
    pthread_cond_t q_nonempty;
    int plugin_thread_bound;
    in_slot_t *prev_upload; /* plugin thread: the slot whose DMA was queued by the previous callback */
    int async_upload;       /* TSDR_GPU_ASYNC_UPLOAD=1 */
    int zero_copy;
    int immutable;   /* the plugin promises that a block's CONTENTS stay as they are while tsdrplugin_readasync runs: its DMA may
                        still be in flight when the callback returns (tsdrplugin_memory_stable, bits 1 / 2) */
    /* the second half of a bounce-buffer copy runs on a helper thread while the plugin's thread copies the first */
    pthread_t th_copy;
    int copy_thread_on;
    pthread_mutex_t cm;
    pthread_cond_t c_wake;
    int copy_state; /* 0 idle, 1 job posted, 2 done (atomic) */
    int copy_quit;  /* (atomic) */
    void *copy_dst; const void *copy_src; size_t copy_n;
    /* per-frame min/max out of the resampler (frame tracking) for the frames that wait in the pixel stream: what the
     * fused run needs instead of a statistics read of its own */
    float *d_mm_min, *d_mm_max;
    int mm_off, mm_n;        /* valid entries [mm_off, mm_off + mm_n): one per whole frame behind the `mm_nohead` first ones */
    int mm_nohead;           /* whole frames at the head of the pixel stream that have no min/max (tracking started later) */
    int mm_drop_first;       /* the tracker's next completed frame began before tracking did: its min/max is partial */
    int64_t track_P;         /* frame size the resampler is tracking (0: off) */
    long n_fused_batches, n_fused_frames;
    struct { char *p; size_t n; } reg[MAX_HOSTREG];
    int nreg;
    struct { char *p; size_t n; } rej[MAX_HOSTREJ]; /* ranges hipHostRegister refused */
    int nrej;

    devstream_t iq;   /* samples for the resampler (interleaved IQ; magnitude in super mode) */
    int iq_is_mag;
    devstream_t det;  /* samples for the frame-rate detector */
    devstream_t pix;  /* resampled pixel stream */
    float *d_rs; size_t rs_cap; /* resampler scratch for the calls whose first pixels are skipped */
    out_buf_t out[NOUT];
    int out_next;
    int32_t *d_rgb_state; size_t rgb_state_cap; /* the viewer's pixel buffer: transparent pixels keep the previous frame's colour */

    int64_t dev_difference; /* samples still to skip (process(), TSDRLibrary.c:284-295) */
    int64_t pix_difference; /* pixels still to skip (decimatingthread, TSDRLibrary.c:342-346) */
    int pp_runs;

    /* video delivery */
    frame_slot_t fq[NFRAMEQ];
    int fq_head, fq_count, fq_issued;  /* fq_issued of the fq_count queued frames have their download on the DOWNLOAD lane */
    pthread_mutex_t fm;
    pthread_cond_t f_nonempty;         /* a download was queued (video thread) */
    pthread_cond_t f_queued;           /* a frame was queued (download thread) / a batch has no download pending (device thread) */

    /* plot delivery: one message in flight; its geometry travels with it (the detector may be rebuilt for a
     * new sample rate while the host still looks at the previous plots) */
    plot_msg_t plot;
    tsdrgpu_event_t *plot_ready, *plot_home; /* snapshot taken (detector's lane) / copied to the host (DOWNLOAD lane) */
    int plot_pending, plot_reset_announce, plot_dumped_announce;
    pthread_mutex_t pm;
    pthread_cond_t p_nonempty;

    int alive;  /* delivery threads keep going (A_LD / A_ST) */
    int failed; /* (atomic) a device call failed: the session is being torn down (gpu_ok) */
    char fail_msg[400];

    /* TSDR_GPU_STATS=1: where the host threads spend their time (printed to stderr when the run ends) */
    int stats;
    double t_start;
    double s_plugin_busy, s_plugin_dma, s_dev_busy, s_dev_wait_out, s_video_wait, s_video_cb;
    double s_dev_blocks, s_dev_rs, s_dev_frames, s_dev_det; /* device thread: appending blocks / resampler / frame path / detector */
    long n_blocks, n_blocks_lost, n_frames_made, n_frames_lost, n_batches, n_resample_calls, n_windows;

    /* blocks of the current look at the queue that go to both sample streams with one launch (gather_flush) */
    const void *g_src[32];
    size_t g_bytes[32];
    int gn;
    float *g_det, *g_iq;

    /* super-bandwidth */
    int super_state, super_hop, super_gathered, super_to_gather, super_frame, super_to_pause;
    uint32_t super_rate;
    float *d_hops[SUPER_HOPS];
    float *d_super_out; size_t super_out_cap;
};

/* ---- small helpers ----------------------------------------------------------- */
/* the counters tsdrx_get_stats may read while their one writer counts (a snapshot, not a barrier): relaxed atomics */
#define STAT_ADD(x, n) ((void)__atomic_fetch_add(&(x), (n), __ATOMIC_RELAXED))
#define STAT_GET(x) __atomic_load_n(&(x), __ATOMIC_RELAXED)
/* Contract-exact modes are the default; `name`=0 (or TSDR_GPU_EXACT=0 for all of them) switches one off. */
static int exact_wanted(const char *name)
{
    const char *one = getenv(name), *all = getenv("TSDR_GPU_EXACT");
    if (one && one[0]) return one[0] != '0';
    if (all && all[0]) return all[0] != '0';
    return 1;
}

/* A failing device call ends the session: the first failure is remembered, the workers stop, the plugin is told to
 * stop, and tsdr_readasync returns TSDR_CANNOT_OPEN_DEVICE with this text (the reference surfaces its failures the
 * same way, through the return value of tsdr_readasync and tsdr_getlasterrortext).  Nothing is retried. */
static __thread int in_plugin_callback; /* this thread is inside on_block_any */
static int gpu_ok(struct engine *e, int rc, const char *what)
{
    if (rc == 0) return 1;
    if (!__atomic_exchange_n(&e->failed, 1, __ATOMIC_ACQ_REL)) {
        snprintf(e->fail_msg, sizeof(e->fail_msg), "GPU stage '%s' failed (%d): %s", what, rc, tsdrgpu_last_error(e->g));
        fprintf(stderr, "tsdr: %s\n", e->fail_msg);
        A_ST(e->t->running, 0);
        /* On the plugin's own thread — inside its readasync callback — the plugin is NOT told to stop from here: a plugin
         * whose tsdrplugin_stop waits for its streaming thread would wait for itself.  The device thread sees running == 0
         * within its 30 ms poll and makes the call (device_thread's last lines).  From any other thread the call is made at
         * once (tsdr_stop on the host's thread may be doing the same right now: tsdr_plugin_stop_once). */
        if (!in_plugin_callback) (void)tsdr_plugin_stop_once(e->t);
    }
    return 0;
}

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

static void deadline_ms(struct timespec *ts, long ms)
{
    clock_gettime(CLOCK_REALTIME, ts);
    ts->tv_nsec += ms * 1000000L;
    if (ts->tv_nsec >= 1000000000L) { ts->tv_sec++; ts->tv_nsec -= 1000000000L; }
}

static int stream_reserve(struct engine *e, devstream_t *s, size_t extra)
{
    if (s->wr + extra <= s->cap) return 1;
    const size_t live = s->wr - s->rd;
    size_t need = live + extra;
    if (need <= s->cap && s->d_alt) { /* compact into the twin buffer (COMPUTE lane: ordered behind every reader) */
        if (live && !gpu_ok(e, tsdrgpu_copy(e->g, s->d_alt, s->d + s->rd, live * sizeof(float)), "compact")) return 0;
        float *tmp = s->d; s->d = s->d_alt; s->d_alt = tmp;
        s->rd = 0; s->wr = live;
        return 1;
    }
    /* generous: a stream is compacted (its live part copied to the twin) every time the write position reaches the
     * end, so the slack behind the live part sets how often that copy happens (HBM is not the scarce resource) */
    size_t cap = need * 4 + ((size_t)16 << 20);
    float *n1 = NULL, *n2 = NULL;
    if (!gpu_ok(e, tsdrgpu_alloc(e->g, (void **)&n1, cap * sizeof(float)), "stream buffer") ||
        !gpu_ok(e, tsdrgpu_alloc(e->g, (void **)&n2, cap * sizeof(float)), "stream buffer")) {
        tsdrgpu_free(e->g, n1);
        return 0;
    }
    if (live && !gpu_ok(e, tsdrgpu_copy(e->g, n1, s->d + s->rd, live * sizeof(float)), "grow")) {
        tsdrgpu_free(e->g, n1);
        tsdrgpu_free(e->g, n2);
        return 0;
    }
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
/* Is [p, p+n) page-locked for DMA?  Only asked for plugins whose memory is stable (tsdrplugin_memory_stable,
 * include/TSDRLibraryExt.h: a range, once seen, stays allocated and mapped until after tsdrplugin_readasync has
 * returned — which is when the ranges are unlocked again).  Such plugins hand over blocks of one region again and
 * again, so ranges are registered on first sight and remembered; what cannot be registered is remembered too (the
 * failing call is not repeated for every block), and anything that overlaps a known range only partly goes through
 * the slot's pinned bounce buffer, like every block of a plugin that made no promise. */
static int plugin_memory_pinned(struct engine *e, void *p, size_t n)
{
    if (!e->zero_copy) return 0;
    char *c = (char *)p;
    for (int i = 0; i < e->nreg; i++) {
        if (c >= e->reg[i].p && c + n <= e->reg[i].p + e->reg[i].n) return 1;
        if (c < e->reg[i].p + e->reg[i].n && e->reg[i].p < c + n) return 0; /* partial overlap */
    }
    for (int i = 0; i < e->nrej; i++)
        if (c < e->rej[i].p + e->rej[i].n && e->rej[i].p < c + n) return 0; /* refused before */
    if (e->nreg == MAX_HOSTREG) return 0;
    if (tsdrgpu_host_register(e->g, p, n)) {
        if (e->nreg == 0 || e->nrej == MAX_HOSTREJ) e->zero_copy = 0; /* this plugin's memory cannot be page-locked: stop trying */
        else { e->rej[e->nrej].p = c; e->rej[e->nrej].n = n; e->nrej++; }
        return 0;
    }
    e->reg[e->nreg].p = c;
    e->reg[e->nreg].n = n;
    e->nreg++;
    return 1;
}

static size_t sample_bytes(int type)
{
    return (type == TSDRX_SAMPLE_INT8 || type == TSDRX_SAMPLE_UINT8) ? 1 : ((type == TSDRX_SAMPLE_INT16 || type == TSDRX_SAMPLE_UINT16) ? 2 : 4);
}

/* ---- bounce-buffer copy in two halves ------------------------------------------------ */
/* A plugin that made no promise about its memory (every reference plugin) is served through pinned buffers of ours, and
 * the copy into them has to be complete when the callback returns.  2 MiB at memcpy speed is ~60-200 us on the plugin's
 * own thread — more than the DMA that follows it — so a helper takes the second half.  The helper spins for a short while
 * after a job (blocks of a free-running source follow each other within ~100 us) and sleeps on a condition variable
 * otherwise (a live source's blocks are milliseconds apart: the wake-up latency does not matter there). */
#if defined(__x86_64__) || defined(__i386__)
#define cpu_relax() __builtin_ia32_pause()
#elif defined(__aarch64__) || defined(__arm__)
#define cpu_relax() __asm__ __volatile__("yield" ::: "memory")
#else
#define cpu_relax() sched_yield()
#endif
/* copy_state: 0 idle, 1 job posted (copy_dst / copy_src / copy_n valid), 2 done — handed over with acquire / release */
static void *copy_thread(void *arg)
{
    struct engine *e = (struct engine *)arg;
    double last = now_s();
    for (;;) {
        if (__atomic_load_n(&e->copy_state, __ATOMIC_ACQUIRE) == 1) {
            memcpy(e->copy_dst, e->copy_src, e->copy_n);
            __atomic_store_n(&e->copy_state, 2, __ATOMIC_RELEASE);
            last = now_s();
            continue;
        }
        if (A_LD(e->copy_quit)) break;
        if (now_s() - last < 300e-6) { cpu_relax(); continue; }
        pthread_mutex_lock(&e->cm);
        while (A_LD(e->copy_state) != 1 && !A_LD(e->copy_quit)) pthread_cond_wait(&e->c_wake, &e->cm);
        pthread_mutex_unlock(&e->cm);
    }
    return NULL;
}

static void bounce_copy(struct engine *e, void *dst, const void *src, size_t n)
{
    if (!e->copy_thread_on || n < ((size_t)256 << 10)) { memcpy(dst, src, n); return; }
    const size_t half = (n / 2) & ~(size_t)63;
    e->copy_dst = (char *)dst + half;
    e->copy_src = (const char *)src + half;
    e->copy_n = n - half;
    /* posted under the helper's mutex: a flag read outside it ("is the helper asleep?") can miss a helper that is just
     * going to sleep (store-load reordering on both sides) — that was a deadlock */
    pthread_mutex_lock(&e->cm);
    __atomic_store_n(&e->copy_state, 1, __ATOMIC_RELEASE);
    pthread_cond_signal(&e->c_wake);
    pthread_mutex_unlock(&e->cm);
    memcpy(dst, src, half);
    while (__atomic_load_n(&e->copy_state, __ATOMIC_ACQUIRE) != 2) cpu_relax();
    __atomic_store_n(&e->copy_state, 0, __ATOMIC_RELAXED);
}

static void on_block_inner(const void *buf, uint64_t items, int type, struct engine *e, int64_t dropped);
static void on_block_any(const void *buf, uint64_t items, int type, void *ctx, int64_t dropped)
{
    struct engine *e = (struct engine *)ctx;
    if (!A_LD(e->t->running) || (items & 1)) return;
    if (!e->plugin_thread_bound) { tsdrgpu_bind_thread(e->g); e->plugin_thread_bound = 1; }
    in_plugin_callback = 1;
    on_block_inner(buf, items, type, e, dropped);
    in_plugin_callback = 0;
}

static void on_block_inner(const void *buf, uint64_t items, int type, struct engine *e, int64_t dropped)
{
    const double t0 = e->stats ? now_s() : 0.0;
    /* device thread is behind: lose the whole block.  Decided without the queue's mutex — a free-running source whose DMAs we
     * no longer wait for comes here tens of millions of times per second, and a plugin thread spinning on the mutex starved
     * the device thread of it (measured: 95 % of the device thread's time went into getting it).  Only this thread raises
     * q_count, so a queue seen full here can only have become emptier; pending_drop and n_blocks_lost are this thread's own. */
    if (A_LD(e->q_count) == NSLOT) {
        e->pending_drop += (int64_t)(items / 2) + dropped;
        STAT_ADD(e->n_blocks_lost, 1);
        return;
    }
    pthread_mutex_lock(&e->qm);
    in_slot_t *s = &e->slot[(e->q_head + A_LD(e->q_count)) % NSLOT]; /* only this thread produces: the slot stays ours */
    pthread_mutex_unlock(&e->qm);
    int ok = 1;
    if (items) {
        const size_t bytes = (size_t)items * sample_bytes(type);
        if (s->consumed_valid) { ok = gpu_ok(e, tsdrgpu_event_sync(e->g, s->consumed), "slot wait"); s->consumed_valid = 0; } /* long done in practice */
        if (ok && s->dcap < items) {
            tsdrgpu_free(e->g, s->d);
            s->d = NULL; s->dcap = 0;
            if (gpu_ok(e, tsdrgpu_alloc(e->g, (void **)&s->d, (size_t)items * sizeof(float)), "block buffer")) s->dcap = items; else ok = 0;
        }
        void *dst = s->d;
        if (type != TSDRX_SAMPLE_FLOAT32) { /* narrow samples cross PCIe as they are; the device thread decodes them */
            if (ok && s->raw_cap < bytes) {
                tsdrgpu_free(e->g, s->d_raw);
                s->d_raw = NULL; s->raw_cap = 0;
                if (gpu_ok(e, tsdrgpu_alloc(e->g, &s->d_raw, bytes), "raw block buffer")) s->raw_cap = bytes; else ok = 0;
            }
            dst = s->d_raw;
        }
        const void *src = buf;
        /* the previous DMA into this slot's device buffer / out of its bounce buffer (NSLOT blocks ago) is long complete */
        if (ok && s->uploaded_valid) { ok = gpu_ok(e, tsdrgpu_event_sync(e->g, s->uploaded), "slot upload wait"); s->uploaded_valid = 0; }
        int wait_dma = 0; /* must the DMA be complete when we return? */
        if (ok && !plugin_memory_pinned(e, (void *)buf, bytes)) {
            if (s->hcap < items) { /* sized for float32, the widest format */
                tsdrgpu_free_host(e->g, s->h);
                s->h = NULL; s->hcap = 0;
                if (gpu_ok(e, tsdrgpu_alloc_host(e->g, (void **)&s->h, (size_t)items * sizeof(float)), "pinned bounce buffer")) s->hcap = items; else ok = 0;
            }
            /* the copy is what has to be complete on return; the DMA out of OUR buffer could run behind the callback's back
             * (TSDR_GPU_ASYNC_UPLOAD=1) but is waited for by default, see engine_run */
            if (ok) { bounce_copy(e, s->h, buf, bytes); src = s->h; wait_dma = !e->async_upload; }
        } else if (ok) {
            /* straight out of the plugin's memory: the block is the plugin's again when we return — unless it promised that
             * its contents do not change while it streams (a recording): then this DMA and the next ones overlap */
            wait_dma = !e->immutable;
        }
        const double t1 = e->stats ? now_s() : 0.0;
        /* a failing device call ends the session here as well (gpu_ok; the error text is this thread's own) */
        if (ok) ok = gpu_ok(e, tsdrgpu_upload_lane(e->g, dst, src, bytes), "upload");
        if (ok && wait_dma) ok = gpu_ok(e, tsdrgpu_lane_sync(e->g, TSDRGPU_LANE_UPLOAD), "upload wait");
        else if (ok) {
            /* TWO in flight: this block's DMA is queued, then the PREVIOUS block's is waited for — the copy engine always has
             * the next transfer at hand (one at a time, each waited for, left it idle for ~5-8 us per 45 us block), and the
             * plugin's thread still advances at the link's pace.  (Queueing without ever waiting was measured and is worse:
             * 25 GB/s through the lane against 35-41 with waits, and a free-running source then calls in tens of millions of
             * times per second only to be turned away.) */
            ok = gpu_ok(e, tsdrgpu_event_record(e->g, s->uploaded, TSDRGPU_LANE_UPLOAD), "upload event");
            s->uploaded_valid = ok;
            if (ok && e->prev_upload && e->prev_upload != s && e->prev_upload->uploaded_valid)
                ok = gpu_ok(e, tsdrgpu_event_sync(e->g, e->prev_upload->uploaded), "upload wait");
            e->prev_upload = s;
        }
        if (e->stats) e->s_plugin_dma += now_s() - t1;
    }
    s->raw_type = type;
    if (!ok) { /* could not stage the block: count it as lost, like a full queue */
        e->pending_drop += (int64_t)(items / 2) + dropped;
        return;
    }
    pthread_mutex_lock(&e->qm);
    s->nfloats = items;
    s->dropped = dropped + e->pending_drop;
    e->pending_drop = 0;
    __atomic_fetch_add(&e->q_count, 1, __ATOMIC_RELEASE);
    STAT_ADD(e->n_blocks, 1);
    pthread_cond_signal(&e->q_nonempty);
    pthread_mutex_unlock(&e->qm);
    if (e->stats) e->s_plugin_busy += now_s() - t0;
}

static void on_block(float *buf, uint64_t items, void *ctx, int64_t dropped) /* the tsdrplugin_readasync callback */
{
    on_block_any(buf, items, TSDRX_SAMPLE_FLOAT32, ctx, dropped);
}

static void on_block_raw(const void *buf, uint64_t items, int type, void *ctx, int64_t dropped) /* tsdrplugin_readasync_raw */
{
    if (type < TSDRX_SAMPLE_FLOAT32 || type > TSDRX_SAMPLE_UINT16) return;
    on_block_any(buf, items, type, ctx, dropped);
}

/* ---- video thread ---------------------------------------------------------------- */
static void *video_thread(void *arg)
{
    struct engine *e = (struct engine *)arg;
    tsdr_lib_t *t = e->t;
    tsdrgpu_bind_thread(e->g);
    pthread_mutex_lock(&e->fm);
    while (A_LD(e->alive) || e->fq_count) {
        if (!e->fq_issued) {
            struct timespec ts;
            deadline_ms(&ts, 30);
            pthread_cond_timedwait(&e->f_nonempty, &e->fm, &ts);
            continue;
        }
        frame_slot_t *f = &e->fq[e->fq_head];
        pthread_mutex_unlock(&e->fm);
        const double t0 = e->stats ? now_s() : 0.0;
        const int arrived = tsdrgpu_event_sync(e->g, f->ready) == 0;
        const double t1 = e->stats ? now_s() : 0.0;
        if (arrived && A_LD(t->running)) {
            /* dsp.c:231-235: autogain values every 7th frame */
            if (f->announce_autogain) tsdr_announce_value(t, VALUE_ID_AUTOGAIN_VALUES, f->h_info->lastmin, f->h_info->lastmax);
            if (t->rgb_cb) t->rgb_cb((int32_t *)(void *)f->h, f->width, f->height, e->cbctx);
            else e->cb(f->h, f->width, f->height, e->cbctx);
        }
        if (e->stats) { e->s_video_wait += t1 - t0; e->s_video_cb += now_s() - t1; }
        pthread_mutex_lock(&e->fm);
        e->fq_head = (e->fq_head + 1) % NFRAMEQ;
        e->fq_count--;
        e->fq_issued--;
    }
    pthread_mutex_unlock(&e->fm);
    return NULL;
}

/* ---- download thread ---------------------------------------------------------------- */
/* Waits ON THE HOST for a batch to be computed and only then queues its frames' way home.  Letting the DOWNLOAD lane
 * wait on the device instead (hipStreamWaitEvent) parks a barrier packet in that lane's hardware queue until the batch
 * is done, and a parked barrier slows the COMPUTE lane's own queue down — measured on MI355X: every kernel of the
 * frame path took ~40 us longer and the engine ran at 1.8 instead of 5 GS/s, depending on which hardware queues the
 * lanes happened to get. */
static void *download_thread(void *arg)
{
    struct engine *e = (struct engine *)arg;
    tsdrgpu_bind_thread(e->g);
    pthread_mutex_lock(&e->fm);
    while (A_LD(e->alive) || e->fq_issued < e->fq_count) {
        if (e->fq_issued == e->fq_count) {
            struct timespec ts;
            deadline_ms(&ts, 30);
            pthread_cond_timedwait(&e->f_queued, &e->fm, &ts);
            continue;
        }
        frame_slot_t *s = &e->fq[(e->fq_head + e->fq_issued) % NFRAMEQ];
        pthread_mutex_unlock(&e->fm);
        out_buf_t *ob = s->ob;
        tsdrgpu_event_sync(e->g, ob->done);
        const size_t P = (size_t)s->width * s->height;
        (void)tsdrgpu_download_lane(e->g, s->h, s->d_src, P * sizeof(float));
        (void)tsdrgpu_download_lane(e->g, s->h_info, s->d_info_src, sizeof(tsdrgpu_pp_frameinfo_t));
        (void)tsdrgpu_event_record(e->g, s->ready, TSDRGPU_LANE_DOWNLOAD);
        (void)tsdrgpu_event_record(e->g, ob->last_dl, TSDRGPU_LANE_DOWNLOAD);
        pthread_mutex_lock(&e->fm);
        e->fq_issued++;
        ob->pending--;
        pthread_cond_signal(&e->f_nonempty);
        if (!ob->pending) pthread_cond_broadcast(&e->f_queued);
    }
    pthread_mutex_unlock(&e->fm);
    return NULL;
}

/* ---- plot thread ------------------------------------------------------------------- */
static void *plot_thread(void *arg)
{
    struct engine *e = (struct engine *)arg;
    tsdr_lib_t *t = e->t;
    tsdrgpu_bind_thread(e->g);
    pthread_mutex_lock(&e->pm);
    while (A_LD(e->alive)) {
        if (!e->plot_pending && !e->plot_reset_announce && !e->plot_dumped_announce) {
            struct timespec ts;
            deadline_ms(&ts, 30);
            pthread_cond_timedwait(&e->p_nonempty, &e->pm, &ts);
            continue;
        }
        const int reset = e->plot_reset_announce, dumped = e->plot_dumped_announce, plots = e->plot_pending;
        e->plot_reset_announce = e->plot_dumped_announce = 0;
        /* the message (arrays and geometry) stays untouched by the device thread while plot_pending is set */
        pthread_mutex_unlock(&e->pm);
        if (reset) tsdr_announce_value(t, VALUE_ID_AUTOCORRECT_RESET, 0, 0);
        if (dumped) tsdr_announce_value(t, VALUE_ID_AUTOCORRECT_DUMPED, 0, 0);
        /* the snapshot is complete once plot_ready has fired; its way home is queued here, on the DOWNLOAD lane, so
         * that the detector's lane never waits for a copy engine */
        int deliver = plots && tsdrgpu_event_sync(e->g, e->plot_ready) == 0;
        if (plots && e->plot.certify) {
            /* certified mode: the argmax kernels ran in front of the snapshot.  A plot whose argmax is not provably the
             * reference's does not leave: the device thread replays the epoch exactly and publishes that plot instead.
             * (The result is collected in any case, so that the next update can queue its own.) */
            int32_t fi, li;
            tsdrgpu_ac_certificate_t c;
            const int got = tsdrgpu_autocorr_argmax_result(e->plot.ac, &fi, &li) == 0 && tsdrgpu_autocorr_certificate(e->plot.ac, &c) == 0;
            if (!got) deliver = 0; /* no certificate, no delivery */
            else if (deliver && !(c.frame_certified && c.line_certified)) {
                deliver = 0;
                STAT_ADD(e->n_plots_held, 1);
                A_ST(e->det_promote, 1);
            }
        }
        if (deliver &&
            tsdrgpu_download_lane(e->g, e->plot.h_frame, e->plot.d_snapshot, sizeof(double) * (size_t)e->plot.flen) == 0 &&
            tsdrgpu_download_lane(e->g, e->plot.h_line, e->plot.d_snapshot + e->plot.flen, sizeof(double) * (size_t)e->plot.llen) == 0 &&
            tsdrgpu_event_record(e->g, e->plot_home, TSDRGPU_LANE_DOWNLOAD) == 0 && tsdrgpu_event_sync(e->g, e->plot_home) == 0) {
            const plot_msg_t *m = &e->plot;
            tsdr_on_plot_ready_callback pcb = A_LD(t->plotready_callback);
            if (pcb) { /* frameratedetector.c:121-124 */
                pcb(PLOT_ID_FRAME, m->flo, m->h_frame, m->flen, m->rate, t->callbackctx);
                pcb(PLOT_ID_LINE, m->llo, m->h_line, m->llen, m->rate, t->callbackctx);
            }
            tsdr_announce_value(t, VALUE_ID_AUTOCORRECT_FRAMES_COUNT, 0, (double)m->calls);
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
    /* the reference walks the first fft_getrealsize(2*capture)/2 = n floats of the correlation, i.e. n/2 rows */
    const uint32_t maxels = n;
    float *h = (float *)malloc(sizeof(float) * ((size_t)maxels + 2));
    if (!h) return;
    if (tsdrgpu_download(e->g, h, d_corr, sizeof(float) * maxels) == 0 && tsdrgpu_sync(e->g) == 0) {
        FILE *f = fopen("autocorr.csv", "w");
        if (f) {
            fprintf(f, "%s, %s\n", "ms", "dB");
            for (uint32_t i = 0; i + 1 < maxels; i += 2) {
                const double re = h[i], im = h[i + 1];
                fprintf(f, "%f, %f\n", 1000.0 * (i / 2) / (double)e->ac_rate, 10.0 * log10(sqrt(re * re + im * im)));
            }
            fclose(f);
        }
    }
    free(h);
}

static void detector_rebuild(struct engine *e, uint32_t fs)
{
    /* the plot thread may still be copying the old detector's snapshot home */
    pthread_mutex_lock(&e->pm);
    while (e->plot_pending && A_LD(e->alive)) {
        pthread_mutex_unlock(&e->pm);
        struct timespec ts = {0, 1000000};
        nanosleep(&ts, NULL);
        pthread_mutex_lock(&e->pm);
    }
    pthread_mutex_unlock(&e->pm);
    if (e->ac) tsdrgpu_autocorr_destroy(e->ac);
    e->ac = NULL;
    e->ac_rate = 0;
    if (tsdrgpu_autocorr_create(e->g, &e->ac, fs)) { /* rate too low for the lag windows, or no memory */
        e->ac = NULL;
        e->ac_failed_rate = fs;
        return;
    }
    /* Detector mode.  Default: CERTIFIED — the float32 three-trip transform; every plot that leaves carries an argmax
     * certificate (tsdrgpu_autocorr_set_certify), and an epoch whose certificate fails is replayed in the reference's own
     * FFT arithmetic before its plot is delivered, so the argmax the host takes (PlotVisualizer.java:233-236) is always
     * the CPU library's; such an epoch (and any epoch that outgrows the retention ring — TSDR_GPU_AUTOCORR_RETAIN_MB;
     * unset: a quarter of the free HBM, at most 32 GiB, allocated in segments in the background —) continues bit-identical.  TSDR_GPU_AUTOCORR=exact (or TSDR_GPU_EXACT_AUTOCORR=1 / TSDR_GPU_EXACT=1): every window in the
     * reference's arithmetic, plots bit-identical always.  TSDR_GPU_AUTOCORR=fast (or TSDR_GPU_EXACT_AUTOCORR=0 /
     * TSDR_GPU_EXACT=0): plain float32 form, plots within 1e-4*max, no guarantee on ties. */
    {
        const char *m = getenv("TSDR_GPU_AUTOCORR"), *one = getenv("TSDR_GPU_EXACT_AUTOCORR"), *all = getenv("TSDR_GPU_EXACT");
        int mode = 1; /* 0 fast, 1 certified, 2 exact */
        if (m && m[0]) mode = m[0] == 'e' ? 2 : (m[0] == 'f' ? 0 : 1);
        else if (one && one[0]) mode = one[0] != '0' ? 2 : 0;
        else if (all && all[0]) mode = all[0] != '0' ? 2 : 0;
        e->ac_certified = 0;
        if (mode == 2) (void)tsdrgpu_autocorr_set_exact(e->ac, 1);
        else if (mode == 1) {
            const char *mb = getenv("TSDR_GPU_AUTOCORR_RETAIN_MB");
            const size_t bytes = (mb && atol(mb) > 0) ? (size_t)atol(mb) << 20 : 0;
            if (tsdrgpu_autocorr_set_certify(e->ac, 1, bytes) == 0) e->ac_certified = 1;
            else (void)tsdrgpu_autocorr_set_exact(e->ac, 1); /* no room for the ring: the exact form needs none */
        }
    }
    A_ST(e->det_promote, 0);
    e->det_replaying = 0;
    /* The detector's transforms run in line on the COMPUTE lane.  On its own (BACKGROUND) lane they would overlap the
     * frame path, but every window then needs two device-side waits between the lanes, and a barrier packet parked in
     * one hardware queue slows the other queues of the process down (see download_thread): measured 1.8-2.4 GS/s
     * against 4-5 GS/s, depending on which hardware queues the lanes were given.  TSDR_GPU_DETECTOR_LANE=background
     * opts back in. */
    {
        const char *lane = getenv("TSDR_GPU_DETECTOR_LANE");
        (void)tsdrgpu_autocorr_set_async(e->ac, lane && lane[0] == 'b');
    }
    plot_msg_t nm;
    memset(&nm, 0, sizeof(nm));
    uint32_t n = 0;
    tsdrgpu_autocorr_geometry(e->ac, &nm.flo, &nm.flen, &nm.llo, &nm.llen, &e->ac_capture, &n);
    nm.rate = fs;
    if (tsdrgpu_alloc_host(e->g, (void **)&nm.h_frame, sizeof(double) * (size_t)nm.flen) ||
        tsdrgpu_alloc_host(e->g, (void **)&nm.h_line, sizeof(double) * (size_t)nm.llen)) {
        tsdrgpu_free_host(e->g, nm.h_frame);
        tsdrgpu_autocorr_destroy(e->ac);
        e->ac = NULL;
        e->ac_failed_rate = fs;
        return;
    }
    /* swap the message inside the critical section, once the host is done with the previous plots */
    pthread_mutex_lock(&e->pm);
    while (e->plot_pending && A_LD(e->alive)) {
        pthread_mutex_unlock(&e->pm);
        struct timespec ts = {0, 1000000};
        nanosleep(&ts, NULL);
        pthread_mutex_lock(&e->pm);
    }
    plot_msg_t old = e->plot;
    e->plot = nm;
    pthread_mutex_unlock(&e->pm);
    tsdrgpu_free_host(e->g, old.h_frame);
    tsdrgpu_free_host(e->g, old.h_line);
    e->ac_rate = fs;
    e->ac_failed_rate = 0;
}

/* A plot update: snapshot of the plots on the detector's lane (in certified mode behind the argmax kernels that
 * produce the certificate), picked up by the plot thread.  Skipped while the host still shows the previous one. */
static void publish_plot(struct engine *e)
{
    pthread_mutex_lock(&e->pm);
    const int busy = e->plot_pending;
    pthread_mutex_unlock(&e->pm);
    if (busy) return;
    uint64_t calls = 0;
    const double *snap = NULL;
    const int certify = e->ac_certified && tsdrgpu_autocorr_argmax_async(e->ac) == 0;
    if (tsdrgpu_autocorr_plots_snapshot(e->ac, &snap, &calls) == 0 &&
        tsdrgpu_event_record(e->g, e->plot_ready, tsdrgpu_autocorr_lane(e->ac)) == 0) {
        pthread_mutex_lock(&e->pm);
        e->plot.calls = calls;
        e->plot.d_snapshot = snap;
        e->plot.certify = certify;
        e->plot.ac = e->ac;
        e->plot_pending = 1;
        pthread_cond_signal(&e->p_nonempty);
        pthread_mutex_unlock(&e->pm);
    } else if (certify) {
        int32_t a, b;
        (void)tsdrgpu_autocorr_argmax_result(e->ac, &a, &b); /* nobody will collect it */
    }
}

static void run_detector(struct engine *e, uint32_t fs)
{
    tsdr_lib_t *t = e->t;
    if (A_LD(t->params_int[PARAM_AUTOCORR_PLOTS_OFF]) || e->iq_is_mag) { e->det.rd = e->det.wr = 0; return; } /* nothing is fed in super mode */
    if (fs == e->ac_failed_rate) { e->det.rd = e->det.wr = 0; return; } /* no detector exists for this rate */
    if (!e->ac || e->ac_rate != fs) {
        detector_rebuild(e, fs);
        if (!e->ac) { e->det.rd = e->det.wr = 0; return; }
    }
    const uint32_t capture = e->ac_capture;
    if (A_LD(e->det_promote) && !e->det_replaying) { /* the plot thread held a plot back: its epoch once more, in the reference's arithmetic */
        pthread_mutex_lock(&e->pm);
        const int busy = e->plot_pending; /* (the plot thread clears it right after raising the request) */
        pthread_mutex_unlock(&e->pm);
        if (!busy) {
            A_ST(e->det_promote, 0);
            e->det_replaying = 1;
        }
    }
    if (e->det_replaying && !A_LD(t->detector_purge) && !A_LD(t->params_int[PARAM_AUTOCORR_PLOTS_RESET])) {
        /* DET_REPLAY_STEP windows per device-thread turn, the frame path's launches in between; the capture windows that
         * arrive meanwhile are skipped, like the reference's detector thread skips what arrives while it is busy
         * (frameratedetector.c:128-187).  (A pending reset cancels the replay: the loop below sees to it.) */
        int remaining = 0;
        if (!gpu_ok(e, tsdrgpu_autocorr_promote_step(e->ac, DET_REPLAY_STEP, &remaining), "autocorr promote")) return;
        e->det.rd += ((e->det.wr - e->det.rd) / 2 / capture) * (size_t)capture * 2;
        if (remaining > 0) return;
        e->det_replaying = 0;
        STAT_ADD(e->n_promotions, 1);
        publish_plot(e);
    }
    while ((e->det.wr - e->det.rd) / 2 >= capture && A_LD(t->running)) {
        if (A_LD(t->detector_purge)) { /* frameratedetector.c:171-176 */
            A_ST(t->detector_purge, 0);
            tsdrgpu_autocorr_reset(e->ac);
            A_ST(e->det_promote, 0);
            e->det_replaying = 0;
        }
        if (A_LD(t->params_int[PARAM_AUTOCORR_PLOTS_RESET])) { /* frameratedetector.c:97-104 */
            const uint32_t orig = __atomic_exchange_n(&t->params_int[PARAM_AUTOCORR_PLOTS_RESET], 0, __ATOMIC_ACQ_REL);
            tsdrgpu_autocorr_reset(e->ac);
            A_ST(e->det_promote, 0);
            e->det_replaying = 0;
            if (orig == 1) {
                pthread_mutex_lock(&e->pm);
                e->plot_reset_announce = 1;
                pthread_cond_signal(&e->p_nonempty);
                pthread_mutex_unlock(&e->pm);
            }
        }
        if (e->ac_certified && !e->det_replaying) {
            /* a certified epoch that is about to outgrow the retention ring continues in the reference's arithmetic: replayed
             * here in steps, not by the library in one go (0.28 s of transforms for 2048 windows of 2^22 samples) */
            int ring = 0, ready = 0, kept = 0, exact = 0; /* (ready: what of the ring is allocated so far, tsdrgpu.h) */
            if (tsdrgpu_autocorr_retention(e->ac, &ring, &ready, &kept, &exact) == 0 && !exact && ring > 0 && kept + 1 > ready) e->det_replaying = 1;
        }
        if (e->det_replaying) return; /* (the next turn starts stepping, above) */
        if (!gpu_ok(e, tsdrgpu_autocorr_run(e->ac, e->det.d + e->det.rd, 1, capture, 1, 0), "autocorr")) return;
        e->det.rd += (size_t)capture * 2;
        STAT_ADD(e->n_windows, 1);
        /* the window is read on the detector's lane; the COMPUTE lane must not recycle that memory before (process_block) */
        if (tsdrgpu_autocorr_lane(e->ac) != TSDRGPU_LANE_COMPUTE) { /* (in line, the lane's own order does it) */
            if (tsdrgpu_event_record(e->g, e->det_read, tsdrgpu_autocorr_lane(e->ac)) == 0) e->det_read_valid = 1;
            else tsdrgpu_sync(e->g);
        }
        if (A_LD(t->params_int[PARAM_AUTOCORR_DUMP])) {
            A_ST(t->params_int[PARAM_AUTOCORR_DUMP], 0);
            dump_autocorr(e);
            pthread_mutex_lock(&e->pm);
            e->plot_dumped_announce = 1;
            pthread_cond_signal(&e->p_nonempty);
            pthread_mutex_unlock(&e->pm);
        }
        publish_plot(e);
    }
}

/* Hands a batch's frames to the download thread: one pinned slot and one event per frame; the video thread picks a
 * frame up when its event has fired.  A full queue drops the frame (the viewer is slower than the stream), like the
 * reference's lossy video ring. */
static void deliver_frames(struct engine *e, out_buf_t *ob, int F, int W, int H)
{
    const size_t P = (size_t)W * H;
    for (int f = 0; f < F; f++) {
        const int announce = e->pp_runs++ > AUTOGAIN_REPORT_EVERY_FRAMES;
        if (announce) e->pp_runs = 0;
        STAT_ADD(e->n_frames_made, 1);
        pthread_mutex_lock(&e->fm);
        if (e->fq_count == NFRAMEQ) {
            pthread_mutex_unlock(&e->fm);
            STAT_ADD(e->n_frames_lost, 1);
            continue;
        }
        frame_slot_t *s = &e->fq[(e->fq_head + e->fq_count) % NFRAMEQ]; /* free: only this thread produces */
        pthread_mutex_unlock(&e->fm);
        if (s->cap < P) {
            tsdrgpu_free_host(e->g, s->h);
            s->h = NULL; s->cap = 0;
            if (tsdrgpu_alloc_host(e->g, (void **)&s->h, P * sizeof(float))) continue;
            s->cap = P;
        }
        s->d_src = e->t->rgb_cb ? (const void *)(ob->d_rgb + (size_t)f * P) : (const void *)(ob->d + (size_t)f * P);
        s->d_info_src = ob->d_info + f;
        s->ob = ob;
        s->width = W; s->height = H;
        s->announce_autogain = announce;
        pthread_mutex_lock(&e->fm);
        e->fq_count++;
        ob->pending++;
        ob->busy = 1;
        pthread_cond_broadcast(&e->f_queued);
        pthread_mutex_unlock(&e->fm);
    }
}

static void track_off(struct engine *e);

/* parameter sets the fused run handles itself (the others silently take its split path: nothing gained) */
static int fused_wanted(const tsdrgpu_pp_params_t *p)
{
    return !p->lowpass_before_sync && !p->autogain_after_proc && !p->autoshift && !p->pll;
}

static void run_frames(struct engine *e)
{
    tsdr_lib_t *t = e->t;
    while (A_LD(t->running)) {
        pthread_mutex_lock(&t->lock);
        const int W = t->width, H = t->height;
        const float blur = t->motionblur;
        pthread_mutex_unlock(&t->lock);
        if (W < 1 || H < 1 || (long long)W * H > TSDRGPU_MAX_FRAME_PIXELS) {
            /* More pixels than tsdr_readasync itself would have accepted at the start (MAX_ARR_SIZE, TSDRLibrary.c:31,489; a host can
             * only get here through tsdr_setresolution mid-stream): nothing to show while it is set, nothing kept for later, and no
             * frame grid for the resampler's tracking to go on with (run_resampler does the same in its branch). */
            e->pix.rd = e->pix.wr;
            track_off(e);
            return;
        }
        const size_t P = (size_t)W * H;
        size_t avail = e->pix.wr - e->pix.rd;
        if (avail < P) return;
        int F = (int)(avail / P);
        if (F > MAX_FRAME_BATCH) F = MAX_FRAME_BATCH;
        tsdrgpu_pp_params_t prm;
        prm.lowpass_before_sync = (int)A_LD(t->params_int[PARAM_LOW_PASS_BEFORE_SYNC]);
        prm.autogain_after_proc = (int)A_LD(t->params_int[PARAM_AUTOGAIN_AFTER_PROCESSING]);
        prm.autoshift = (int)A_LD(t->params_int[PARAM_INT_AUTOSHIFT]);
        prm.pll = (int)A_LD(t->params_int[PARAM_INT_FRAMERATE_PLL]);
        prm.superresolution = (int)A_LD(t->params_int[PARAM_AUTOCORR_SUPERRESOLUTION]);
        prm.motionblur = blur;
        prm.lowpasscoeff = NORMALISATION_LOWPASS_COEFF;
        if (prm.pll) F = 1; /* the PLL's nudge feeds back into the geometry between frames */
        if (e->mm_nohead > 0 && F > e->mm_nohead) F = e->mm_nohead; /* frames from before the tracking started go on their own */
        out_buf_t *ob = &e->out[e->out_next];
        e->out_next = (e->out_next + 1) % NOUT;
        if (ob->busy) { /* its frames have left the device? */
            const double t0 = e->stats ? now_s() : 0.0;
            pthread_mutex_lock(&e->fm);
            while (ob->pending && A_LD(e->alive)) {
                struct timespec ts;
                deadline_ms(&ts, 30);
                pthread_cond_timedwait(&e->f_queued, &e->fm, &ts);
            }
            pthread_mutex_unlock(&e->fm);
            tsdrgpu_event_sync(e->g, ob->last_dl);
            ob->busy = 0;
            if (e->stats) e->s_dev_wait_out += now_s() - t0;
        }
        e->n_batches++;
        if (!ensure_dev(e, &ob->d, &ob->cap, P * (size_t)F)) return;
        if (ob->info_cap < F) {
            tsdrgpu_sync(e->g);
            tsdrgpu_free(e->g, ob->d_info);
            ob->d_info = NULL; ob->info_cap = 0;
            if (tsdrgpu_alloc(e->g, (void **)&ob->d_info, sizeof(tsdrgpu_pp_frameinfo_t) * MAX_FRAME_BATCH)) return;
            ob->info_cap = MAX_FRAME_BATCH;
        }
        tsdrgpu_pp_frameinfo_t info;
        /* A backlog (the source runs ahead of us) goes through the FUSED run — the path bench.py times: the resampler has
         * left every frame's min / max (frame tracking, run_resampler), so one trip over the raw frames gathers the sync
         * detector's sums and writes the normalised frames (12 instead of 16 bytes per pixel moved).  Frames and state are
         * bit-identical to tsdrgpu_postproc_run's (tests/test_gpu_postproc.py).  Frames that arrive one or two at a time
         * — a live source — take the plain run: there the launches, not the bytes, are what costs. */
        /* the min / max entries belong to the frame grid the resampler tracked (track_P pixels per frame): a
         * tsdr_setresolution between the resampler call and this turn makes them another grid's — those frames take the plain
         * run and the tracking restarts at the next resampler call */
        if (e->track_P && (int64_t)P != e->track_P) track_off(e);
        if (e->mm_nohead > 0 && F > e->mm_nohead) F = e->mm_nohead;
        const int fuse = e->mm_nohead == 0 && (int64_t)P == e->track_P && F >= FUSE_MIN_FRAMES && e->mm_n >= F && fused_wanted(&prm);
        if (fuse) {
            if (!gpu_ok(e, tsdrgpu_postproc_begin_minmax(e->pp, e->pix.d + e->pix.rd, F, W, H, &prm, e->d_mm_min + e->mm_off, e->d_mm_max + e->mm_off, ob->d),
                        "postproc (fused)") ||
                !gpu_ok(e, tsdrgpu_postproc_finish(e->pp, ob->d, NULL), "postproc (fused)"))
                return;
            e->n_fused_batches++;
            STAT_ADD(e->n_fused_frames, F);
        } else if (!gpu_ok(e, tsdrgpu_postproc_run(e->pp, e->pix.d + e->pix.rd, F, W, H, &prm, ob->d, prm.pll ? &info : NULL), "postproc")) return;
        if (t->rgb_cb) {
            /* the JNI shim's pixel loop (TSDRLibraryNDK.c:222-276) on the device: frame after frame into the viewer's
             * persistent pixel buffer (transparent pixels keep their colour), a copy of which goes home */
            if (e->rgb_state_cap < P) {
                tsdrgpu_sync(e->g);
                tsdrgpu_free(e->g, e->d_rgb_state);
                e->d_rgb_state = NULL; e->rgb_state_cap = 0;
                if (tsdrgpu_alloc(e->g, (void **)&e->d_rgb_state, P * sizeof(int32_t))) return;
                e->rgb_state_cap = P;
            }
            if (ob->rgb_cap < P * (size_t)F) {
                tsdrgpu_sync(e->g);
                tsdrgpu_free(e->g, ob->d_rgb);
                ob->d_rgb = NULL; ob->rgb_cap = 0;
                if (tsdrgpu_alloc(e->g, (void **)&ob->d_rgb, P * (size_t)MAX_FRAME_BATCH * sizeof(int32_t))) return;
                ob->rgb_cap = P * (size_t)MAX_FRAME_BATCH;
            }
            for (int f = 0; f < F; f++)
                if (!gpu_ok(e, tsdrgpu_frame_to_rgb(e->g, ob->d + (size_t)f * P, e->d_rgb_state, (int64_t)P, t->rgb_inverted), "frame_to_rgb") ||
                    !gpu_ok(e, tsdrgpu_copy(e->g, ob->d_rgb + (size_t)f * P, e->d_rgb_state, P * sizeof(int32_t)), "rgb copy"))
                    return;
        }
        if (!gpu_ok(e, tsdrgpu_postproc_info_pack(e->pp, ob->d_info, F), "info") ||
            !gpu_ok(e, tsdrgpu_event_record(e->g, ob->done, TSDRGPU_LANE_COMPUTE), "event"))
            return;
        e->pix.rd += P * (size_t)F;
        /* the frames' min/max entries go with them (together with the read position: an early return above leaves both alone) */
        if (e->mm_nohead > 0) e->mm_nohead -= F; /* (F was capped to the head frames above) */
        else if (e->mm_n >= F) { e->mm_off += F; e->mm_n -= F; if (!e->mm_n) e->mm_off = 0; }
        else { e->mm_off = e->mm_n = 0; }
        if (prm.pll && info.pll_fired) { /* syncdetector.c:149-151: the one place the host has to see a result at once */
            pthread_mutex_lock(&t->lock);
            t->refreshrate -= info.frameratediff;
            tsdr_geometry_update(t, t->samplerate);
            const double rate = t->refreshrate;
            pthread_mutex_unlock(&t->lock);
            tsdr_announce_value(t, VALUE_ID_PLL_FRAMERATE, rate, 0);
        }
        deliver_frames(e, ob, F, W, H);
    }
}

/* stops the resampler's frame tracking; frames still waiting in the pixel stream lose their min/max (plain run) */
static void track_off(struct engine *e)
{
    if (e->track_P) (void)tsdrgpu_resampler_track_frames(e->rs, 0, 0);
    e->track_P = 0;
    e->mm_off = e->mm_n = 0;
    e->mm_nohead = 1 << 30; /* every frame now in the stream, and every later one until tracking restarts */
    e->mm_drop_first = 0;
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
        const int usable = W >= 1 && H >= 1 && (long long)W * H <= TSDRGPU_MAX_FRAME_PIXELS && refresh > 0;
        const int chunk = usable ? (int)(FRAMES_TO_POLL * fs / refresh) : 0; /* TSDRLibrary.c:335 */
        if (chunk <= 0) { /* no frame can be made of this geometry: the samples are not kept for one (the stream would grow without bound) */
            e->iq.rd = e->iq.wr;
            track_off(e);
            return;
        }
        const double up = W * H * refresh, down = fs; /* TSDRLibrary.c:340 */
        const int totalpixels = W * H;
        const size_t have = (e->iq.wr - e->iq.rd) / per;
        int nchunks = (int)(have / (size_t)chunk);
        if (nchunks <= 0) break;
        /* while pixels are being skipped or a manual shift is pending go chunk by chunk like the reference */
        if (e->pix_difference != 0 || A_LD(t->syncoffset) != 0) nchunks = 1;
        else if (nchunks > MAX_CHUNKS_PER_CALL) nchunks = MAX_CHUNKS_PER_CALL;
        if (nchunks > 1 && A_LD(t->params_int[PARAM_INT_FRAMERATE_PLL])) {
            /* PLL on: a frame completed by a chunk may nudge the refresh rate, which the NEXT chunk's ratio must already
             * see (TSDRLibrary.c:335-340 re-reads the geometry per chunk) — so one call takes the chunks up to and
             * including the one that completes the frame being filled, no further */
            const size_t pending = e->pix.wr - e->pix.rd;
            int k = 1;
            while (k < nchunks) {
                const int64_t c = tsdrgpu_resample_count(e->rs, (uint32_t)chunk, k, up, down);
                if (c < 0 || pending + (size_t)c >= (size_t)totalpixels) break;
                k++;
            }
            nchunks = k;
        }
        const int64_t count = tsdrgpu_resample_count(e->rs, (uint32_t)chunk, nchunks, up, down);
        if (count < 0) return;
        e->n_resample_calls++;
        const int nearest = (int)A_LD(t->params_int[PARAM_NEAREST_NEIGHBOUR_RESAMPLING]);
        int64_t n = 0;
        if (e->pix_difference == 0) {
            /* the usual case: straight into the pixel stream */
            if (!stream_reserve(e, &e->pix, (size_t)count)) return;
            /* frame tracking for the fused run: on while whole batches can use it (area mode, default stage order, frames
             * of >= 4096 pixels), restarted whenever the frame grid of the pixel stream moved */
            const int want_track = !nearest && totalpixels >= 4096 && !A_LD(t->params_int[PARAM_LOW_PASS_BEFORE_SYNC]) &&
                                   !A_LD(t->params_int[PARAM_AUTOGAIN_AFTER_PROCESSING]) && !A_LD(t->params_int[PARAM_INT_AUTOSHIFT]) &&
                                   !A_LD(t->params_int[PARAM_INT_FRAMERATE_PLL]) && e->d_mm_min != NULL;
            if (!want_track) track_off(e);
            else if (e->track_P != (int64_t)totalpixels) {
                const size_t live = e->pix.wr - e->pix.rd;
                const int64_t phase = (int64_t)(live % (size_t)totalpixels);
                if (tsdrgpu_resampler_track_frames(e->rs, (int64_t)totalpixels, phase) == 0) {
                    e->track_P = totalpixels;
                    e->mm_off = e->mm_n = 0;
                    e->mm_nohead = (int)(live / (size_t)totalpixels) + (phase ? 1 : 0); /* their pixels (or part of them) predate the tracking */
                    e->mm_drop_first = phase ? 1 : 0;
                } else track_off(e);
            }
            if (!gpu_ok(e, tsdrgpu_resample(e->rs, e->iq.d + e->iq.rd, !e->iq_is_mag, (uint32_t)chunk, nchunks, up, down, nearest,
                                            e->pix.d + e->pix.wr, (int64_t)(e->pix.cap - e->pix.wr), &n),
                        "resample"))
                return;
            e->pix.wr += (size_t)n;
            if (e->track_P) {
                const float *mn = NULL, *mx = NULL;
                int k = 0;
                if (tsdrgpu_resampler_frame_minmax(e->rs, &mn, &mx, &k) == 0 && k > 0) {
                    const int skip = e->mm_drop_first ? 1 : 0;
                    e->mm_drop_first = 0;
                    const int take = k - skip;
                    if (take > 0) {
                        if (e->mm_off + e->mm_n + take > MM_CAP) track_off(e); /* (thousands of frames waiting: cannot happen with the queues' sizes) */
                        else if (gpu_ok(e, tsdrgpu_copy(e->g, e->d_mm_min + e->mm_off + e->mm_n, mn + skip, (size_t)take * sizeof(float)), "min/max") &&
                                 gpu_ok(e, tsdrgpu_copy(e->g, e->d_mm_max + e->mm_off + e->mm_n, mx + skip, (size_t)take * sizeof(float)), "min/max"))
                            e->mm_n += take;
                        else return;
                    }
                } else if (k < 0) track_off(e);
            }
        } else {
            track_off(e); /* pixels are about to be skipped: the frame grid moves */
            /* dsp_dropped_compensation_add with a ring that always accepts (dsp.c:326-346) */
            if (!ensure_dev(e, &e->d_rs, &e->rs_cap, (size_t)count + 16)) return;
            if (!gpu_ok(e, tsdrgpu_resample(e->rs, e->iq.d + e->iq.rd, !e->iq_is_mag, (uint32_t)chunk, nchunks, up, down, nearest,
                                            e->d_rs, (int64_t)e->rs_cap, &n),
                        "resample"))
                return;
            if ((int64_t)n <= e->pix_difference) e->pix_difference -= n;
            else {
                if (!stream_append(e, &e->pix, e->d_rs + e->pix_difference, (size_t)(n - e->pix_difference))) return;
                e->pix_difference = 0;
            }
        }
        e->iq.rd += (size_t)nchunks * chunk * per;
        /* manual sync, TSDRLibrary.c:345-346 */
        const int so = __atomic_exchange_n(&t->syncoffset, 0, __ATOMIC_ACQ_REL); /* (tsdr_sync adds to it atomically: no shift is lost) */
        e->pix_difference = drop_shift_with(e->pix_difference, (uint32_t)totalpixels, -(int64_t)so);
        const double tf = e->stats ? now_s() : 0.0;
        run_frames(e);
        if (e->stats) e->s_dev_frames += now_s() - tf;
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
        if (A_LD(t->samplerate_real) != e->super_rate || !e->d_hops[0]) {
            e->super_rate = A_LD(t->samplerate_real);
            pthread_mutex_lock(&t->lock);
            const double refresh = t->refreshrate;
            pthread_mutex_unlock(&t->lock);
            e->super_frame = (int)(e->super_rate / refresh);
            e->super_to_gather = SUPER_FRAMES_TO_RECORD * e->super_frame;
            e->super_to_pause = (int)(SUPER_SECS_TO_PAUSE * e->super_rate);
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
            if (!gpu_ok(e, tsdrgpu_copy(e->g, e->d_hops[e->super_hop] + 2 * (size_t)e->super_gathered, d_blk, nfloats * sizeof(float)), "hop copy")) return 0;
            e->super_gathered += now;
        } else {
            const int remain = e->super_to_gather - e->super_gathered;
            if (!gpu_ok(e, tsdrgpu_copy(e->g, e->d_hops[e->super_hop] + 2 * (size_t)e->super_gathered, d_blk, (size_t)remain * 2 * sizeof(float)), "hop copy")) return 0;
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
            if (t->plugin.loaded) t->plugin.setbasefreq(A_LD(t->centfreq) + (uint32_t)((e->super_hop - SUPER_HOPS / 2) * (int64_t)e->super_rate));
            e->super_state = SUPER_PAUSE;
        }
    }
    return 0;
}

/* the blocks collected by process_block go behind each other into the detector's and the resampler's stream */
static void gather_flush(struct engine *e)
{
    if (!e->gn) return;
    gpu_ok(e, tsdrgpu_gather2(e->g, e->g_det, e->g_iq, e->g_src, e->g_bytes, e->gn), "append");
    e->gn = 0;
}

/* One plugin block, already on the device (slot->d): everything below only queues work. */
static void process_block(struct engine *e, in_slot_t *slot)
{
    tsdr_lib_t *t = e->t;
    const float *d_blk = slot->d;
    const size_t nfloats = slot->nfloats;
    /* TSDRPlugin_RawFile.c:241-261 on the device, bit-exact (double division, float store) */
    if (nfloats && slot->raw_type != TSDRX_SAMPLE_FLOAT32 &&
        !gpu_ok(e, tsdrgpu_decode_samples(e->g, slot->d_raw, slot->raw_type, slot->d, (int64_t)nfloats), "decode"))
        return;
    const int64_t dropped = slot->dropped;
    const size_t size2 = nfloats / 2;

    if (A_LD(t->params_int[PARAM_AUTOCORR_SUPERRESOLUTION])) { /* TSDRLibrary.c:271-279 */
        gather_flush(e);
        if (!e->iq_is_mag) { e->iq.rd = e->iq.wr = 0; e->det.rd = e->det.wr = 0; e->iq_is_mag = 1; }
        uint32_t total = 0;
        if (nfloats && super_feed(e, d_blk, nfloats, dropped, &total)) {
            /* am_demod of the stitched buffer, then on to the resampler at 4x the rate */
            if (stream_reserve(e, &e->iq, total) && gpu_ok(e, tsdrgpu_am_demod(e->g, e->d_super_out, e->iq.d + e->iq.wr, total), "am_demod"))
                e->iq.wr += total;
        }
    } else {
        if (e->iq_is_mag || e->super_state != SUPER_STOPPED) { /* superb_stop, superbandwidth.c:256-264 */
            gather_flush(e);
            super_reset(e);
            if (t->plugin.loaded) t->plugin.setbasefreq(A_LD(t->centfreq));
            pthread_mutex_lock(&t->lock);
            tsdr_geometry_update(t, A_LD(t->samplerate_real));
            pthread_mutex_unlock(&t->lock);
            e->iq.rd = e->iq.wr = 0; e->det.rd = e->det.wr = 0; e->iq_is_mag = 0;
        }
        pthread_mutex_lock(&t->lock);
        const int block = (int)round(((t->width * t->height) << 1) * t->pixeltimeoversampletime); /* TSDRLibrary.c:284 */
        pthread_mutex_unlock(&t->lock);
        e->dev_difference = drop_shift_with(e->dev_difference, (uint32_t)block, dropped);
        const int drop_all = (int64_t)size2 <= e->dev_difference;
        const int plots_on = !A_LD(t->params_int[PARAM_AUTOCORR_PLOTS_OFF]);
        /* frameratedetector_run, frameratedetector.c:215-230 */
        if (plots_on && e->det_read_valid) { /* appends may compact the stream into memory an earlier window still occupies */
            tsdrgpu_lane_wait(e->g, TSDRGPU_LANE_COMPUTE, e->det_read);
            e->det_read_valid = 0;
        }
        const int usual = plots_on && dropped == 0 && !drop_all && nfloats && e->dev_difference == 0;
        /* anything but another usual block ends the run of blocks that one launch appends; so does a stream that has
         * to be compacted or grown first (stream_reserve may move it) */
        if (!usual || e->det.wr + nfloats > e->det.cap || e->iq.wr + nfloats > e->iq.cap) gather_flush(e);
        if (plots_on && dropped != 0) e->det.rd = e->det.wr = 0;
        if (usual && stream_reserve(e, &e->det, nfloats) && stream_reserve(e, &e->iq, nfloats)) {
            /* the usual block — nothing lost, nothing to skip — goes to both streams, together with its neighbours */
            if (!e->gn) { e->g_det = e->det.d + e->det.wr; e->g_iq = e->iq.d + e->iq.wr; }
            e->g_src[e->gn] = d_blk;
            e->g_bytes[e->gn] = nfloats * sizeof(float);
            e->gn++;
            e->det.wr += nfloats;
            e->iq.wr += nfloats;
            if (e->gn == 32) gather_flush(e);
        } else {
            if (plots_on && dropped == 0 && !drop_all && nfloats && !stream_append(e, &e->det, d_blk, nfloats)) e->det.rd = e->det.wr = 0;
            /* dsp_dropped_compensation_add, dsp.c:326-346 */
            if (drop_all) e->dev_difference -= (int64_t)size2;
            else if (stream_append(e, &e->iq, d_blk + 2 * e->dev_difference, nfloats - 2 * (size_t)e->dev_difference)) e->dev_difference = 0;
        }
    }
}

static void *device_thread(void *arg)
{
    struct engine *e = (struct engine *)arg;
    tsdr_lib_t *t = e->t;
    tsdrgpu_bind_thread(e->g);
    while (A_LD(t->running)) {
        pthread_mutex_lock(&e->qm);
        if (!A_LD(e->q_count)) {
            struct timespec ts;
            deadline_ms(&ts, 30);
            pthread_cond_timedwait(&e->q_nonempty, &e->qm, &ts);
            pthread_mutex_unlock(&e->qm);
            continue;
        }
        /* everything that is queued goes into the sample streams first: when the source runs ahead, the resampler and
         * the frame path below then work on several blocks per launch instead of one */
        /* ... but at most half of the slots per turn: the plugin thread refills (and the UPLOAD lane fills) the other half
         * while this one is being worked on */
        const int queued = A_LD(e->q_count);
        const int head = e->q_head, n = queued > NSLOT / 2 ? NSLOT / 2 : queued;
        pthread_mutex_unlock(&e->qm);
        const double t0 = e->stats ? now_s() : 0.0;
        /* blocks whose DMA may still be in flight (the plugin thread did not wait for it): the UPLOAD lane is in order,
         * so the newest of them stands for all — waited for here, on the host, never on the device (see download_thread) */
        for (int i = n - 1; i >= 0; i--) {
            in_slot_t *sl = &e->slot[(head + i) % NSLOT];
            if (!sl->uploaded_valid || !sl->nfloats) continue;
            gpu_ok(e, tsdrgpu_event_sync(e->g, sl->uploaded), "upload wait");
            break;
        }
        for (int i = 0; i < n; i++) process_block(e, &e->slot[(head + i) % NSLOT]);
        gather_flush(e);
        /* the plugin thread may refill these slots once the COMPUTE lane has read them */
        for (int i = 0; i < n; i++) {
            in_slot_t *sl = &e->slot[(head + i) % NSLOT];
            if (!sl->nfloats) continue;
            if (tsdrgpu_event_record(e->g, sl->consumed, TSDRGPU_LANE_COMPUTE) == 0) sl->consumed_valid = 1;
            else tsdrgpu_sync(e->g);
        }
        pthread_mutex_lock(&e->qm);
        e->q_head = (e->q_head + n) % NSLOT;
        __atomic_fetch_sub(&e->q_count, n, __ATOMIC_RELEASE);
        pthread_mutex_unlock(&e->qm);
        const double t1 = e->stats ? now_s() : 0.0;
        run_resampler(e);
        const double t2 = e->stats ? now_s() : 0.0;
        pthread_mutex_lock(&t->lock);
        const uint32_t fs_now = t->samplerate;
        pthread_mutex_unlock(&t->lock);
        run_detector(e, fs_now);
        if (e->stats) {
            const double t3 = now_s();
            e->s_dev_busy += t3 - t0;
            e->s_dev_blocks += t1 - t0;
            e->s_dev_rs += t2 - t1; /* includes the frame path, which run_resampler drives */
            e->s_dev_det += t3 - t2;
        }
    }
    /* a device call that failed on the plugin's thread left the stop to us (gpu_ok) */
    if (A_LD(e->failed)) (void)tsdr_plugin_stop_once(t);
    return NULL;
}

/* ---- entry ----------------------------------------------------------------------------- */
void engine_stats(struct engine *e, tsdrx_stats_t *out)
{
    out->blocks_in = STAT_GET(e->n_blocks);
    out->blocks_lost = STAT_GET(e->n_blocks_lost);
    out->frames_made = STAT_GET(e->n_frames_made);
    out->frames_lost_to_viewer = STAT_GET(e->n_frames_lost);
    out->windows = STAT_GET(e->n_windows);
    out->plots_held = STAT_GET(e->n_plots_held);
    out->epochs_replayed = STAT_GET(e->n_promotions);
    out->frames_fused = STAT_GET(e->n_fused_frames);
}

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
    int ok = tsdrgpu_create(&e->g, dev) == 0 && tsdrgpu_resampler_create(e->g, &e->rs) == 0 && tsdrgpu_postproc_create(e->g, &e->pp) == 0 &&
             tsdrgpu_event_create(e->g, &e->plot_ready) == 0 && tsdrgpu_event_create(e->g, &e->plot_home) == 0 &&
             tsdrgpu_event_create(e->g, &e->det_read) == 0;
    for (int i = 0; ok && i < NSLOT; i++) ok = tsdrgpu_event_create(e->g, &e->slot[i].consumed) == 0 && tsdrgpu_event_create(e->g, &e->slot[i].uploaded) == 0;
    if (ok) ok = tsdrgpu_alloc(e->g, (void **)&e->d_mm_min, MM_CAP * sizeof(float)) == 0 && tsdrgpu_alloc(e->g, (void **)&e->d_mm_max, MM_CAP * sizeof(float)) == 0;
    e->mm_nohead = 1 << 30; /* no tracking yet */
    for (int i = 0; ok && i < NFRAMEQ; i++)
        ok = tsdrgpu_event_create(e->g, &e->fq[i].ready) == 0 && tsdrgpu_alloc_host(e->g, (void **)&e->fq[i].h_info, sizeof(tsdrgpu_pp_frameinfo_t)) == 0;
    for (int i = 0; ok && i < NOUT; i++) ok = tsdrgpu_event_create(e->g, &e->out[i].done) == 0 && tsdrgpu_event_create(e->g, &e->out[i].last_dl) == 0;
    if (!ok) {
        if (e->g) {
            for (int i = 0; i < NSLOT; i++) { tsdrgpu_event_destroy(e->g, e->slot[i].consumed); tsdrgpu_event_destroy(e->g, e->slot[i].uploaded); }
            tsdrgpu_free(e->g, e->d_mm_min); tsdrgpu_free(e->g, e->d_mm_max);
            for (int i = 0; i < NFRAMEQ; i++) { tsdrgpu_event_destroy(e->g, e->fq[i].ready); tsdrgpu_free_host(e->g, e->fq[i].h_info); }
            for (int i = 0; i < NOUT; i++) { tsdrgpu_event_destroy(e->g, e->out[i].done); tsdrgpu_event_destroy(e->g, e->out[i].last_dl); }
            tsdrgpu_event_destroy(e->g, e->plot_ready); tsdrgpu_event_destroy(e->g, e->plot_home);
            tsdrgpu_event_destroy(e->g, e->det_read);
            if (e->pp) tsdrgpu_postproc_destroy(e->pp);
            if (e->rs) tsdrgpu_resampler_destroy(e->rs);
            tsdrgpu_destroy(e->g);
        }
        free(e);
        return tsdr_set_error(t, TSDR_CANNOT_OPEN_DEVICE, "No usable MI355X/HIP device: this library has no CPU path.");
    }
    /* Sync-detector decisions that are toss-ups at the precision of the collapsed strips are detected and redone
     * with the reference's own strip arithmetic (tsdrgpu_postproc_set_exact_ties; on by default in the library as
     * well).  TSDR_GPU_EXACT_SYNC=0 / TSDR_GPU_EXACT=0 opt out. */
    (void)tsdrgpu_postproc_set_exact_ties(e->pp, exact_wanted("TSDR_GPU_EXACT_SYNC"));
    {
        const char *st = getenv("TSDR_GPU_STATS");
        e->stats = st && st[0] == '1';
        e->t_start = now_s();
    }
    {   /* DMA straight out of the plugin's memory only when the plugin promises that it is stable
         * (tsdrplugin_memory_stable, include/TSDRLibraryExt.h); TSDR_GPU_ZEROCOPY=0 / =1 override */
        const char *z = getenv("TSDR_GPU_ZEROCOPY");
        const int promise = t->plugin.memory_stable ? t->plugin.memory_stable() : 0;
        if (z && (z[0] == '0' || z[0] == '1')) e->zero_copy = z[0] == '1';
        else e->zero_copy = (promise & TSDRX_MEMORY_MAPPED) != 0;
        /* ... and its DMAs overlap the plugin's next blocks only when the plugin also promises that a block's contents stay
         * untouched while it streams (decided below, once it is known which of its two entry points is used);
         * and only with TSDR_GPU_ASYNC_UPLOAD=1 — the default waits for every DMA: measured faster, see below) */
        e->immutable = promise;
    }
    {   /* the helper for bounce-buffer copies (only plugins without the promise need it); TSDR_GPU_COPY_THREAD=0: off */
        const char *c = getenv("TSDR_GPU_COPY_THREAD");
        pthread_mutex_init(&e->cm, NULL); pthread_cond_init(&e->c_wake, NULL);
        if (!(c && c[0] == '0') && !e->zero_copy) e->copy_thread_on = pthread_create(&e->th_copy, NULL, copy_thread, e) == 0;
    }
    pthread_mutex_init(&e->qm, NULL); pthread_cond_init(&e->q_nonempty, NULL);
    pthread_mutex_init(&e->fm, NULL); pthread_cond_init(&e->f_nonempty, NULL); pthread_cond_init(&e->f_queued, NULL);
    pthread_mutex_init(&e->pm, NULL); pthread_cond_init(&e->p_nonempty, NULL);
    A_ST(e->alive, 1);
    t->eng = e;
    /* frameratedetector_startthread flushes the cached estimation, frameratedetector.c:203-209 */
    A_ST(t->detector_purge, 1);
    A_ST(t->params_int[PARAM_AUTOCORR_PLOTS_RESET], 2);

    pthread_t th_dev, th_video, th_plot, th_down;
    pthread_create(&th_dev, NULL, device_thread, e);
    pthread_create(&th_video, NULL, video_thread, e);
    pthread_create(&th_down, NULL, download_thread, e);
    pthread_create(&th_plot, NULL, plot_thread, e);

    /* blocks until tsdr_stop / plugin failure; a plugin that offers its blocks in their native sample format
     * (TSDRLibraryExt.h) is taken up on it unless TSDR_GPU_RAW=0 */
    const char *rawenv = getenv("TSDR_GPU_RAW");
    const int use_raw = t->plugin.readasync_raw && !(rawenv && rawenv[0] == '0');
    {   /* TSDR_GPU_ASYNC_UPLOAD=1 opts in to leaving a block's DMA in flight when the callback returns (two in flight at a
         * time).  Measured on MI355X it is SLOWER than waiting for every DMA — 20-25 GB/s through the UPLOAD lane against
         * 35-41 GB/s (profiles/round4_e2e_variants.txt) — so the default waits, as in round 3. */
        const char *a = getenv("TSDR_GPU_ASYNC_UPLOAD");
        e->async_upload = a && a[0] == '1';
        e->immutable = e->async_upload && e->zero_copy && (e->immutable & (use_raw ? TSDRX_MEMORY_IMMUTABLE_RAW : TSDRX_MEMORY_IMMUTABLE)) != 0;
    }
    const int status = use_raw ? t->plugin.readasync_raw(on_block_raw, e) : t->plugin.readasync(on_block, e);

    A_ST(t->running, 0);
    pthread_join(th_dev, NULL);
    A_ST(e->alive, 0);
    pthread_mutex_lock(&e->fm); pthread_cond_broadcast(&e->f_nonempty); pthread_cond_broadcast(&e->f_queued); pthread_mutex_unlock(&e->fm);
    pthread_mutex_lock(&e->pm); pthread_cond_broadcast(&e->p_nonempty); pthread_mutex_unlock(&e->pm);
    pthread_join(th_down, NULL);
    pthread_join(th_video, NULL);
    pthread_join(th_plot, NULL);
    if (e->copy_thread_on) {
        pthread_mutex_lock(&e->cm);
        A_ST(e->copy_quit, 1);
        pthread_cond_signal(&e->c_wake);
        pthread_mutex_unlock(&e->cm);
        pthread_join(e->th_copy, NULL);
    }
    pthread_mutex_destroy(&e->cm); pthread_cond_destroy(&e->c_wake);

    if (e->stats) {
        const double T = now_s() - e->t_start;
        fprintf(stderr,
                "tsdr stats: %.2f s | blocks in %ld lost %ld | frames made %ld lost-to-viewer %ld in %ld batches | resample calls %ld | windows %ld | "
                "page-locked ranges %d\n"
                "tsdr stats: plugin thread busy %.0f%% (DMA wait %.0f%%) | device thread busy %.0f%% (waiting for output buffers %.0f%%) | "
                "video thread: waiting for frames %.0f%%, in the callback %.0f%%\n"
                "tsdr stats: device thread: appending blocks %.0f%% | resampler %.0f%% | frame path %.0f%% | detector %.0f%%\n"
                "tsdr stats: frames through the fused run %ld in %ld batches | uploads %s\n"
                "tsdr stats: detector %s | plots held back for an exact replay %ld | epochs replayed %ld\n",
                T, e->n_blocks, e->n_blocks_lost, e->n_frames_made, e->n_frames_lost, e->n_batches, e->n_resample_calls, e->n_windows, e->nreg,
                100 * e->s_plugin_busy / T, 100 * e->s_plugin_dma / T, 100 * e->s_dev_busy / T, 100 * e->s_dev_wait_out / T,
                100 * e->s_video_wait / T, 100 * e->s_video_cb / T,
                100 * e->s_dev_blocks / T, 100 * (e->s_dev_rs - e->s_dev_frames) / T, 100 * e->s_dev_frames / T, 100 * e->s_dev_det / T,
                e->n_fused_frames, e->n_fused_batches,
                e->zero_copy ? (e->immutable ? "straight out of the plugin's memory, DMAs in flight behind the callback" : "straight out of the plugin's memory, each waited for")
                             : (e->copy_thread_on ? (e->async_upload ? "through pinned bounce buffers (copy in two halves), DMAs in flight behind the callback"
                                                                       : "through pinned bounce buffers (copy in two halves), each waited for")
                                                  : "through pinned bounce buffers"),
                e->ac_certified ? "certified (float32 + argmax certificate)" : "exact or plain (TSDR_GPU_AUTOCORR)", e->n_plots_held, e->n_promotions);
    }
    tsdrgpu_bind_thread(e->g);
    tsdrgpu_sync(e->g);
    tsdrgpu_lane_sync(e->g, TSDRGPU_LANE_UPLOAD);
    tsdrgpu_lane_sync(e->g, TSDRGPU_LANE_DOWNLOAD);
    if (e->super_state != SUPER_STOPPED && t->plugin.loaded) t->plugin.setbasefreq(A_LD(t->centfreq));
    /* tsdrplugin_readasync has returned and every DMA out of the plugin's memory is complete (UPLOAD lane drained
     * above): the ranges are unlocked now, before anything can reach tsdrplugin_cleanup / tsdrplugin_init — the memory
     * of a plugin that promised stability (tsdrplugin_memory_stable) is still allocated here */
    for (int i = 0; i < e->nreg; i++) tsdrgpu_host_unregister(e->g, e->reg[i].p);
    for (int i = 0; i < NSLOT; i++) {
        tsdrgpu_free_host(e->g, e->slot[i].h);
        tsdrgpu_free(e->g, e->slot[i].d);
        tsdrgpu_free(e->g, e->slot[i].d_raw);
        tsdrgpu_event_destroy(e->g, e->slot[i].consumed);
        tsdrgpu_event_destroy(e->g, e->slot[i].uploaded);
    }
    tsdrgpu_free(e->g, e->d_mm_min);
    tsdrgpu_free(e->g, e->d_mm_max);
    for (int i = 0; i < NFRAMEQ; i++) {
        tsdrgpu_free_host(e->g, e->fq[i].h);
        tsdrgpu_free_host(e->g, e->fq[i].h_info);
        tsdrgpu_event_destroy(e->g, e->fq[i].ready);
    }
    for (int i = 0; i < NOUT; i++) {
        tsdrgpu_free(e->g, e->out[i].d);
        tsdrgpu_free(e->g, e->out[i].d_rgb);
        tsdrgpu_free(e->g, e->out[i].d_info);
        tsdrgpu_event_destroy(e->g, e->out[i].done);
        tsdrgpu_event_destroy(e->g, e->out[i].last_dl);
    }
    for (int i = 0; i < SUPER_HOPS; i++) tsdrgpu_free(e->g, e->d_hops[i]);
    tsdrgpu_free_host(e->g, e->plot.h_frame);
    tsdrgpu_free_host(e->g, e->plot.h_line);
    tsdrgpu_event_destroy(e->g, e->plot_ready); tsdrgpu_event_destroy(e->g, e->plot_home);
    tsdrgpu_event_destroy(e->g, e->det_read);
    tsdrgpu_free(e->g, e->d_rs);
    tsdrgpu_free(e->g, e->d_rgb_state);
    tsdrgpu_free(e->g, e->d_super_out);
    stream_free(e, &e->iq); stream_free(e, &e->det); stream_free(e, &e->pix);
    if (e->ac) tsdrgpu_autocorr_destroy(e->ac);
    tsdrgpu_postproc_destroy(e->pp);
    tsdrgpu_resampler_destroy(e->rs);
    tsdrgpu_destroy(e->g);
    pthread_mutex_destroy(&e->qm); pthread_cond_destroy(&e->q_nonempty);
    pthread_mutex_destroy(&e->fm); pthread_cond_destroy(&e->f_nonempty); pthread_cond_destroy(&e->f_queued);
    pthread_mutex_destroy(&e->pm); pthread_cond_destroy(&e->p_nonempty);
    pthread_mutex_lock(&t->lock); /* tsdrx_get_stats may be looking at the engine */
    engine_stats(e, &t->last_stats);
    t->eng = NULL;
    pthread_mutex_unlock(&t->lock);
    const int failed = A_LD(e->failed);
    char fail_msg[400];
    memcpy(fail_msg, e->fail_msg, sizeof(fail_msg));
    free(e);
    if (failed) return tsdr_set_error(t, TSDR_CANNOT_OPEN_DEVICE, fail_msg);
    if (status != TSDR_OK) return tsdr_set_error(t, status, t->plugin.getlasterrortext());
    return tsdr_set_error(t, TSDR_OK, NULL);
}
