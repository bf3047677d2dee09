/* tests/sanitize/stub_tsdrgpu.c — a HOST-MEMORY STAND-IN for libtsdrgpu.so, for ONE purpose: letting ThreadSanitizer /
 * AddressSanitizer watch the host library's threads (tempestsdr_amd/csrc/host/engine.c: plugin, device, download, video,
 * plot and copy threads, their queues and flags) on a machine without a GPU.  TEST INFRASTRUCTURE ONLY.  It is linked into
 * tests/sanitize/host_stress_*_stub and nothing else; it is NOT a CPU fallback and computes NO signal processing: "device"
 * memory is malloc'ed, copies are memcpy, every kernel entry point fills its output with a constant and returns the
 * shapes (pixel counts, plot geometry, certificates) the engine's control flow needs.  Frames that come out of it are
 * meaningless by design.  The product (libTSDRLibrary.so) links the real libtsdrgpu.so and fails loudly without a device.
 *
 * Model of time: every "queued" operation completes inside the call, on the calling thread.  An event carries the
 * happens-before edge the real one does (hipEventRecord / hipEventSynchronize: what was queued before the record is
 * visible to whoever waited) as a release store / acquire load, so the sanitizer neither sees edges the GPU would not give
 * nor misses the ones it does. */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "tsdrgpu.h"

struct tsdrgpu { int device; };
struct tsdrgpu_event { int seq; };
struct tsdrgpu_resampler { tsdrgpu_t *g; int64_t track_P, phase; int k; float mn[512], mx[512]; };
struct tsdrgpu_postproc { tsdrgpu_t *g; int open, F; };
struct tsdrgpu_autocorr {
    tsdrgpu_t *g;
    uint32_t fs, capture;
    int32_t flo, flen, llo, llen;
    double *plots, *snap; /* the snapshot is a buffer of its own, like the real one's */
    float *corr;
    uint64_t calls;
    int certify, exact, async, argmax_pending, exact_epoch, replay_left, ring;
    long argmaxes;
};

/* Failure injection for the multi-rank sweep (tests/test_host_sanitizers.py): STUB_FAIL_CREATE_DEVICE=k: tsdrgpu_create(device k)
 * fails; STUB_FAIL_RUN_DEVICE=k: tsdrgpu_autocorr_run fails on that device's detector; STUB_UNCERTIFIED_DEVICE=k: only that device's
 * detector refuses its certificate (the premise check of the rank's own newest window); STUB_UNCERTIFIED_DEVICE=all: every rank's. */
static int stub_env_is(const char *name, int device)
{
    const char *e = getenv(name);
    if (!e || !e[0]) return 0;
    return (e[0] == 'a') || atoi(e) == device;
}

int tsdrgpu_create(tsdrgpu_t **out, int device)
{
    *out = NULL;
    if (stub_env_is("STUB_FAIL_CREATE_DEVICE", device)) return TSDRGPU_EHIP;
    *out = (tsdrgpu_t *)calloc(1, sizeof(**out));
    if (*out) (*out)->device = device;
    return *out ? 0 : TSDRGPU_ENOMEM;
}
void tsdrgpu_destroy(tsdrgpu_t *g) { free(g); }
const char *tsdrgpu_last_error(tsdrgpu_t *g) { (void)g; return "stub"; }
int tsdrgpu_sync(tsdrgpu_t *g) { (void)g; return 0; }
int tsdrgpu_bind_thread(tsdrgpu_t *g) { (void)g; return 0; }
int tsdrgpu_lane_sync(tsdrgpu_t *g, int lane) { (void)g; (void)lane; return 0; }

int tsdrgpu_alloc(tsdrgpu_t *g, void **p, size_t bytes)
{
    (void)g;
    *p = calloc(bytes ? bytes : 1, 1); /* (untouched pages: a memset of a 100 MB stream buffer costs a sanitizer seconds) */
    return *p ? 0 : TSDRGPU_ENOMEM;
}
int tsdrgpu_free(tsdrgpu_t *g, void *p) { (void)g; free(p); return 0; }
int tsdrgpu_alloc_host(tsdrgpu_t *g, void **p, size_t bytes) { return tsdrgpu_alloc(g, p, bytes); }
int tsdrgpu_free_host(tsdrgpu_t *g, void *p) { (void)g; free(p); return 0; }
int tsdrgpu_host_register(tsdrgpu_t *g, void *p, size_t n) { (void)g; (void)p; (void)n; return 0; }
int tsdrgpu_host_unregister(tsdrgpu_t *g, void *p) { (void)g; (void)p; return 0; }

static int cp(void *d, const void *s, size_t n) { if (n) memcpy(d, s, n); return 0; }
int tsdrgpu_copy(tsdrgpu_t *g, void *d, const void *s, size_t n) { (void)g; if (n) memmove(d, s, n); return 0; }
int tsdrgpu_download(tsdrgpu_t *g, void *d, const void *s, size_t n) { (void)g; return cp(d, s, n); }
int tsdrgpu_upload(tsdrgpu_t *g, void *d, const void *s, size_t n) { (void)g; return cp(d, s, n); }
int tsdrgpu_upload_lane(tsdrgpu_t *g, void *d, const void *s, size_t n) { (void)g; return cp(d, s, n); }
int tsdrgpu_download_lane(tsdrgpu_t *g, void *d, const void *s, size_t n) { (void)g; return cp(d, s, n); }
int tsdrgpu_gather2(tsdrgpu_t *g, void *d1, void *d2, const void *const *srcs, const size_t *bytes, int n)
{
    (void)g;
    size_t off = 0;
    for (int i = 0; i < n; i++) {
        memcpy((char *)d1 + off, srcs[i], bytes[i]);
        memcpy((char *)d2 + off, srcs[i], bytes[i]);
        off += bytes[i];
    }
    return 0;
}

int tsdrgpu_event_create(tsdrgpu_t *g, tsdrgpu_event_t **out)
{
    (void)g;
    *out = (tsdrgpu_event_t *)calloc(1, sizeof(**out));
    return *out ? 0 : TSDRGPU_ENOMEM;
}
void tsdrgpu_event_destroy(tsdrgpu_t *g, tsdrgpu_event_t *ev) { (void)g; free(ev); }
int tsdrgpu_event_record(tsdrgpu_t *g, tsdrgpu_event_t *ev, int lane)
{
    (void)g; (void)lane;
    __atomic_add_fetch(&ev->seq, 1, __ATOMIC_RELEASE);
    return 0;
}
int tsdrgpu_event_sync(tsdrgpu_t *g, tsdrgpu_event_t *ev) { (void)g; (void)__atomic_load_n(&ev->seq, __ATOMIC_ACQUIRE); return 0; }
int tsdrgpu_lane_wait(tsdrgpu_t *g, int lane, tsdrgpu_event_t *ev) { (void)lane; return tsdrgpu_event_sync(g, ev); }

static void fill(float *p, int64_t n, float v) { for (int64_t i = 0; i < n; i++) p[i] = v; }

int tsdrgpu_am_demod(tsdrgpu_t *g, const float *d_iq, float *d_out, int64_t n)
{
    (void)g;
    for (int64_t i = 0; i < n; i++) d_out[i] = fabsf(d_iq[2 * i]) + fabsf(d_iq[2 * i + 1]);
    return 0;
}
int tsdrgpu_decode_samples(tsdrgpu_t *g, const void *d_raw, int type, float *d_out, int64_t n)
{
    (void)g; (void)type;
    const unsigned char *b = (const unsigned char *)d_raw;
    for (int64_t i = 0; i < n; i++) d_out[i] = (float)b[i] / 255.0f; /* reads at least one byte per value of every format */
    return 0;
}
int tsdrgpu_frame_to_rgb(tsdrgpu_t *g, const float *d_frame, int32_t *d_rgb, int64_t n, int inverted)
{
    (void)g;
    for (int64_t i = 0; i < n; i++) d_rgb[i] = (int32_t)(d_frame[i] * 255.0f) ^ (inverted ? 0xffffff : 0);
    return 0;
}

/* ---- resampler ----------------------------------------------------------------------------------------------------- */
int tsdrgpu_resampler_create(tsdrgpu_t *g, tsdrgpu_resampler_t **out)
{
    *out = (tsdrgpu_resampler_t *)calloc(1, sizeof(**out));
    if (!*out) return TSDRGPU_ENOMEM;
    (*out)->g = g;
    return 0;
}
void tsdrgpu_resampler_destroy(tsdrgpu_resampler_t *rs) { free(rs); }
int64_t tsdrgpu_resample_count(tsdrgpu_resampler_t *rs, uint32_t chunk, int nchunks, double up, double down)
{
    (void)rs;
    if (!(up > 0) || !(down > 0)) return -1;
    return (int64_t)((double)chunk * nchunks * up / down);
}
int tsdrgpu_resample(tsdrgpu_resampler_t *rs, const float *d_in, int in_is_iq, uint32_t chunk, int nchunks, double up, double down, int nearest,
                     float *d_out, int64_t cap, int64_t *h_n)
{
    (void)nearest;
    const int64_t n = tsdrgpu_resample_count(rs, chunk, nchunks, up, down);
    if (n < 0 || n > cap) return TSDRGPU_EINVAL;
    /* touches the whole input it was given and the whole output it produces */
    const int64_t nin = (int64_t)chunk * nchunks * (in_is_iq ? 2 : 1);
    float acc = 0.0f;
    for (int64_t i = 0; i < nin; i += 64) acc += d_in[i];
    if (nin) acc += d_in[nin - 1];
    fill(d_out, n, 0.25f + 0.0f * acc);
    *h_n = n;
    rs->k = 0;
    if (rs->track_P) {
        const int64_t tot = rs->phase + n;
        rs->k = (int)(tot / rs->track_P);
        if (rs->k > 512) rs->k = -1; /* more frames than the arrays hold: the real one reports it the same way */
        rs->phase = tot % rs->track_P;
        for (int i = 0; i < rs->k; i++) { rs->mn[i] = 0.25f; rs->mx[i] = 0.25f; }
    }
    return 0;
}
int tsdrgpu_resampler_track_frames(tsdrgpu_resampler_t *rs, int64_t P, int64_t phase)
{
    if (P != 0 && P < 4096) return TSDRGPU_EINVAL;
    rs->track_P = P;
    rs->phase = phase;
    rs->k = 0;
    return 0;
}
int tsdrgpu_resampler_frame_minmax(tsdrgpu_resampler_t *rs, const float **mn, const float **mx, int *k)
{
    *mn = rs->mn; *mx = rs->mx; *k = rs->k;
    return 0;
}

/* ---- frame post-processing ------------------------------------------------------------------------------------------- */
int tsdrgpu_postproc_create(tsdrgpu_t *g, tsdrgpu_postproc_t **out)
{
    *out = (tsdrgpu_postproc_t *)calloc(1, sizeof(**out));
    if (!*out) return TSDRGPU_ENOMEM;
    (*out)->g = g;
    return 0;
}
void tsdrgpu_postproc_destroy(tsdrgpu_postproc_t *pp) { free(pp); }
int tsdrgpu_postproc_set_exact_ties(tsdrgpu_postproc_t *pp, int on) { (void)pp; (void)on; return 0; }
int tsdrgpu_postproc_run(tsdrgpu_postproc_t *pp, const float *d_frames, int F, int W, int H, const tsdrgpu_pp_params_t *prm, float *d_out,
                         tsdrgpu_pp_frameinfo_t *h_info)
{
    if (pp->open) return TSDRGPU_ESTATE;
    if (W < 1 || H < 1 || (long long)W * H > TSDRGPU_MAX_FRAME_PIXELS) return TSDRGPU_EINVAL; /* like the real one: the reference's own bound */
    {   /* STUB_FAIL_POSTPROC_AFTER=n: the n-th call fails, like a device call that fails in the middle of a session */
        static int calls;
        const char *e = getenv("STUB_FAIL_POSTPROC_AFTER");
        if (e && __atomic_add_fetch(&calls, 1, __ATOMIC_RELAXED) == atoi(e)) return TSDRGPU_EHIP;
    }
    memcpy(d_out, d_frames, sizeof(float) * (size_t)F * W * H);
    pp->F = F;
    if (h_info) {
        memset(h_info, 0, sizeof(*h_info) * (size_t)F);
        if (prm->pll) { h_info[0].pll_fired = 1; h_info[0].frameratediff = 1e-7; } /* the engine's geometry update under its lock */
    }
    return 0;
}
int tsdrgpu_postproc_begin_minmax(tsdrgpu_postproc_t *pp, const float *d_frames, int F, int W, int H, const tsdrgpu_pp_params_t *prm, const float *mn,
                                  const float *mx, float *d_out)
{
    (void)prm;
    if (pp->open) return TSDRGPU_ESTATE;
    float acc = 0.0f;
    for (int i = 0; i < F; i++) acc += mn[i] + mx[i];
    memcpy(d_out, d_frames, sizeof(float) * (size_t)F * W * H);
    d_out[0] += 0.0f * acc;
    pp->open = 1;
    pp->F = F;
    return 0;
}
int tsdrgpu_postproc_finish(tsdrgpu_postproc_t *pp, float *d_out, tsdrgpu_pp_frameinfo_t *h_info)
{
    (void)d_out;
    if (!pp->open) return TSDRGPU_ESTATE;
    pp->open = 0;
    if (h_info) memset(h_info, 0, sizeof(*h_info) * (size_t)pp->F);
    return 0;
}
int tsdrgpu_postproc_info_pack(tsdrgpu_postproc_t *pp, tsdrgpu_pp_frameinfo_t *d_info, int F)
{
    (void)pp;
    memset(d_info, 0, sizeof(*d_info) * (size_t)F);
    for (int i = 0; i < F; i++) { d_info[i].lastmin = 0.0f; d_info[i].lastmax = 1.0f; }
    return 0;
}

/* ---- frame-rate detector --------------------------------------------------------------------------------------------- */
int tsdrgpu_autocorr_create(tsdrgpu_t *g, tsdrgpu_autocorr_t **out, uint32_t fs)
{
    if (fs < 100000) return TSDRGPU_EINVAL;
    tsdrgpu_autocorr_t *ac = (tsdrgpu_autocorr_t *)calloc(1, sizeof(*ac));
    if (!ac) return TSDRGPU_ENOMEM;
    ac->g = g;
    ac->fs = fs;
    ac->capture = (uint32_t)(3.1 * fs / 55.0);
    ac->flo = (int32_t)(fs / 87.0); ac->flen = (int32_t)(fs / 55.0) - ac->flo;
    ac->llo = 16; ac->llen = 1024;
    ac->plots = (double *)calloc((size_t)ac->flen + ac->llen, sizeof(double));
    ac->snap = (double *)calloc((size_t)ac->flen + ac->llen, sizeof(double));
    ac->corr = (float *)calloc(4096, sizeof(float));
    ac->ring = 24;
    *out = ac;
    return 0;
}
void tsdrgpu_autocorr_destroy(tsdrgpu_autocorr_t *ac) { if (ac) { free(ac->plots); free(ac->snap); free(ac->corr); free(ac); } }
int tsdrgpu_autocorr_reset(tsdrgpu_autocorr_t *ac) { ac->calls = 0; ac->exact_epoch = 0; ac->replay_left = 0; return 0; }
int tsdrgpu_autocorr_geometry(tsdrgpu_autocorr_t *ac, int32_t *flo, int32_t *flen, int32_t *llo, int32_t *llen, uint32_t *capture, uint32_t *n)
{
    *flo = ac->flo; *flen = ac->flen; *llo = ac->llo; *llen = ac->llen; *capture = ac->capture;
    uint32_t p = 1;
    while (p * 2 <= ac->capture) p *= 2;
    *n = p;
    return 0;
}
int tsdrgpu_autocorr_run(tsdrgpu_autocorr_t *ac, const float *d_in, int in_is_iq, int64_t stride, int nwin, int mode)
{
    (void)mode;
    if (ac->replay_left) return TSDRGPU_ESTATE;
    if (stub_env_is("STUB_FAIL_RUN_DEVICE", ac->g->device)) return TSDRGPU_EHIP;
    for (int w = 0; w < nwin; w++) {
        const float *x = d_in + (size_t)w * stride * (in_is_iq ? 2 : 1);
        double acc = 0.0;
        for (uint32_t i = 0; i < ac->capture * (in_is_iq ? 2u : 1u); i += 256) acc += x[i];
        ac->plots[0] += acc;
        ac->plots[ac->flen + ac->llen - 1] += acc;
        ac->calls++;
    }
    return 0;
}
int tsdrgpu_autocorr_set_async(tsdrgpu_autocorr_t *ac, int on) { ac->async = on; return 0; }
int tsdrgpu_autocorr_set_exact(tsdrgpu_autocorr_t *ac, int on) { ac->exact = on; return 0; }
int tsdrgpu_autocorr_set_certify(tsdrgpu_autocorr_t *ac, int mode, size_t bytes) { (void)bytes; ac->certify = mode; return 0; }
int tsdrgpu_autocorr_lane(tsdrgpu_autocorr_t *ac) { return ac->async ? TSDRGPU_LANE_BACKGROUND : TSDRGPU_LANE_COMPUTE; }
int tsdrgpu_autocorr_retention(tsdrgpu_autocorr_t *ac, int *ring, int *ready, int *kept, int *exact)
{
    *ring = ac->ring; *ready = ac->ring; *kept = ac->exact_epoch ? 0 : (int)ac->calls; *exact = ac->exact_epoch;
    return 0;
}
int tsdrgpu_autocorr_promote_step(tsdrgpu_autocorr_t *ac, int max_windows, int *remaining)
{
    if (!ac->exact_epoch && !ac->replay_left) ac->replay_left = (int)ac->calls + 1; /* a replay begins */
    ac->replay_left -= max_windows;
    if (ac->replay_left <= 0) { ac->replay_left = 0; ac->exact_epoch = 1; }
    *remaining = ac->replay_left;
    return 0;
}
int tsdrgpu_autocorr_argmax_async(tsdrgpu_autocorr_t *ac)
{
    if (ac->replay_left) return TSDRGPU_ESTATE;
    ac->argmax_pending = 1;
    return 0;
}
int tsdrgpu_autocorr_argmax_result(tsdrgpu_autocorr_t *ac, int32_t *fi, int32_t *li)
{
    if (!ac->argmax_pending) return TSDRGPU_ESTATE;
    ac->argmax_pending = 0;
    ac->argmaxes++;
    *fi = ac->flo; *li = ac->llo;
    return 0;
}
int tsdrgpu_autocorr_certificate(tsdrgpu_autocorr_t *ac, tsdrgpu_ac_certificate_t *c)
{
    memset(c, 0, sizeof(*c));
    /* every seventh plot of a float32 epoch is "not certified": the engine's hold-back and replay path runs */
    int okay = ac->exact_epoch || (ac->argmaxes % 7) != 0;
    if (getenv("STUB_UNCERTIFIED_DEVICE")) okay = ac->exact_epoch || !stub_env_is("STUB_UNCERTIFIED_DEVICE", ac->g->device);
    c->frame_certified = c->line_certified = okay;
    c->exact_epoch = ac->exact_epoch;
    return 0;
}
int tsdrgpu_autocorr_plots_snapshot(tsdrgpu_autocorr_t *ac, const double **snap, uint64_t *calls)
{
    memcpy(ac->snap, ac->plots, sizeof(double) * ((size_t)ac->flen + ac->llen));
    *snap = ac->snap;
    *calls = ac->calls;
    return 0;
}
int tsdrgpu_autocorr_last_corr(tsdrgpu_autocorr_t *ac, const float **d_corr, uint32_t *n) { *d_corr = ac->corr; *n = 4096; return 0; }

/* ---- super-bandwidth stitch ------------------------------------------------------------------------------------------- */
int tsdrgpu_superb_stitch(tsdrgpu_t *g, float *const *d_hops, int nhops, int gathered, int sif, float *d_out, int32_t *h_off, uint32_t *h_total)
{
    (void)g; (void)sif;
    uint32_t per = 1;
    while (per * 2 <= (uint32_t)gathered) per *= 2;
    for (int i = 0; i < nhops; i++) {
        memcpy(d_out + (size_t)i * per * 2, d_hops[i], sizeof(float) * 2 * per);
        if (h_off) h_off[i] = 0;
    }
    if (h_total) *h_total = (uint32_t)nhops * per;
    return 0;
}
int tsdrgpu_superb_stitch_exact(tsdrgpu_t *g, float *const *d_hops, int nhops, int gathered, int sif, float *d_out, int32_t *h_off, uint32_t *h_total)
{
    return tsdrgpu_superb_stitch(g, d_hops, nhops, gathered, sif, d_out, h_off, h_total);
}

/* ---- what tsdr_sweep.c needs on top: the synchronous argmax, the sums' exchange between the ranks (threads of one process) ---- */
int tsdrgpu_autocorr_argmax(tsdrgpu_autocorr_t *ac, int32_t *fi, int32_t *li)
{
    if (tsdrgpu_autocorr_argmax_async(ac)) return TSDRGPU_ESTATE;
    return tsdrgpu_autocorr_argmax_result(ac, fi, li);
}
int tsdrgpu_autocorr_finalize_sums(tsdrgpu_autocorr_t *ac, uint64_t total)
{
    const size_t L = (size_t)ac->flen + ac->llen;
    for (size_t i = 0; i < L; i++) ac->plots[i] /= (double)(total ? total : 1);
    ac->calls = total;
    return 0;
}
int tsdrgpu_autocorr_promote(tsdrgpu_autocorr_t *ac)
{
    ac->exact_epoch = 1; /* (the replayed sums are the same numbers here) */
    const size_t L = (size_t)ac->flen + ac->llen;
    for (size_t i = 0; i < L; i++) ac->plots[i] *= (double)(ac->calls ? ac->calls : 1); /* back to this rank's sums */
    return 0;
}
int tsdrgpu_autocorr_plots(tsdrgpu_autocorr_t *ac, double *hf, double *hl, uint64_t *calls)
{
    memcpy(hf, ac->plots, sizeof(double) * (size_t)ac->flen);
    memcpy(hl, ac->plots + ac->flen, sizeof(double) * (size_t)ac->llen);
    if (calls) *calls = ac->calls;
    return 0;
}

/* One communicator per process is enough for the sweep tool.  Creating it is a collective like ncclCommInitRank: every rank
 * must arrive, or the ones that did wait for ever — which is what tsdr_sweep.c's meeting points are there to prevent. */
struct tsdrgpu_comm { int world, rank; };
static pthread_mutex_t comm_lock = PTHREAD_MUTEX_INITIALIZER;
static pthread_barrier_t comm_bar;
static int comm_bar_world;
static double *comm_acc;
static size_t comm_acc_n;
int tsdrgpu_rccl_unique_id(void *id128) { memset(id128, 7, TSDRGPU_RCCL_ID_BYTES); return 0; }
int tsdrgpu_comm_create(tsdrgpu_t *g, tsdrgpu_comm_t **out, int world, int rank, const void *id128)
{
    (void)g; (void)id128;
    pthread_mutex_lock(&comm_lock);
    if (comm_bar_world != world) {
        pthread_barrier_init(&comm_bar, NULL, (unsigned)world);
        comm_bar_world = world;
    }
    pthread_mutex_unlock(&comm_lock);
    *out = (tsdrgpu_comm_t *)calloc(1, sizeof(**out));
    if (!*out) return TSDRGPU_ENOMEM;
    (*out)->world = world; (*out)->rank = rank;
    pthread_barrier_wait(&comm_bar);
    return 0;
}
void tsdrgpu_comm_destroy(tsdrgpu_comm_t *c) { free(c); }
int tsdrgpu_autocorr_allreduce(tsdrgpu_autocorr_t *ac, tsdrgpu_comm_t *c, uint64_t total)
{
    const size_t L = (size_t)ac->flen + ac->llen;
    pthread_barrier_wait(&comm_bar);
    pthread_mutex_lock(&comm_lock);
    if (comm_acc_n != L) { free(comm_acc); comm_acc = (double *)calloc(L, sizeof(double)); comm_acc_n = L; }
    for (size_t i = 0; i < L; i++) comm_acc[i] += ac->plots[i];
    pthread_mutex_unlock(&comm_lock);
    pthread_barrier_wait(&comm_bar);
    for (size_t i = 0; i < L; i++) ac->plots[i] = comm_acc[i] / (double)(total ? total : 1);
    ac->calls = total;
    pthread_barrier_wait(&comm_bar);
    if (c->rank == 0) memset(comm_acc, 0, sizeof(double) * L);
    pthread_barrier_wait(&comm_bar);
    return 0;
}
int tsdrgpu_modedetect_create(tsdrgpu_modedetect_t **out) { *out = NULL; return TSDRGPU_ESTATE; } /* (the tool goes on without it) */
void tsdrgpu_modedetect_destroy(tsdrgpu_modedetect_t *d) { (void)d; }
int tsdrgpu_modedetect_feed(tsdrgpu_modedetect_t *d, int fo, int fi, int lo, int li, uint32_t rate, tsdrgpu_detection_t *out)
{
    (void)d; (void)fo; (void)fi; (void)lo; (void)li; (void)rate; (void)out;
    return TSDRGPU_ESTATE;
}
