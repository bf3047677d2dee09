/* tsdr_sweep.c — the frame-rate detector's lag sweep over a recording, sharded across the GPUs of one node, in C.
 *
 * The reference's analogue is frameratedetector_thread (TempestSDR/src/frameratedetector.c:128-187): consecutive
 * capture windows of (3.1 * fs / 55) samples are correlated (fft_autocorrelation, fft.c:49-64) and their |R| folded into
 * the frame- and line-lag plots (accummulate, frameratedetector.c:34-62); the host then takes the argmax of each plot
 * (PlotVisualizer.java:233-236) and turns the pair into frame rate and line count (Main.java:1301-1303,1346-1350).
 * Capture windows are independent, so here window k goes to device k mod G (BASELINE configs[3]):
 *
 *   one host thread per device, each with its own tsdrgpu context, detector object and RCCL rank (ncclCommInitRank from
 *   the threads of this ONE process, no launcher, no Python), its windows resident in that device's HBM;
 *   tsdrgpu_autocorr_run(mode 1: per-lag sums)  ->  tsdrgpu_autocorr_allreduce (ncclAllReduce of the L + 1 doubles over
 *   xGMI, queued on the detector's lane, then the division by the global window count)  ->  argmax + certificate on the
 *   merged plots, which are identical on every rank, so every rank decides alike  ->  if the certificate fails: every rank
 *   replays ITS windows in the reference's arithmetic (tsdrgpu_autocorr_promote) and the exact sums are exchanged again.
 *
 * With one device the exchange disappears and the windows run in the reference's own recurrence (mode 0), i.e. exactly the
 * engine's detector; --force-comm takes the sums + all-reduce path with a one-rank communicator (what a 1-GPU box can
 * exercise of it).  Input: a RawFile recording (TSDRPlugin_RawFile/src/TSDRPlugin_RawFile.c:166-179: float / int8 / uint8 /
 * int16 / uint16 interleaved IQ); narrow formats are decoded on the device (tsdrgpu_decode_samples).
 *
 * usage: tsdr_sweep <file> <samplerate> [float|int8|uint8|int16|uint16] [--devices 0,1,..] [--windows N]
 *                   [--detector certified|exact|fast] [--force-comm] [--plots out.f64]
 * prints one JSON line; exit status 0 on success.  There is no CPU path: without a HIP device it fails. */
#define _GNU_SOURCE
#include <errno.h>
#include <fcntl.h>
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include "tsdrgpu.h"

#define MAX_DEV 64

typedef struct {
    /* shared, read-only */
    const char *path;
    uint32_t rate;
    int type;          /* tsdrgpu_decode_samples numbering */
    size_t elem;       /* bytes per value */
    int world;
    int nwin;          /* windows of the whole sweep */
    int detector;      /* 0 fast, 1 certified, 2 exact */
    int use_comm;
    const unsigned char *id;
    pthread_barrier_t *bar;
    /* per rank */
    int rank, device;
    int rc;
    char err[512];
    int my_windows;
    int32_t fi, li;
    int certified, promoted;
    double r0, margin, gap_frame, gap_line;
    double ms_upload, ms_compute;
    /* rank 0 only */
    int32_t flo, flen, llo, llen;
    uint32_t capture, fft_n;
    double *plots; /* flen + llen, malloc'ed by rank 0 when asked for */
    int want_plots;
    volatile int *votes; /* one slot per rank: how the ranks agree before every step that leads into a collective (agree()) */
} rank_t;

static double now_ms(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

/* Every rank contributes flag bits; all of them get the OR.  The ranks are threads of this process, so the exchange is shared
 * memory between two barriers.  The ranks meet like this before every step that leads into a collective (ncclCommInitRank,
 * the all-reduces): a rank that failed on its own, or whose certificate alone was refused (the premise check looks at the
 * rank's OWN newest window), would otherwise enter — or skip — a collective the others do not, and they would wait for ever.
 * AGREE_FAILED is raised by a rank that has given up: everybody leaves at that meeting point. */
#define AGREE_FAILED 2
static int agree(rank_t *r, int flags)
{
    if (r->world == 1) return flags;
    r->votes[r->rank] = flags;
    pthread_barrier_wait(r->bar);
    int any = 0;
    for (int k = 0; k < r->world; k++) any |= r->votes[k];
    pthread_barrier_wait(r->bar); /* everybody has read: the slots may be written again */
    return any;
}
/* a meeting point of the healthy path: leaves through `out` when some rank has failed */
#define MEET(r_, flags_, result_)                                                                   \
    do {                                                                                            \
        (result_) = agree((r_), (flags_));                                                          \
        if ((result_) & AGREE_FAILED) {                                                             \
            snprintf((r_)->err, sizeof((r_)->err), "another rank failed");                          \
            (r_)->rc = 2;                                                                           \
            announced = 1;                                                                          \
            goto out;                                                                               \
        }                                                                                           \
    } while (0)

#define FAIL(r_, g_, what_)                                                                                    \
    do {                                                                                                       \
        snprintf((r_)->err, sizeof((r_)->err), "%s: %s", (what_), (g_) ? tsdrgpu_last_error(g_) : "no context"); \
        (r_)->rc = 1;                                                                                          \
        goto out;                                                                                              \
    } while (0)

static void *rank_main(void *arg)
{
    rank_t *r = (rank_t *)arg;
    tsdrgpu_t *g = NULL;
    tsdrgpu_autocorr_t *ac = NULL;
    tsdrgpu_comm_t *comm = NULL;
    float *d_win = NULL;
    void *d_raw = NULL, *h_buf = NULL;
    int fd = -1;
    int announced = 0, met = 0; /* announced: the others know that this rank is leaving (or it learnt of a failure from them) */

    if (tsdrgpu_create(&g, r->device)) { snprintf(r->err, sizeof(r->err), "tsdrgpu_create(device %d) failed: no usable HIP device (this tool has no CPU path)", r->device); r->rc = 1; g = NULL; goto out; }
    if (tsdrgpu_autocorr_create(g, &ac, r->rate)) FAIL(r, g, "tsdrgpu_autocorr_create");
    uint32_t capture = 0, n = 0;
    int32_t flo, flen, llo, llen;
    tsdrgpu_autocorr_geometry(ac, &flo, &flen, &llo, &llen, &capture, &n);
    if (r->detector == 2 && tsdrgpu_autocorr_set_exact(ac, 1)) FAIL(r, g, "tsdrgpu_autocorr_set_exact");
    /* certified: the windows stay where they are in this device's memory until the sweep is over (mode 2: no copy) */
    if (r->detector == 1 && tsdrgpu_autocorr_set_certify(ac, 2, 0)) FAIL(r, g, "tsdrgpu_autocorr_set_certify");
    /* ncclCommInitRank is itself a collective: only enter it once every rank has its context and its detector */
    MEET(r, 0, met);
    if (r->use_comm && tsdrgpu_comm_create(g, &comm, r->world, r->rank, r->id)) FAIL(r, g, "tsdrgpu_comm_create");

    /* this rank's windows: k = rank, rank + world, ... -> contiguous in its own buffer */
    int mine = 0;
    for (int k = r->rank; k < r->nwin; k += r->world) mine++;
    r->my_windows = mine;
    const size_t win_values = (size_t)capture * 2;
    const size_t win_bytes = win_values * r->elem;
    if (mine) {
        if (tsdrgpu_alloc(g, (void **)&d_win, (size_t)mine * win_values * sizeof(float))) FAIL(r, g, "window buffer");
        if (r->type != 0 && tsdrgpu_alloc(g, &d_raw, win_bytes)) FAIL(r, g, "raw window buffer");
        if (tsdrgpu_alloc_host(g, &h_buf, win_bytes)) FAIL(r, g, "pinned buffer");
        fd = open(r->path, O_RDONLY);
        if (fd < 0) { snprintf(r->err, sizeof(r->err), "cannot open %s: %s", r->path, strerror(errno)); r->rc = 1; goto out; }
    }
    const double t0 = now_ms();
    for (int j = 0; j < mine; j++) {
        const int k = r->rank + j * r->world;
        size_t got = 0;
        while (got < win_bytes) {
            const ssize_t x = pread(fd, (char *)h_buf + got, win_bytes - got, (off_t)((size_t)k * win_bytes + got));
            if (x <= 0) { snprintf(r->err, sizeof(r->err), "short read of window %d", k); r->rc = 1; goto out; }
            got += (size_t)x;
        }
        float *dst = d_win + (size_t)j * win_values;
        if (r->type == 0) {
            if (tsdrgpu_upload(g, dst, h_buf, win_bytes)) FAIL(r, g, "upload");
        } else {
            /* TSDRPlugin_RawFile.c:241-261 on the device, bit-exact */
            if (tsdrgpu_upload(g, d_raw, h_buf, win_bytes) || tsdrgpu_decode_samples(g, d_raw, r->type, dst, (int64_t)win_values)) FAIL(r, g, "decode");
        }
        if (tsdrgpu_sync(g)) FAIL(r, g, "sync"); /* the pinned buffer is reused */
    }
    const double t1 = now_ms();
    r->ms_upload = t1 - t0;
    MEET(r, 0, met); /* the timed part starts together */

    const double t2 = now_ms();
    const int sums = r->world > 1 || r->use_comm;
    if (mine && tsdrgpu_autocorr_run(ac, d_win, 1, (int64_t)capture, mine, sums ? 1 : 0)) FAIL(r, g, "tsdrgpu_autocorr_run");
    MEET(r, 0, met); /* nobody enters the exchange alone */
    if (sums) {
        if (comm ? tsdrgpu_autocorr_allreduce(ac, comm, (uint64_t)r->nwin) : tsdrgpu_autocorr_finalize_sums(ac, (uint64_t)r->nwin)) FAIL(r, g, "exchange");
    }
    if (tsdrgpu_autocorr_argmax(ac, &r->fi, &r->li)) FAIL(r, g, "tsdrgpu_autocorr_argmax");
    tsdrgpu_ac_certificate_t c;
    tsdrgpu_autocorr_certificate(ac, &c);
    r->certified = (c.frame_certified && c.line_certified) || c.exact_epoch;
    /* The merged plots are identical on every rank, but a certificate also folds in the premise check of the rank's OWN newest
     * window (ac_premise_check), so one rank alone may be refused: the ranks agree first, and one refusal sends all of them
     * through the replay and the second exchange. */
    MEET(r, (r->detector == 1 && !r->certified) ? 1 : 0, met);
    if (r->detector == 1 && (met & 1)) {
        if (tsdrgpu_autocorr_promote(ac)) FAIL(r, g, "tsdrgpu_autocorr_promote");
        MEET(r, 0, met);
        if (sums && (comm ? tsdrgpu_autocorr_allreduce(ac, comm, (uint64_t)r->nwin) : tsdrgpu_autocorr_finalize_sums(ac, (uint64_t)r->nwin)))
            FAIL(r, g, "second exchange");
        if (tsdrgpu_autocorr_argmax(ac, &r->fi, &r->li)) FAIL(r, g, "tsdrgpu_autocorr_argmax");
        tsdrgpu_autocorr_certificate(ac, &c);
        r->promoted = 1;
        r->certified = 1; /* the reference's own arithmetic */
    }
    r->r0 = c.r0;
    r->margin = c.margin;
    r->gap_frame = c.frame_best - c.frame_runner_up;
    r->gap_line = c.line_best - c.line_runner_up;
    r->ms_compute = now_ms() - t2;
    if (r->rank == 0) {
        r->flo = flo; r->flen = flen; r->llo = llo; r->llen = llen; r->capture = capture; r->fft_n = n;
        if (r->want_plots) {
            r->plots = (double *)malloc(sizeof(double) * ((size_t)flen + llen));
            uint64_t calls = 0;
            if (!r->plots || tsdrgpu_autocorr_plots(ac, r->plots, r->plots + flen, &calls)) FAIL(r, g, "tsdrgpu_autocorr_plots");
        }
    }
    MEET(r, 0, met); /* the last meeting point: a rank that failed behind the exchanges finds the others here */
out:
    /* a rank that failed on its own tells the others at their next meeting point, whichever that is; they all leave there.
     * (What cannot be announced: a device fault INSIDE a collective the others have already entered.) */
    if (r->rc && !announced) (void)agree(r, AGREE_FAILED);
    if (fd >= 0) close(fd);
    if (g) {
        tsdrgpu_sync(g);
        if (comm) tsdrgpu_comm_destroy(comm);
        if (ac) tsdrgpu_autocorr_destroy(ac);
        tsdrgpu_free(g, d_win);
        tsdrgpu_free(g, d_raw);
        tsdrgpu_free_host(g, h_buf);
        tsdrgpu_destroy(g);
    }
    return NULL;
}

static int parse_type(const char *s, int *type, size_t *elem)
{
    if (!strcmp(s, "float")) { *type = 0; *elem = 4; }
    else if (!strcmp(s, "int8")) { *type = 1; *elem = 1; }
    else if (!strcmp(s, "int16")) { *type = 2; *elem = 2; }
    else if (!strcmp(s, "uint8")) { *type = 3; *elem = 1; }
    else if (!strcmp(s, "uint16")) { *type = 4; *elem = 2; }
    else return 0;
    return 1;
}

int main(int argc, char **argv)
{
    if (argc < 3) {
        fprintf(stderr, "usage: %s <file> <samplerate> [float|int8|uint8|int16|uint16] [--devices 0,1,..] [--windows N] "
                        "[--detector certified|exact|fast] [--force-comm] [--plots out.f64]\n", argv[0]);
        return 2;
    }
    const char *path = argv[1];
    const uint32_t rate = (uint32_t)strtoul(argv[2], NULL, 10);
    int type = 0, detector = 1, force_comm = 0, max_windows = 0;
    size_t elem = 4;
    int devs[MAX_DEV], ndev = 0;
    const char *plots_path = NULL;
    {
        const char *env = getenv("TSDR_GPU_DEVICES"); /* the same list the tool takes with --devices */
        if (env)
            for (const char *p = env; *p && ndev < MAX_DEV;) { devs[ndev++] = (int)strtol(p, (char **)&p, 10); while (*p == ',' || *p == ' ') p++; }
    }
    for (int i = 3; i < argc; i++) {
        if (parse_type(argv[i], &type, &elem)) continue;
        if (!strcmp(argv[i], "--devices") && i + 1 < argc) {
            ndev = 0;
            for (const char *p = argv[++i]; *p && ndev < MAX_DEV;) { devs[ndev++] = (int)strtol(p, (char **)&p, 10); while (*p == ',') p++; }
        } else if (!strcmp(argv[i], "--windows") && i + 1 < argc) max_windows = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--detector") && i + 1 < argc) {
            const char *d = argv[++i];
            detector = d[0] == 'e' ? 2 : (d[0] == 'f' ? 0 : 1);
        } else if (!strcmp(argv[i], "--force-comm")) force_comm = 1;
        else if (!strcmp(argv[i], "--plots") && i + 1 < argc) plots_path = argv[++i];
        else { fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
    }
    if (!ndev) devs[ndev++] = 0;
    if (rate == 0) { fprintf(stderr, "bad sample rate\n"); return 2; }
    struct stat st;
    if (stat(path, &st)) { fprintf(stderr, "cannot stat %s: %s\n", path, strerror(errno)); return 1; }
    /* frameratedetector.c:160: the capture size */
    const uint32_t capture = (uint32_t)(3.1 * rate / (double)(55));
    const size_t win_bytes = (size_t)capture * 2 * elem;
    int nwin = (int)((size_t)st.st_size / win_bytes);
    if (max_windows > 0 && nwin > max_windows) nwin = max_windows;
    if (nwin < 1) { fprintf(stderr, "the recording holds less than one capture window of %u samples\n", capture); return 1; }

    const int use_comm = ndev > 1 || force_comm;
    unsigned char id[TSDRGPU_RCCL_ID_BYTES];
    memset(id, 0, sizeof(id));
    if (use_comm && tsdrgpu_rccl_unique_id(id)) { fprintf(stderr, "RCCL is not available (librccl.so.1)\n"); return 1; }
    pthread_barrier_t bar;
    pthread_barrier_init(&bar, NULL, (unsigned)ndev);
    static rank_t ranks[MAX_DEV];
    static volatile int votes[MAX_DEV];
    pthread_t th[MAX_DEV];
    for (int r = 0; r < ndev; r++) {
        rank_t *k = &ranks[r];
        memset(k, 0, sizeof(*k));
        k->path = path; k->rate = rate; k->type = type; k->elem = elem; k->world = ndev; k->nwin = nwin; k->detector = detector;
        k->use_comm = use_comm; k->id = id; k->bar = &bar; k->votes = votes; k->rank = r; k->device = devs[r]; k->want_plots = plots_path != NULL;
        pthread_create(&th[r], NULL, rank_main, k);
    }
    for (int r = 0; r < ndev; r++) pthread_join(th[r], NULL);
    pthread_barrier_destroy(&bar);
    for (int pass = 1; pass <= 2; pass++) /* the rank that failed first (rc 1) before those that left because of it (rc 2) */
        for (int r = 0; r < ndev; r++)
            if (ranks[r].rc == pass) { fprintf(stderr, "rank %d (device %d): %s\n", r, ranks[r].device, ranks[r].err); return 1; }
    /* every rank holds the same merged plots: they must agree on the argmax and on what they did about the certificate */
    for (int r = 1; r < ndev; r++)
        if (ranks[r].fi != ranks[0].fi || ranks[r].li != ranks[0].li || ranks[r].promoted != ranks[0].promoted) {
            fprintf(stderr, "ranks disagree: rank %d has (%d, %d, promoted %d), rank 0 (%d, %d, promoted %d)\n", r, ranks[r].fi, ranks[r].li,
                    ranks[r].promoted, ranks[0].fi, ranks[0].li, ranks[0].promoted);
            return 1;
        }
    const rank_t *z = &ranks[0];
    if (plots_path) {
        FILE *f = fopen(plots_path, "wb");
        if (!f || fwrite(z->plots, sizeof(double), (size_t)z->flen + z->llen, f) != (size_t)z->flen + z->llen) { fprintf(stderr, "cannot write %s\n", plots_path); return 1; }
        fclose(f);
    }
    /* the host's half: lags -> rates (Main.java:1301-1303,1346-1350), through the library's restatement of it */
    tsdrgpu_modedetect_t *md = NULL;
    tsdrgpu_detection_t det;
    memset(&det, 0, sizeof(det));
    if (tsdrgpu_modedetect_create(&md) == 0) {
        tsdrgpu_modedetect_feed(md, z->flo, z->fi, z->llo, z->li, rate, &det);
        tsdrgpu_modedetect_destroy(md);
    }
    double ms = 0, ms_up = 0;
    for (int r = 0; r < ndev; r++) { if (ranks[r].ms_compute > ms) ms = ranks[r].ms_compute; if (ranks[r].ms_upload > ms_up) ms_up = ranks[r].ms_upload; }
    printf("{\"file\": \"%s\", \"samplerate\": %u, \"capture\": %u, \"fft_n\": %u, \"windows\": %d, \"devices\": [", path, rate, z->capture, z->fft_n, nwin);
    for (int r = 0; r < ndev; r++) printf("%s%d", r ? ", " : "", devs[r]);
    printf("], \"windows_per_device\": [");
    for (int r = 0; r < ndev; r++) printf("%s%d", r ? ", " : "", ranks[r].my_windows);
    printf("], \"detector\": \"%s\", \"exchange\": \"%s\", \"frame_idx\": %d, \"line_idx\": %d, \"frame_lag\": %d, \"line_lag\": %d, "
           "\"framerate\": %.9g, \"linerate\": %.9g, \"height\": %d, \"certified\": %d, \"epoch_replayed_exact\": %d, "
           "\"r0\": %.17g, \"margin\": %.17g, \"gap_frame\": %.17g, \"gap_line\": %.17g, \"closest_mode\": \"%s\", "
           "\"ms_sweep\": %.3f, \"ms_load\": %.3f, \"windows_per_s\": %.1f}\n",
           detector == 2 ? "exact" : (detector == 1 ? "certified" : "fast"),
           use_comm ? "ncclAllReduce(f64 sum) of the per-lag sums, queued by tsdrgpu_autocorr_allreduce on the detector's lane" : "none (one device: the reference's running mean)",
           z->fi, z->li, det.frame_lag, det.line_lag, det.framerate, det.linerate, det.height, z->certified, z->promoted, z->r0, z->margin, z->gap_frame,
           z->gap_line, det.mode_id >= 0 ? det.mode_name : "", ms, ms_up, ms > 0 ? nwin / (ms * 1e-3) : 0.0);
    return 0;
}
