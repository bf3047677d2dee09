/*
 * tsdr_oracle.c — CPU restatement of the TempestSDR DSP hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (tempestsdr_amd/, include/,
 * libTSDRLibrary) may link, import or execute this file; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only as
 * the checker / reported CPU baseline.
 *
 * Every function states the reference file:line (relative to
 * /root/reference/TempestSDR/src/) whose arithmetic it restates.  The
 * arithmetic (operand types, evaluation order, rounding points) is kept
 * identical to the reference so that results are bit-comparable; structure,
 * naming and state handling are this repository's own.
 *
 * Parity pinning: the reference has no tests / golden vectors,
 * so this file is pinned against the reference ITSELF, compiled from
 * /root/reference into oracle/_ref/ by oracle/Makefile (tests/test_oracle_vs_ref.py,
 * bit-exact) and against golden vectors produced by that build
 * (tests/golden/, script tests/golden/make_golden.py).
 * orc_frame_to_rgb is pinned against the reference's JNI shim itself, compiled from
 * JavaGUI/jni/TSDRLibraryNDK.c behind a stub jni.h (oracle/jni_stub, oracle/ref_shim_jni.c;
 * tests/test_oracle_vs_ref.py::test_frame_to_rgb_*).
 * The restatements of JAVA code — orc_plot_populate / orc_plotscale_default
 * (PlotVisualizer.populateData, ZoomableXScale) — and the library's mode-detection logic
 * (Main.java / VideoMode.java) are pinned against the reference's own compiled classes: the image
 * has no JVM, so the released jar (Release/JavaGUI/JTempestSDR.jar) is executed by a bytecode
 * interpreter of ours (tests/golden/minijvm.py, driven by tests/golden/make_java_fixtures_jvm.py);
 * what the classes computed is committed as tests/golden/java_fixtures_jvm.json and checked
 * exactly by tests/test_extras_cpu.py (which also re-runs the interpreter on fresh cases where the
 * jar is present).  A JDK library call made by those methods (Math, boxing, HashMap) is the
 * interpreter's native, not the JDK's: that much of the pin is ours.
 *
 * Build: gcc -O3 -fPIC -shared -ffp-contract=off (no fast-math; the reference
 * is built -O3 without fast-math, TempestSDR/makefile:21).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* sentinel pixel values, include/TSDRLibrary.h:20-24 */
#define ORC_PIXEL_G 512.0f
#define ORC_PIXEL_B 1024.0f

/* ------------------------------------------------------------------------ */
/* A.1  AM demodulation — TSDRLibrary.c:244-262 (== fft.c:24-32)             */
/* ------------------------------------------------------------------------ */
ORC_API void orc_am_demod(const float *iq, float *out, int64_t n)
{
    /* out may alias iq (the reference works in place): out[i] is written
       after iq[2i], iq[2i+1] were read and i <= 2i. */
    for (int64_t i = 0; i < n; i++) {
        const float re = iq[2 * i];
        const float im = iq[2 * i + 1];
        out[i] = sqrtf(re * re + im * im);
    }
}

/* ------------------------------------------------------------------------ */
/* A.2  Fractional area resampler — dsp.c:250-307                            */
/* ------------------------------------------------------------------------ */
typedef struct {
    double contrib; /* unfinished-pixel accumulator carried across calls */
    double offset;  /* phase carried across calls (in input samples, <= 0) */
} orc_resample_t;

ORC_API void orc_resample_init(orc_resample_t *st)
{
    st->contrib = 0.0;
    st->offset = 0.0;
}

/* Number of output pixels the reference announces for a chunk (dsp.c:262). */
ORC_API uint32_t orc_resample_count(const orc_resample_t *st, uint32_t size,
                                    double up, double down)
{
    const double r = up / down;
    return (uint32_t)(int)(((double)size - st->offset) * r);
}

/* Returns the announced output count; *emitted receives how many pixels the
   loop really stored (they differ only in the aligned edge case).
   `out` must have room for count+2 floats. */
ORC_API uint32_t orc_resample_process(orc_resample_t *st, const float *in,
                                      uint32_t size, float *out, double up,
                                      double down, int nearest,
                                      uint32_t *emitted)
{
    const double r = up / down;    /* output pixels per input sample */
    const double rinv = down / up; /* input samples per output pixel */
    const uint32_t count = (uint32_t)(int)(((double)size - st->offset) * r);
    const double o = -st->offset * r;
    uint32_t w = 0;

    if (nearest) {
        /* dsp.c:274-276 */
        for (uint32_t p = 0; p < count; p++)
            out[w++] = in[((uint64_t)size * p) / count];
    } else {
        /* dsp.c:279-303: sample `id` covers [lo, hi) on the pixel axis */
        uint32_t pix = 0;
        for (uint32_t id = 0; id < size; id++) {
            const double lo = id * r + o;
            const double hi = lo + r;
            const double hi_m1 = lo + r - 1.0;
            const double v = in[id];

            if (pix < lo && pix < hi_m1) {
                /* pixel started in an earlier sample and ends inside this one */
                out[w++] = (float)(st->contrib + v * (1.0 - lo + pix));
                st->contrib = 0;
                pix++;
            }
            while (pix < hi_m1) { /* pixels wholly inside this sample */
                out[w++] = (float)v;
                pix++;
            }
            if (pix < hi && pix > lo)
                st->contrib += (hi - pix) * v; /* tail that spills into pixel `pix` */
            else
                st->contrib += r * v; /* whole sample inside one pixel */
        }
    }
    if (emitted) *emitted = w;
    st->offset += count * rinv - size; /* dsp.c:306 */
    return count;
}

/* ------------------------------------------------------------------------ */
/* A.6  Temporal IIR ("motion blur") — dsp.c:22-33                           */
/* ------------------------------------------------------------------------ */
ORC_API void orc_timelowpass_run(float a, int n, const float *in, float *screen)
{
    const double one_minus_a = 1.0 - a; /* double, dsp.c:29 */
    for (int i = 0; i < n; i++)
        screen[i] = screen[i] * a + in[i] * one_minus_a; /* f32*f32 + f32*f64 -> f64 -> f32 */
}

/* ------------------------------------------------------------------------ */
/* A.3  Autogain — dsp.c:35-94                                               */
/* ------------------------------------------------------------------------ */
typedef struct {
    float lastmax;
    float lastmin;
    float snr;
} orc_autogain_t;

ORC_API void orc_autogain_init(orc_autogain_t *ag)
{
    ag->lastmax = 0;
    ag->lastmin = 0;
    ag->snr = 1.0;
}

ORC_API void orc_autogain_run(orc_autogain_t *ag, int n, const float *in,
                              float *out, float norm)
{
    float lo = in[0]; /* taken before the sentinel test, dsp.c:50-51 */
    float hi = lo;
    double sum = 0.0;
    for (int i = 0; i < n; i++) {
        const float v = in[i];
        if (v > 250.0 || v < -250) continue; /* sentinel colours skipped */
        if (v > hi) hi = v; else if (v < lo) lo = v;
        sum += v;
    }
    const float keep = 1.0f - norm;
    ag->lastmax = keep * ag->lastmax + norm * hi;
    ag->lastmin = keep * ag->lastmin + norm * lo;
    const float span = (ag->lastmax == ag->lastmin) ? 1.0f : (ag->lastmax - ag->lastmin);

    const double mean = sum / (double)n;
    double s2 = 0.0, s1 = 0.0;
    for (int i = 0; i < n; i++) {
        const float v = in[i];
        out[i] = (v > 250.0 || v < -250) ? v : ((in[i] - ag->lastmin) / span);
        const double d = v - mean;
        s2 += d * d;
        s1 += d;
    }
    const double stdev = sqrt((s2 - s1 * s1 / (double)n) / (double)(n - 1));
    ag->snr = mean / stdev;
}

/* ------------------------------------------------------------------------ */
/* A.4  Row / column collapse — dsp.c:96-110                                 */
/* ------------------------------------------------------------------------ */
ORC_API void orc_average_v_h(int width, int height, const float *frame,
                             float *colsum, float *rowsum)
{
    for (int x = 0; x < width; x++) colsum[x] = 0.0f;
    for (int y = 0; y < height; y++) rowsum[y] = 0.0f;
    const int total = width * height;
    for (int i = 0; i < total; i++) { /* f32 accumulation in raster order */
        const float v = frame[i];
        colsum[i % width] += v;
        rowsum[i / width] += v;
    }
}

/* ------------------------------------------------------------------------ */
/* A.5  Sync detector — gaussian.c:18-79, syncdetector.c:26-225              */
/* ------------------------------------------------------------------------ */

/* The five normalised taps, gaussian.c:17-28.  CALC_GAUSSCOEFF(5,i) expands
   textually to expf(-2.0f*1.0f*1.0f*i*i/(5*5)). */
ORC_API void orc_gaussian_taps(float taps[5])
{
    const float e2 = expf(-2.0f * 1.0f * 1.0f * -2 * -2 / (5 * 5));
    const float e1 = expf(-2.0f * 1.0f * 1.0f * -1 * -1 / (5 * 5));
    const float e0 = expf(-2.0f * 1.0f * 1.0f * 0 * 0 / (5 * 5));
    const float f1 = expf(-2.0f * 1.0f * 1.0f * 1 * 1 / (5 * 5));
    const float f2 = expf(-2.0f * 1.0f * 1.0f * 2 * 2 / (5 * 5));
    const float norm = e2 + e1 + e0 + f1 + f2;
    taps[0] = e2 / norm;
    taps[1] = e1 / norm;
    taps[2] = e0 / norm;
    taps[3] = f1 / norm;
    taps[4] = f2 / norm;
}

/* Circular 5-tap blur.  The reference works in place with a 5-value
   look-ahead (gaussian.c:30-78); for n >= 2 that equals
   out[(i+2) mod n] = sum_k taps[k]*orig[(i+k) mod n], summed left to right in f32. */
ORC_API void orc_gaussianblur(float *data, int n)
{
    float taps[5];
    orc_gaussian_taps(taps);
    float *orig = (float *)malloc(sizeof(float) * (size_t)n);
    memcpy(orig, data, sizeof(float) * (size_t)n);
    for (int i = 0; i < n; i++) {
        const float a = orig[i % n], b = orig[(i + 1) % n], c = orig[(i + 2) % n];
        const float d = orig[(i + 3) % n], e = orig[(i + 4) % n];
        data[(i + 2) % n] = a * taps[0] + b * taps[1] + c * taps[2] + d * taps[3] + e * taps[4];
    }
    free(orig);
}

typedef struct {
    int dx;
    int vx;
    int absvx;
    int curr_stripsize;
} orc_sweetspot_t;

/* syncdetector.c:26-58.  NB the `total` parameter is narrowed to float. */
static void orc_findbestfit(const float *data, int n, float total, int strip,
                            double *best, int *best_at)
{
    const double rest_len = n - strip;
    const double strip_len = strip;
    double run = 0.0;
    for (int i = 0; i < strip; i++) run += data[i];

    const double d0 = (total - run) / rest_len - run / strip_len;
    *best = d0 * d0;
    *best_at = 0;

    const int wrap_from = n - strip;
    for (int i = 0; i < n - 1; i++) {
        const double leaving = data[i];
        const int enter_at = (i < wrap_from) ? (i + strip) : (i - wrap_from);
        const double entering = data[enter_at];
        run = run - leaving + entering;
        const double d = (total - run) / rest_len - run / strip_len;
        const double fit = d * d;
        if (fit > *best) {
            *best = fit;
            *best_at = i; /* index just removed, kept literally (syncdetector.c:55) */
        }
    }
}

/* syncdetector.c:71-119.  Blurs `data` in place and marks two entries. */
ORC_API void orc_findthesweetspot(orc_sweetspot_t *db, float *data, int n,
                                  int minsize, double lowpass)
{
    if (minsize < 1) minsize = 1;
    const int half = n >> 1;

    if (db->curr_stripsize < minsize) db->curr_stripsize = minsize;
    else if (db->curr_stripsize > half) db->curr_stripsize = half;

    orc_gaussianblur(data, n);

    double total = 0.0;
    for (int i = 0; i < n; i++) total += data[i];

    int best_size = db->curr_stripsize;
    int best_start;
    double best_fit;
    orc_findbestfit(data, n, (float)total, db->curr_stripsize, &best_fit, &best_start);

    const int trial[4] = { db->curr_stripsize - 4, db->curr_stripsize + 4,
                           db->curr_stripsize >> 1, db->curr_stripsize << 1 };
    for (int t = 0; t < 4; t++) {
        const int s = trial[t];
        if (s >= minsize && s < half && s != db->curr_stripsize) {
            double fit;
            int at;
            orc_findbestfit(data, n, (float)total, s, &fit, &at);
            if (fit > best_fit) {
                best_fit = fit;
                best_start = at;
                best_size = s;
            }
        }
    }
    db->curr_stripsize = best_size;

    data[best_start] = ORC_PIXEL_B;
    data[(best_start + best_size) % n] = ORC_PIXEL_B;

    const int h2 = n / 2;
    int centre = (best_start + best_size / 2) % n;
    const int rawdiff = centre - db->dx;
    if (rawdiff > h2)
        db->dx += n;
    else if (rawdiff < -h2)
        centre += n;

    const int last = db->dx;
    db->dx = (int)(((int64_t)round(centre * lowpass + (1.0 - lowpass) * db->dx)) % ((int64_t)n));

    const int rawvx = db->dx - last;
    db->vx = (rawvx > h2) ? (n - rawvx) : ((rawvx < -h2) ? (-n - rawvx) : rawvx);
    db->absvx = (db->vx >= 0) ? db->vx : -db->vx;
}

/* Geometry derivation — TSDRLibrary.c:540-550 */
typedef struct {
    uint32_t samplerate;
    int width;
    int height;
    double refreshrate;
    double pixelrate;
    double pixeltimeoversampletime;
} orc_geometry_t;

ORC_API void orc_set_internal_samplerate(orc_geometry_t *g, uint32_t samplerate)
{
    g->samplerate = samplerate;
    const double real_width = samplerate / (g->refreshrate * g->height);
    g->width = (int)2 * real_width;
    g->pixelrate = g->width * g->height * g->refreshrate;
    if (g->samplerate != 0 && g->pixelrate != 0)
        g->pixeltimeoversampletime = ((double)g->samplerate) / g->pixelrate;
}

typedef struct {
    orc_sweetspot_t db_x;
    orc_sweetspot_t db_y;
    double last_frame_diff;
    int state; /* 0 not locked, 1 locked */
    double avg_speed;
} orc_syncdetector_t;

ORC_API void orc_syncdetector_init(orc_syncdetector_t *sy)
{
    memset(sy, 0, sizeof(*sy));
}

/* syncdetector.c:133-153.  Returns 1 when the refresh rate was nudged. */
ORC_API int orc_frameratepll(orc_syncdetector_t *sy, orc_geometry_t *g, int pll_enabled)
{
    sy->avg_speed = sy->avg_speed * 0.99 + 0.01 * sy->db_x.vx;
    if (sy->avg_speed < 0.5 && sy->avg_speed > -0.5)
        sy->state = 1;
    else
        sy->state = 0;

    if (pll_enabled && sy->db_x.vx != 0) {
        double diff;
        if (sy->state == 0)
            diff = sy->db_x.vx * 0.00001;
        else
            diff = sy->avg_speed * 0.000001;
        g->refreshrate -= diff;
        orc_set_internal_samplerate(g, g->samplerate);
        return 1;
    }
    return 0;
}

/* syncdetector.c:171-225.  Returns which buffer holds the result:
   0 = `data` (possibly with lines painted in place), 1 = `outdata`. */
ORC_API int orc_syncdetector_run(orc_syncdetector_t *sy, orc_geometry_t *g,
                                 int autoshift, int pll_enabled, float *data,
                                 float *outdata, int width, int height,
                                 float *colsum, float *rowsum, int greenlines,
                                 int modify_data_allowed, int *pll_fired)
{
    orc_findthesweetspot(&sy->db_x, colsum, width, (int)(width * 0.05f), 0.9);
    orc_findthesweetspot(&sy->db_y, rowsum, height, (int)(height * 0.01f), 0.1);

    const int fired = orc_frameratepll(sy, g, pll_enabled);
    if (pll_fired) *pll_fired = fired;

    const int dx = sy->db_x.dx, dy = sy->db_y.dx;
    if (autoshift) {
        /* 2-D circular roll: out[y][x] = in[(y+dy) mod H][(x+dx) mod W]
           (syncdetector.c:187-207, written there as row-wise memcpy pairs) */
        for (int y = 0; y < height; y++) {
            const int sy_ = (y + dy) % height;
            const float *src = data + (size_t)sy_ * width;
            float *dst = outdata + (size_t)y * width;
            memcpy(dst, src + dx, sizeof(float) * (size_t)(width - dx));
            memcpy(dst + (width - dx), src, sizeof(float) * (size_t)dx);
        }
        return 1;
    }
    if (greenlines) {
        float *t = data;
        if (!modify_data_allowed) {
            memcpy(outdata, data, sizeof(float) * (size_t)width * height);
            t = outdata;
        }
        for (int y = 0; y < height; y++) t[dx + (size_t)width * y] = ORC_PIXEL_G;
        for (int x = 0; x < width; x++) t[x + (size_t)width * dy] = ORC_PIXEL_G;
        return modify_data_allowed ? 0 : 1;
    }
    return 0;
}

/* ------------------------------------------------------------------------ */
/* a8  Post-processing orchestration — dsp.c:134-239                         */
/* ------------------------------------------------------------------------ */
typedef struct {
    float *screen, *send, *corrected;
    float *colsum, *rowsum;
    int n, width, height, cap;
    int runs;
    orc_autogain_t ag;
    orc_syncdetector_t sync;
    int lowpass_before_sync;
    /* last call's observable side effects */
    int pll_fired;
    int autogain_reported;
} orc_postprocess_t;

ORC_API orc_postprocess_t *orc_postprocess_new(void)
{
    orc_postprocess_t *pp = (orc_postprocess_t *)calloc(1, sizeof(*pp));
    orc_autogain_init(&pp->ag);
    orc_syncdetector_init(&pp->sync);
    return pp;
}

ORC_API void orc_postprocess_free(orc_postprocess_t *pp)
{
    if (!pp) return;
    free(pp->screen); free(pp->send); free(pp->corrected);
    free(pp->colsum); free(pp->rowsum);
    free(pp);
}

/* params: autoshift, pll, superresolution = params_int[0],[1],[4].
   Returns a pointer into pp's buffers (or `frame` itself is never returned:
   as in the reference the result always lives in a pp buffer). */
ORC_API float *orc_post_process(orc_postprocess_t *pp, orc_geometry_t *g,
                                float *frame, int width, int height,
                                float motionblur, float lowpasscoeff,
                                int lowpass_before_sync, int autogain_after_proc,
                                int autoshift, int pll_enabled, int superres)
{
    if (height != pp->height || width != pp->width) {
        const int oldw = pp->width, oldh = pp->height;
        pp->height = height;
        pp->width = width;
        pp->n = width * height;
        if (pp->n > pp->cap) {
            pp->cap = pp->n;
            pp->screen = (float *)realloc(pp->screen, sizeof(float) * (size_t)pp->cap);
            pp->send = (float *)realloc(pp->send, sizeof(float) * (size_t)pp->cap);
            pp->corrected = (float *)realloc(pp->corrected, sizeof(float) * (size_t)pp->cap);
            for (int i = 0; i < pp->cap; i++) pp->screen[i] = 0.0f; /* dsp.c:167 */
        }
        if (width != oldw) pp->colsum = (float *)realloc(pp->colsum, sizeof(float) * (size_t)width);
        if (height != oldh) pp->rowsum = (float *)realloc(pp->rowsum, sizeof(float) * (size_t)height);
    }
    if (pp->lowpass_before_sync != lowpass_before_sync) { /* dsp.c:178-186 */
        pp->lowpass_before_sync = lowpass_before_sync;
        for (int i = 0; i < pp->n; i++) {
            pp->screen[i] = 0.0f;
            pp->send[i] = 0.0f;
            pp->corrected[i] = 0.0f;
        }
    }

    float *input = frame;
    if (!autogain_after_proc) {
        orc_autogain_run(&pp->ag, pp->n, input, pp->send, lowpasscoeff);
        input = pp->send;
    }

    float *result;
    int fired = 0;
    if (lowpass_before_sync) {
        orc_timelowpass_run(motionblur, pp->n, input, pp->screen);
        orc_average_v_h(width, height, pp->screen, pp->colsum, pp->rowsum);
        const int which = orc_syncdetector_run(&pp->sync, g, autoshift, pll_enabled,
                                               pp->screen, pp->corrected, width, height,
                                               pp->colsum, pp->rowsum, !superres, 0, &fired);
        float *syncresult = which ? pp->corrected : pp->screen;
        if (autogain_after_proc) {
            orc_autogain_run(&pp->ag, pp->n, syncresult, pp->send, lowpasscoeff);
            result = pp->send;
        } else
            result = syncresult;
    } else {
        orc_average_v_h(width, height, input, pp->colsum, pp->rowsum);
        const int which = orc_syncdetector_run(&pp->sync, g, autoshift, pll_enabled, input,
                                               pp->corrected, width, height, pp->colsum,
                                               pp->rowsum, (motionblur == 0.0f) && !superres,
                                               1, &fired);
        float *syncresult = which ? pp->corrected : input;
        orc_timelowpass_run(motionblur, pp->n, syncresult, pp->screen);
        if (autogain_after_proc) {
            orc_autogain_run(&pp->ag, pp->n, pp->screen, pp->send, lowpasscoeff);
            result = pp->send;
        } else
            result = pp->screen;
    }
    pp->pll_fired = fired;
    pp->autogain_reported = 0;
    if (pp->runs++ > 5) { /* dsp.c:231-235 */
        pp->runs = 0;
        pp->autogain_reported = 1;
    }
    return result;
}

/* read-only accessors so Python does not need the struct layout */
ORC_API void orc_postprocess_state(const orc_postprocess_t *pp, int32_t out_i[10], double out_d[4])
{
    out_i[0] = pp->sync.db_x.dx; out_i[1] = pp->sync.db_x.vx;
    out_i[2] = pp->sync.db_x.curr_stripsize;
    out_i[3] = pp->sync.db_y.dx; out_i[4] = pp->sync.db_y.vx;
    out_i[5] = pp->sync.db_y.curr_stripsize;
    out_i[6] = pp->sync.state; out_i[7] = pp->pll_fired;
    out_i[8] = pp->autogain_reported; out_i[9] = pp->runs;
    out_d[0] = pp->ag.lastmin; out_d[1] = pp->ag.lastmax;
    out_d[2] = pp->ag.snr; out_d[3] = pp->sync.avg_speed;
}
ORC_API const float *orc_postprocess_colsum(const orc_postprocess_t *pp) { return pp->colsum; }
ORC_API const float *orc_postprocess_rowsum(const orc_postprocess_t *pp) { return pp->rowsum; }

/* ------------------------------------------------------------------------ */
/* A.7  FFT and autocorrelation — fft.c:5-176, frameratedetector.c:34-62     */
/* ------------------------------------------------------------------------ */
ORC_API uint32_t orc_fft_getrealsize(uint32_t size) /* fft.c:5-11 */
{
    uint32_t m = 0;
    while ((size /= 2) != 0) m++;
    return 1u << m;
}

/* In-place radix-2 DIT on interleaved complex f32; butterflies and the
   twiddle recurrence in f64, storage f32; forward scaled by 1/N (fft.c:96-176). */
ORC_API void orc_fft_perform(float *z, uint32_t size, int inverse)
{
    int m = 0;
    while ((size /= 2) != 0) m++;
    const int64_t n = (int64_t)1 << m;

    /* bit-reversal permutation (fft.c:106-130) */
    int64_t j = 0;
    for (int64_t i = 0; i < n - 1; i++) {
        if (i < j) {
            const float tr = z[2 * i], ti = z[2 * i + 1];
            z[2 * i] = z[2 * j];
            z[2 * i + 1] = z[2 * j + 1];
            z[2 * j] = tr;
            z[2 * j + 1] = ti;
        }
        int64_t k = n >> 1;
        while (k <= j) {
            j -= k;
            k >>= 1;
        }
        j += k;
    }

    /* log2(n) butterfly stages (fft.c:132-165) */
    double wr = -1.0, wi = 0.0; /* stage root of unity, advanced by half-angle */
    int64_t span = 1;
    for (int s = 0; s < m; s++) {
        const int64_t half = span;
        span <<= 1;
        double ur = 1.0, ui = 0.0;
        for (int64_t q = 0; q < half; q++) {
            for (int64_t a = q; a < n; a += span) {
                const int64_t b = a + half;
                const double tr = ur * z[2 * b] - ui * z[2 * b + 1];
                const double ti = ur * z[2 * b + 1] + ui * z[2 * b];
                z[2 * b] = (float)(z[2 * a] - tr);
                z[2 * b + 1] = (float)(z[2 * a + 1] - ti);
                z[2 * a] = (float)(z[2 * a] + tr);
                z[2 * a + 1] = (float)(z[2 * a + 1] + ti);
            }
            const double nr = ur * wr - ui * wi;
            ui = ur * wi + ui * wr;
            ur = nr;
        }
        wi = sqrt((1.0 - wr) / 2.0);
        if (!inverse) wi = -wi;
        wr = sqrt((1.0 + wr) / 2.0);
    }

    if (!inverse) {
        const float nf = (float)(uint32_t)n;
        for (int64_t i = 0; i < n; i++) {
            z[2 * i] /= nf;
            z[2 * i + 1] /= nf;
        }
    }
}

/* fft.c:49-64.  `answer` has 2*size floats; only the first
   2*orc_fft_getrealsize(size) are transformed, the magnitude step covers all
   `size` complex entries. */
ORC_API void orc_fft_autocorrelation(float *answer, const float *real, uint32_t size)
{
    for (uint32_t i = 0; i < size; i++) {
        answer[2 * i] = real[i];
        answer[2 * i + 1] = 0.0f;
    }
    const uint32_t n = orc_fft_getrealsize(size);
    orc_fft_perform(answer, n, 0);
    for (uint32_t i = 0; i < size; i++) { /* fft.c:34-45 */
        const float re = answer[2 * i], im = answer[2 * i + 1];
        answer[2 * i] = sqrtf(re * re + im * im);
        answer[2 * i + 1] = 0;
    }
    orc_fft_perform(answer, n, 1);
}

/* frameratedetector.c:34-62: running mean of |R| over windows.  `calls` is
   the 1-based count of windows since the last reset (extbuffer.c:68-81). */
ORC_API void orc_accumulate(double *acc, const float *corr, int start, int length,
                            uint64_t calls)
{
    const float *p = corr + (size_t)start * 2;
    if (calls == 0) {
        for (int i = 0; i < length; i++) {
            const double re = p[2 * i], im = p[2 * i + 1];
            acc[i] = sqrt(re * re + im * im);
        }
    } else {
        const double c = (double)calls;
        const double cm1 = (double)(calls - 1);
        for (int i = 0; i < length; i++) {
            const double re = p[2 * i], im = p[2 * i + 1];
            const double now = sqrt(re * re + im * im);
            acc[i] = (acc[i] * cm1 + now) / c;
        }
    }
}

/* Lag windows — frameratedetector.c:20-24,91-95 */
ORC_API void orc_lag_windows(uint32_t samplerate, int32_t w[4])
{
    const int maxlength = samplerate / (double)(55);
    const int minlength = samplerate / (double)(87);
    const int height_maxlength = samplerate / (double)(590 * 55);
    const int height_minlength = samplerate / (double)(1500 * 87);
    w[0] = minlength;
    w[1] = maxlength - minlength;
    w[2] = height_minlength;
    w[3] = height_maxlength - height_minlength;
}

/* capture size — frameratedetector.c:160 */
ORC_API uint32_t orc_capture_size(uint32_t samplerate)
{
    return (uint32_t)(3.1 * samplerate / (double)(55));
}

/* ------------------------------------------------------------------------ */
/* A.8  Super-bandwidth numerics — fft.c:69-93, superbandwidth.c:67-152      */
/* ------------------------------------------------------------------------ */
ORC_API void orc_fft_crosscorrelation(float *a, float *b, uint32_t samples)
{
    const uint32_t n = orc_fft_getrealsize(samples);
    orc_fft_perform(a, n, 0);
    orc_fft_perform(b, n, 0);
    for (uint32_t i = 0; i < n; i++) {
        const float ar = a[2 * i], ai = a[2 * i + 1];
        const float br = b[2 * i], bi = b[2 * i + 1];
        a[2 * i] = ar * br + ai * bi;
        a[2 * i + 1] = ar * bi - ai * br;
    }
    orc_fft_perform(a, n, 1);
}

/* superbandwidth.c:67-81; `nfloats` counts floats (2 per sample).  Note the
   seed is |z0|^2, not |z0| (kept literally). */
ORC_API void orc_complex_to_abs_diff(float *z, int nfloats)
{
    float prev = z[0] * z[0] + z[1] * z[1];
    for (int i = 0; i < nfloats; i += 2) {
        const float re = z[i], im = z[i + 1];
        const float cur = sqrtf(re * re + im * im);
        const float d = cur - prev;
        prev = cur;
        z[i] = d;
        z[i + 1] = 0;
    }
}

/* superbandwidth.c:83-119.  nfloats = floats in each hop buffer.  Returns the
   best offset in floats (2*lag). */
ORC_API int orc_superb_bestfit(const float *hop0, const float *hopi, int nfloats,
                               int samples_in_frame)
{
    int size = (nfloats / samples_in_frame) * samples_in_frame;
    size = (int)orc_fft_getrealsize((uint32_t)size);
    const int samples = size / 2;
    float *a = (float *)malloc(sizeof(float) * (size_t)size);
    float *b = (float *)malloc(sizeof(float) * (size_t)size);
    memcpy(a, hop0, sizeof(float) * (size_t)size);
    memcpy(b, hopi, sizeof(float) * (size_t)size);
    orc_complex_to_abs_diff(a, size);
    orc_complex_to_abs_diff(b, size);
    orc_fft_crosscorrelation(a, b, (uint32_t)samples);
    int best = 0;
    float bestval = 0;
    for (int i = 0; i < samples; i++) {
        const float re = a[2 * i], im = a[2 * i + 1];
        const float v = sqrtf(re * re + im * im);
        if (i == 0)
            bestval = v;
        else if (v > bestval) {
            bestval = v;
            best = i;
        }
    }
    free(a);
    free(b);
    return 2 * best;
}

/* superbandwidth.c:121-152.  hops: `nhops` buffers of `gathered` complex
   samples each (modified in place, like the reference).  out must hold
   nhops * 2 * orc_fft_getrealsize(gathered) floats.  Returns the number of
   complex output samples; offsets[i] receives each hop's best offset. */
ORC_API uint32_t orc_superb_stitch(float **hops, int nhops, int gathered,
                                   int samples_in_frame, float *out, int *offsets)
{
    const uint32_t per = orc_fft_getrealsize((uint32_t)gathered);
    const uint32_t total = (uint32_t)nhops * per;
    const int nfl = (int)per * 2;
    float *tmp = (float *)malloc(sizeof(float) * (size_t)nfl);
    if (offsets) offsets[0] = 0;
    for (int i = 1; i < nhops; i++) {
        const int off = orc_superb_bestfit(hops[0], hops[i], nfl, samples_in_frame);
        if (offsets) offsets[i] = off;
        /* rotate left by `off` floats (superbandwidth.c:135-137) */
        memcpy(tmp, hops[i] + off, sizeof(float) * (size_t)(nfl - off));
        memcpy(hops[i] + (nfl - off), hops[i], sizeof(float) * (size_t)off);
        memcpy(hops[i], tmp, sizeof(float) * (size_t)(nfl - off));
        orc_fft_perform(hops[i], per, 0);
    }
    orc_fft_perform(hops[0], per, 0);
    for (int i = 0; i < nhops; i++)
        memcpy(out + (size_t)i * per * 2, hops[i], sizeof(float) * (size_t)per * 2);
    orc_fft_perform(out, total, 1);
    free(tmp);
    return total;
}

/* ------------------------------------------------------------------------ */
/* A.9  Dropped-sample bookkeeping — dsp.c:313-368                           */
/* ------------------------------------------------------------------------ */
static uint64_t orc_drop_comp(int block, int dropped) /* dsp.c:321-324 */
{
    const uint64_t frames = dropped / block;
    return ((frames + 1) * block - dropped) % block;
}

ORC_API int64_t orc_dropped_shift_with(int64_t difference, uint32_t block, int64_t syncoffset)
{
    /* dsp.c:354-368; `block` is uint32_t there, so `syncoffset % block`
       promotes per C rules: int64 % uint32 -> int64 */
    if (syncoffset >= 0)
        difference -= syncoffset % block;
    else
        difference -= block + syncoffset % block;
    if (difference < 0) difference = (int64_t)orc_drop_comp((int)block, (int)-difference);
    return difference;
}

/* dsp.c:326-346.  add_ok: whether the ring accepted the data.  Returns the
   new difference; *forward_from / *forward_len describe what was forwarded. */
ORC_API int64_t orc_dropped_add(int64_t difference, uint32_t size, uint32_t block, int add_ok,
                                uint32_t *forward_from, uint32_t *forward_len)
{
    *forward_from = 0;
    *forward_len = 0;
    if (size <= difference) return difference - size;
    if (add_ok) {
        *forward_from = (uint32_t)difference;
        *forward_len = size - (uint32_t)difference;
        return 0;
    }
    difference -= size % block;
    if (difference < 0) difference = (int64_t)orc_drop_comp((int)block, (int)-difference);
    return difference;
}

/* ------------------------------------------------------------------------ */
/* §8(f) rows — adjacent components                                          */
/* ------------------------------------------------------------------------ */

/* RawFile sample formats -> float, TSDRPlugin_RawFile/src/TSDRPlugin_RawFile.c:241-261.
   type: 0 float, 1 int8, 2 int16, 3 uint8, 4 uint16 (the plugin's TYPE_* ids, :29-33). */
ORC_API void orc_decode_samples(const void *raw, int type, float *out, int64_t n)
{
    switch (type) {
        case 0: memcpy(out, raw, sizeof(float) * (size_t)n); break;
        case 1: for (int64_t i = 0; i < n; i++) out[i] = ((const int8_t *)raw)[i] / 128.0; break;
        case 2: for (int64_t i = 0; i < n; i++) out[i] = ((const int16_t *)raw)[i] / 32767.0; break;
        case 3: for (int64_t i = 0; i < n; i++) out[i] = (((const uint8_t *)raw)[i] - 128) / 128.0; break;
        case 4: for (int64_t i = 0; i < n; i++) out[i] = (((const uint16_t *)raw)[i] - 32767) / 32767.0; break;
    }
}

/* frame -> packed 0x00RRGGBB, JavaGUI/jni/TSDRLibraryNDK.c:222-276.  Pixels equal
   to PIXEL_SPECIAL_VALUE_TRANSPARENT keep what `rgb` already holds. */
ORC_API void orc_frame_to_rgb(const float *frame, int32_t *rgb, int64_t n, int inverted)
{
    for (int64_t i = 0; i < n; i++) {
        const float val = frame[i];
        if (val > 0.0f && val <= 1.0f) {
            const int col = inverted ? (255 - (int)(val * 255.0f)) : ((int)(val * 255.0f));
            rgb[i] = col | (col << 8) | (col << 16);
        } else if (val <= 0.0f) {
            rgb[i] = inverted ? (255 | (255 << 8) | (255 << 16)) : 0;
        } else if (val == 256.0f) {
            rgb[i] = 255 << 16;
        } else if (val == 512.0f) {
            rgb[i] = 255 << 8;
        } else if (val == 1024.0f) {
            rgb[i] = 255;
        } else if (val == 2048.0f) {
            /* transparent: left as is */
        } else {
            rgb[i] = inverted ? 0 : (255 | (255 << 8) | (255 << 16));
        }
    }
}

/* ---------------------------------------------------------------------------
   Plot decimation for display, JavaGUI/src/martin/tempest/gui/PlotVisualizer.java:200-247
   (populateData) with the x mapping of gui/scale/ZoomableXScale.java:133-149:
       pixels_to_value_absolute(px) = px*one_px_in_values + offset_val + min_value
       value_to_pixel_absolute(v)   = (int)((v - min_value)*one_val_in_pixels) - offset_px
   visdata[] holds, per pixel column, the maximum of the lags that map to it (columns no lag maps
   to repeat the previous column's value), before the y scaling of PlotVisualizer.java:245-246.
   lowest/highest are what scale_y.setLowestHighestValue receives (:243); max_index is the
   argmax the GUI reads back through getMaxIndex() (:226-229).
   Pinned against the reference's compiled PlotVisualizer / ZoomableXScale classes executed by
   tests/golden/minijvm.py (tests/golden/java_fixtures_jvm.json, tests/test_extras_cpu.py).
   --------------------------------------------------------------------------- */
typedef struct {
    double one_val_in_pixels, one_px_in_values, offset_val, min_value;
    int offset_px;
} orc_plotscale_t;

/* ZoomableXScale after reset() + setMinMaxValue(0, size) + setMaxPixels(nwidth), zoom 1
   (PlotVisualizer.java:259-264,296; ZoomableXScale.java:177-188 with max_zoom_val = 10) */
ORC_API void orc_plotscale_default(int size, int nwidth, orc_plotscale_t *s)
{
    const double min_value = 0.0, max_value = (double)size, max_zoom_val = 10.0;
    double scale = 1.0;
    s->one_val_in_pixels = nwidth / ((max_value - min_value) * scale);
    s->one_px_in_values = ((max_value - min_value) * scale) / nwidth;
    const double values_in_screen = nwidth * s->one_px_in_values;
    if (values_in_screen < max_zoom_val) {
        scale = max_zoom_val / (max_value - min_value);
        s->one_val_in_pixels = nwidth / ((max_value - min_value) * scale);
        s->one_px_in_values = ((max_value - min_value) * scale) / nwidth;
    }
    s->offset_val = 0.0;
    s->min_value = min_value;
    s->offset_px = 0;
}

static double orc_px_to_val(const orc_plotscale_t *s, int px) { return px * s->one_px_in_values + s->offset_val + s->min_value; }
static int orc_val_to_px(const orc_plotscale_t *s, double v) { return (int)((v - s->min_value) * s->one_val_in_pixels) - s->offset_px; }

ORC_API void orc_plot_populate(const double *data, int size, int nwidth, const orc_plotscale_t *s, double *visdata,
                               double *lowest, double *highest, int *max_index)
{
    double highest_val = data[0];
    double lowest_val = highest_val;
    int maxi = 0;
    double max_val = highest_val;
    int prev_px = 0;
    double t = orc_px_to_val(s, 0);
    t = t > 0 ? t : 0;
    const int first_id = (int)(t < size ? t : size);
    t = orc_px_to_val(s, nwidth) + 1;
    t = t > 0 ? t : 0;
    const int last_id = (int)(t < size ? t : size);
    /* Java would throw on data[size]; a scale that starts at or beyond the end shows nothing */
    double localmax = first_id < size ? data[first_id] : data[size - 1];
    for (int id = first_id; id < last_id; id++) {
        const double val = data[id];
        const int px = orc_val_to_px(s, id);
        if (px >= 0 && px < nwidth) {
            if (prev_px != px) {
                if (localmax > highest_val) highest_val = localmax;
                else if (localmax < lowest_val) lowest_val = localmax;
                for (int i = prev_px; i < px; i++) visdata[i] = localmax;
                localmax = val;
                prev_px = px;
            } else if (val > localmax)
                localmax = val;
        }
        if (val > max_val) {
            max_val = val;
            maxi = id;
        }
    }
    for (int i = prev_px; i < nwidth; i++) visdata[i] = localmax;
    *lowest = lowest_val;
    *highest = highest_val;
    *max_index = maxi;
}
