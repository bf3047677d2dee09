/*
 * ref_shim.c — flat accessors around the REAL reference library.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/README.md).  This file is compiled
 * together with the reference's own sources, taken where they lie under
 * /root/reference/TempestSDR/src (never copied into this repository), into
 * oracle/_ref/libtsdr_ref.so by oracle/Makefile.  It exists only because the
 * reference's stage functions take library structs (tsdr_lib_t, extbuffer_t,
 * superbandwidth_t) whose layout ctypes should not have to know: the shim
 * includes the reference headers at build time and exposes plain-pointer
 * entry points for tests/ and tests/golden/make_golden.py.
 */
#include "internaldefinitions.h" /* reference header, -I$(REF)/TempestSDR/src */
#include "fft.h"
#include <string.h>
#include <stdio.h>

#define SHIM_API __attribute__((visibility("default")))

static void shim_value_cb(int id, double a0, double a1, void *ctx)
{
    /* record the last PLL / autogain announcements so tests can see them */
    double *slots = (double *)ctx;
    if (!slots) return;
    if (id >= 0 && id < 6) {
        slots[3 * id] += 1.0;
        slots[3 * id + 1] = a0;
        slots[3 * id + 2] = a1;
    }
}

/* A tsdr_lib_t set up the way SURVEY.md §8(c) prescribes for the
   single-threaded deterministic driver. */
SHIM_API tsdr_lib_t *ref_new(int height, double refreshrate, uint32_t samplerate, float motionblur,
                             double *value_slots /* 18 doubles or NULL */)
{
    tsdr_lib_t *t = NULL;
    tsdr_init(&t, shim_value_cb, NULL, value_slots);
    t->errormsg = NULL; /* tsdr_init leaves it uninitialised, tsdr_free frees it */
    t->height = height;
    t->refreshrate = refreshrate;
    t->samplerate_real = samplerate;
    t->gain = 0.5f;
    t->motionblur = motionblur;
    set_internal_samplerate(t, samplerate);
    return t;
}

SHIM_API void ref_free(tsdr_lib_t *t) { tsdr_free(&t); }
SHIM_API int ref_width(tsdr_lib_t *t) { return t->width; }
SHIM_API int ref_height(tsdr_lib_t *t) { return t->height; }
SHIM_API double ref_refreshrate(tsdr_lib_t *t) { return t->refreshrate; }
SHIM_API double ref_pixelrate(tsdr_lib_t *t) { return t->pixelrate; }
SHIM_API double ref_pixeltimeoversampletime(tsdr_lib_t *t) { return t->pixeltimeoversampletime; }
SHIM_API void ref_setparam(tsdr_lib_t *t, int id, uint32_t v) { t->params_int[id] = v; }

SHIM_API float *ref_post_process(tsdr_lib_t *t, float *frame, float motionblur, float lowpasscoeff,
                                 int lowpass_before_sync, int autogain_after_proc)
{
    return dsp_post_process(t, &t->dsp_postprocess, frame, t->width, t->height, motionblur,
                            lowpasscoeff, lowpass_before_sync, autogain_after_proc);
}

/* same slot layout as orc_postprocess_state (slot 7/8 are not observable here) */
SHIM_API void ref_postprocess_state(tsdr_lib_t *t, int32_t out_i[10], double out_d[4])
{
    dsp_postprocess_t *pp = &t->dsp_postprocess;
    out_i[0] = pp->sync.db_x.dx; out_i[1] = pp->sync.db_x.vx;
    out_i[2] = pp->sync.db_x.curr_stripsize;
    out_i[3] = pp->sync.db_y.dx; out_i[4] = pp->sync.db_y.vx;
    out_i[5] = pp->sync.db_y.curr_stripsize;
    out_i[6] = pp->sync.state; out_i[7] = -1; out_i[8] = -1; out_i[9] = pp->runs;
    out_d[0] = pp->dsp_autogain.lastmin; out_d[1] = pp->dsp_autogain.lastmax;
    out_d[2] = pp->dsp_autogain.snr; out_d[3] = pp->sync.avg_speed;
}
SHIM_API const float *ref_colsum(tsdr_lib_t *t) { return t->dsp_postprocess.widthcollapsebuffer; }
SHIM_API const float *ref_rowsum(tsdr_lib_t *t) { return t->dsp_postprocess.heightcollapsebuffer; }

/* ---- resampler: wrap the extbuffer plumbing ---- */
typedef struct {
    dsp_resample_t st;
    extbuffer_t in, out;
} ref_resampler_t;

SHIM_API ref_resampler_t *ref_resampler_new(void)
{
    ref_resampler_t *r = (ref_resampler_t *)calloc(1, sizeof(*r));
    dsp_resample_init(&r->st);
    extbuffer_init(&r->in);
    extbuffer_init(&r->out);
    return r;
}
SHIM_API void ref_resampler_free(ref_resampler_t *r)
{
    extbuffer_free(&r->in);
    extbuffer_free(&r->out);
    free(r);
}
SHIM_API void ref_resampler_state(ref_resampler_t *r, double st[2])
{
    st[0] = r->st.contrib;
    st[1] = r->st.offset;
}
SHIM_API void ref_resampler_setstate(ref_resampler_t *r, double contrib, double offset)
{
    r->st.contrib = contrib;
    r->st.offset = offset;
}
/* returns the announced output count; copies that many floats to `out` */
SHIM_API uint32_t ref_resampler_process(ref_resampler_t *r, const float *in, uint32_t size, float *out,
                                        double up, double down, int nearest)
{
    extbuffer_preparetohandle(&r->in, size);
    memcpy(r->in.buffer, in, sizeof(float) * size);
    dsp_resample_process(&r->st, &r->in, &r->out, up, down, nearest);
    memcpy(out, r->out.buffer, sizeof(float) * r->out.size_valid_elements);
    return r->out.size_valid_elements;
}

/* ---- accummulate: build the two extbuffers around caller memory ---- */
SHIM_API void ref_accumulate(double *acc, float *corr, int start, int length, uint64_t calls)
{
    extbuffer_t in, out;
    extbuffer_init(&in);
    extbuffer_init_double(&out);
    in.buffer = corr;
    in.valid = 1;
    in.calls = calls;
    in.cleartozero = 0;
    out.dbuffer = acc;
    out.buffer_max_size = (uint32_t)length; /* no realloc in preparetohandle */
    out.valid = 1;
    out.cleartozero = 0;
    accummulate(&out, &in, start, length);
}

/* ---- PARAM_AUTOCORR_DUMP: the reference's own autocorrelate() + dump_autocorrect() (frameratedetector.c:26-32,64-85)
 * on one capture window; writes autocorr.csv into the current directory exactly like the library's detector thread ---- */
void autocorrelate(extbuffer_t *buff, float *data, int size);
void dump_autocorrect(extbuffer_t *rawiq, double samplerate);
SHIM_API int ref_dump_autocorr(float *window, int size, double samplerate)
{
    extbuffer_t buf;
    extbuffer_init(&buf);
    autocorrelate(&buf, window, size);
    if (!buf.valid) return -1;
    dump_autocorrect(&buf, samplerate);
    extbuffer_free(&buf);
    return 0;
}

/* ---- super-bandwidth stitch on caller-provided hop buffers ---- */
SHIM_API uint32_t ref_superb_stitch(tsdr_lib_t *t, float **hops, int nhops, int gathered,
                                    int samples_in_frame, uint32_t samplerate, float *out)
{
    superbandwidth_t bw;
    superb_init(&bw);
    bw.alive = 1;
    bw.tsdr = t;
    bw.samplerate = samplerate;
    bw.samples_in_frame = samples_in_frame;
    bw.buffscount = nhops;
    bw.buffsbuffcount = gathered;
    bw.buffs = hops;
    float *res = NULL;
    int ressize = 0;
    superb_ondataready(&bw, &res, &ressize, t);
    if (res) memcpy(out, res, sizeof(float) * 2 * (size_t)ressize);
    bw.buffs = NULL; /* caller owns the hop buffers */
    bw.buffscount = 0;
    extbuffer_free(&bw.extb);
    extbuffer_free(&bw.extb_out);
    extbuffer_free(&bw.extb_temp);
    return (uint32_t)ressize;
}
