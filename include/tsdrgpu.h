/*
 * tsdrgpu.h — C ABI of the MI355X (gfx950) implementation of TempestSDR's DSP
 * hot path.  Plain pointers and sizes only; every pointer named d_* is a HIP
 * device pointer, h_* a host pointer.  All calls are asynchronous on the
 * context's stream unless stated; tsdrgpu_sync() waits for them.
 *
 * Each entry point names the reference code (file:line under
 * TempestSDR/src/ of martinmarinov/TempestSDR) it replaces.  The reference has
 * no FFI of its own for these stages (they are internal C functions behind the
 * tsdr_* API, include/TSDRLibrary.h); this header is what a maintainer would
 * bind instead of those internals, and what
 * libTSDRLibrary.so in this repository — the drop-in for the tsdr_* API — is
 * built on.
 *
 * Threading: a context and the objects created from it belong to one host thread at a time (the
 * library takes no locks); use one context per thread / per GPU.  Limits: width*height <= 4000*4000 pixels per frame, any
 * width and height >= 1 within that (the reference's own bound, TSDRLibrary.c:31,489); <= 65535 resampler chunks or
 * autocorrelation windows per call.
 *
 * Return value: 0 (TSDRGPU_OK) or a negative TSDRGPU_E* code;
 * tsdrgpu_last_error() gives the text of the CALLING THREAD's last failing call.  There is no CPU fallback: without a
 * HIP device tsdrgpu_create() fails.
 */
#ifndef TSDRGPU_H_
#define TSDRGPU_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TSDRGPU_OK 0
#define TSDRGPU_EHIP (-1)     /* a HIP runtime call failed */
#define TSDRGPU_EINVAL (-2)   /* bad argument */
#define TSDRGPU_ENOMEM (-3)   /* device/host allocation failed */
#define TSDRGPU_ESTATE (-4)   /* object not configured for this call */

typedef struct tsdrgpu tsdrgpu_t;                     /* device context: device, stream, scratch */
typedef struct tsdrgpu_resampler tsdrgpu_resampler_t; /* dsp_resample_t on the device */
typedef struct tsdrgpu_postproc tsdrgpu_postproc_t;   /* dsp_postprocess_t on the device */
typedef struct tsdrgpu_autocorr tsdrgpu_autocorr_t;   /* frameratedetector numerics on the device */

/* ---- context and memory ------------------------------------------------- */
int tsdrgpu_create(tsdrgpu_t **out, int device);
void tsdrgpu_destroy(tsdrgpu_t *g);
const char *tsdrgpu_last_error(tsdrgpu_t *g);
int tsdrgpu_sync(tsdrgpu_t *g);
void *tsdrgpu_stream(tsdrgpu_t *g); /* the hipStream_t all work is queued on */
int tsdrgpu_device_name(tsdrgpu_t *g, char *buf, size_t buflen);

int tsdrgpu_alloc(tsdrgpu_t *g, void **d_ptr, size_t bytes);
int tsdrgpu_free(tsdrgpu_t *g, void *d_ptr);
int tsdrgpu_alloc_host(tsdrgpu_t *g, void **h_ptr, size_t bytes); /* pinned */
int tsdrgpu_free_host(tsdrgpu_t *g, void *h_ptr);
int tsdrgpu_upload(tsdrgpu_t *g, void *d_dst, const void *h_src, size_t bytes);
int tsdrgpu_download(tsdrgpu_t *g, void *h_dst, const void *d_src, size_t bytes);
int tsdrgpu_copy(tsdrgpu_t *g, void *d_dst, const void *d_src, size_t bytes);
int tsdrgpu_copy2(tsdrgpu_t *g, void *d_dst1, void *d_dst2, const void *d_src, size_t bytes); /* one launch, two destinations */
/* n (<= 32) device blocks appended back to back to d_dst1 and, if not NULL, d_dst2, in one launch on the COMPUTE lane */
int tsdrgpu_gather2(tsdrgpu_t *g, void *d_dst1, void *d_dst2, const void *const *d_srcs, const size_t *bytes, int n);
int tsdrgpu_zero(tsdrgpu_t *g, void *d_ptr, size_t bytes);

/* ---- lanes and events: what a streaming host needs to overlap PCIe copies with compute ------------
 * A context owns five in-order queues ("lanes"): COMPUTE (the stream every kernel entry point above and below
 * uses), SIDE (high priority: the sync detector's short chain kernels), BACKGROUND (low priority: an asynchronous
 * autocorrelation), UPLOAD and DOWNLOAD (copy engines).  Nothing orders two lanes except events.  All of this is thread safe as long as each
 * lane is fed by one thread at a time (the engine: the plugin's thread feeds UPLOAD, the device thread the rest). */
#define TSDRGPU_LANE_COMPUTE 0
#define TSDRGPU_LANE_SIDE 1
#define TSDRGPU_LANE_UPLOAD 2
#define TSDRGPU_LANE_DOWNLOAD 3
#define TSDRGPU_LANE_BACKGROUND 4 /* lowest priority: where tsdrgpu_autocorr_set_async puts the detector, to fill the gaps of the frame path */
typedef struct tsdrgpu_event tsdrgpu_event_t;
int tsdrgpu_event_create(tsdrgpu_t *g, tsdrgpu_event_t **out);
void tsdrgpu_event_destroy(tsdrgpu_t *g, tsdrgpu_event_t *ev);
int tsdrgpu_event_record(tsdrgpu_t *g, tsdrgpu_event_t *ev, int lane); /* completes when everything queued on `lane` so far has */
int tsdrgpu_lane_wait(tsdrgpu_t *g, int lane, tsdrgpu_event_t *ev);    /* work queued on `lane` from now on waits for ev */
int tsdrgpu_event_sync(tsdrgpu_t *g, tsdrgpu_event_t *ev);            /* the calling host thread waits */
int tsdrgpu_event_done(tsdrgpu_t *g, tsdrgpu_event_t *ev);            /* 1 done, 0 not yet, < 0 error */
int tsdrgpu_lane_sync(tsdrgpu_t *g, int lane);                        /* the calling host thread waits for the lane */
int tsdrgpu_upload_lane(tsdrgpu_t *g, void *d_dst, const void *h_src, size_t bytes);   /* asynchronous, UPLOAD lane */
int tsdrgpu_download_lane(tsdrgpu_t *g, void *h_dst, const void *d_src, size_t bytes); /* asynchronous, DOWNLOAD lane */
/* Page-locks memory the caller does not own the allocation of (e.g. a source plugin's sample buffer), so that
 * tsdrgpu_upload_lane can DMA straight out of it instead of through a pinned bounce buffer. */
int tsdrgpu_host_register(tsdrgpu_t *g, void *h_ptr, size_t bytes);
int tsdrgpu_host_unregister(tsdrgpu_t *g, void *h_ptr);
/* Makes the context's device the calling thread's current HIP device; call once from every thread that uses
 * the context (allocation entry points do it themselves). */
int tsdrgpu_bind_thread(tsdrgpu_t *g);

/* event timing on the context's stream (used by bench.py for the roofline leg) */
int tsdrgpu_timer_start(tsdrgpu_t *g);
int tsdrgpu_timer_stop_ms(tsdrgpu_t *g, float *ms); /* synchronises */

/* optional per-stage event profiler: while enabled every kernel launch group is
 * bracketed by a HIP event pair on the stream.  tsdrgpu_profile_end()
 * synchronises and reports, per stage, the summed duration and launch count.
 * Stage names are the kernel names (k_fft_pass, k_frame_pass, ...). */
typedef struct tsdrgpu_profile_entry {
    char name[32];
    double total_ms;
    int launches;
} tsdrgpu_profile_entry_t;
int tsdrgpu_profile_begin(tsdrgpu_t *g);
int tsdrgpu_profile_end(tsdrgpu_t *g, tsdrgpu_profile_entry_t *h_entries, int max_entries, int *h_count);

/* ---- a1: AM demodulation -------------------------------------------------- */
/* am_demod, TSDRLibrary.c:244-262 (== complex_to_real, fft.c:24-32):
 * d_out[i] = sqrtf(I*I + Q*Q), float32, bit-exact.  d_out must not overlap d_iq
 * (the reference's in-place form is inherently sequential). */
int tsdrgpu_am_demod(tsdrgpu_t *g, const float *d_iq, float *d_out, int64_t nsamples);

/* ---- a2: fractional area resampler ----------------------------------------- */
/* dsp_resample_t / dsp_resample_process, dsp.c:250-307.  One call replays
 * `nchunks` consecutive dsp_resample_process calls of `chunk` input samples each
 * (the library polls 0.1 frame at a time, TSDRLibrary.c:335-340), carrying
 * `offset` (host, double) and `contrib` (device) exactly like the reference.
 *   d_in        magnitude samples (in_is_iq = 0) or interleaved IQ
 *               (in_is_iq = 1: am_demod is fused, a1+a2)
 *   d_out       pixel stream; out_capacity floats
 *   h_n_out     total pixels written (known on the host immediately: the
 *               counts depend on sizes and phases, not on data)
 */
int tsdrgpu_resampler_create(tsdrgpu_t *g, tsdrgpu_resampler_t **out);
void tsdrgpu_resampler_destroy(tsdrgpu_resampler_t *rs);
int tsdrgpu_resampler_reset(tsdrgpu_resampler_t *rs); /* dsp_resample_init, dsp.c:250-254 */
int tsdrgpu_resampler_setstate(tsdrgpu_resampler_t *rs, double contrib, double offset);
int tsdrgpu_resampler_getstate(tsdrgpu_resampler_t *rs, double *contrib, double *offset); /* syncs */
int tsdrgpu_resample(tsdrgpu_resampler_t *rs, const float *d_in, int in_is_iq, uint32_t chunk,
                     int nchunks, double upsample_by, double downsample_by, int nearest,
                     float *d_out, int64_t out_capacity, int64_t *h_n_out);
/* Frame tracking (optional, area mode).  When on, tsdrgpu_resample also reduces, for every video frame of
 * `frame_pixels` consecutive output pixels, the minimum and maximum over the non-sentinel pixels
 * (|v| <= 250) it emits — pass 1 of dsp_autogain_run (dsp.c:41-66) without reading the frame again.
 * `phase` = pixels of the current, still incomplete frame that earlier calls emitted (0 on a frame
 * boundary); from then on the frame boundaries follow the pixel count from call to call, so call this
 * again after samples were dropped or the resolution changed.  frame_pixels = 0 switches tracking off;
 * otherwise frame_pixels >= 4096.
 * tsdrgpu_resampler_frame_minmax: after a tracked tsdrgpu_resample, device arrays (valid until the next
 * call, written on the context's stream) with the min / max of the frames that call COMPLETED, in order
 * (the first one includes the pixels earlier calls contributed), and their number. */
int tsdrgpu_resampler_track_frames(tsdrgpu_resampler_t *rs, int64_t frame_pixels, int64_t phase);
int tsdrgpu_resampler_frame_minmax(tsdrgpu_resampler_t *rs, const float **d_min, const float **d_max, int *nframes);
/* Row-band form (the frame path sharded over GPUs by rows).  The same `nchunks` calls of
 * dsp_resample_process with the same carried state — every rank passes the same samples and ends with the same
 * `offset` / `contrib` — but only the pixels of rows [y0, y0 + rows) of each width x height frame are computed and
 * stored: frame j, counted from the frame the call's first pixel lies in, goes to d_band + j * rows * width.
 * `phase` = pixels of that first frame which earlier calls produced (0 when the call starts on a frame boundary); the
 * caller carries an incomplete last frame (slot *h_frames_touched - 1 when (phase + *h_n_out) % (width*height) != 0) into
 * slot 0 of its next call.  *h_n_out = pixels the full call would have produced.  Area mode; values bit-identical to
 * the corresponding pixels of tsdrgpu_resample's output.  Ratios 1 <= r <= 8 with frames of >= 4096 pixels (the reference's
 * geometry) run the sample-parallel kernel in its band form — a band is a contiguous pixel range of every frame, workgroups
 * outside it return before loading —, anything else the pixel-group kernel over a table of band entries (dsp.c:256-307
 * entered at each pixel group through the closed form of resample_math.h).  A chunk may be longer than a frame (its output
 * then touches several frames); at most 65535 band entries per call in the pixel-group form.
 * With frame tracking on (tsdrgpu_resampler_track_frames(rs, width*height, phase): same frame size, and `phase` must be the
 * tracker's) the call also leaves the min / max of every completed frame over THIS BAND's pixels (sentinels skipped,
 * dsp.c:57) in tsdrgpu_resampler_frame_minmax's arrays — +inf / -inf for a frame the band holds no ordinary pixel of —, which
 * is what tsdrgpu_postproc_band_begin_minmax exchanges; sample-parallel kernel only. */
int tsdrgpu_resample_band(tsdrgpu_resampler_t *rs, const float *d_in, int in_is_iq, uint32_t chunk, int nchunks,
                          double upsample_by, double downsample_by, int width, int height, int y0, int rows, int64_t phase,
                          float *d_band, int64_t band_capacity_frames, int64_t *h_n_out, int *h_frames_touched);
/* Pixel count the next `nchunks` calls would produce, without running them. */
int64_t tsdrgpu_resample_count(tsdrgpu_resampler_t *rs, uint32_t chunk, int nchunks,
                               double upsample_by, double downsample_by);

/* ---- a3..a8: frame post-processing ------------------------------------------ */
/* dsp_post_process, dsp.c:134-239, with dsp_autogain_run (dsp.c:41-94),
 * dsp_average_v_h (dsp.c:96-110), syncdetector_run incl. gaussianblur,
 * findthesweetspot, frameratepll (syncdetector.c:26-225, gaussian.c:18-79) and
 * dsp_timelowpass_run (dsp.c:22-33), all on the device. */
typedef struct tsdrgpu_pp_params {
    int lowpass_before_sync;  /* PARAM_LOW_PASS_BEFORE_SYNC */
    int autogain_after_proc;  /* PARAM_AUTOGAIN_AFTER_PROCESSING */
    int autoshift;            /* PARAM_INT_AUTOSHIFT */
    int pll;                  /* PARAM_INT_FRAMERATE_PLL */
    int superresolution;      /* PARAM_AUTOCORR_SUPERRESOLUTION (suppresses green lines) */
    float motionblur;         /* tsdr_motionblur coefficient */
    float lowpasscoeff;       /* NORMALISATION_LOWPASS_COEFF = 0.1f, TSDRLibrary.c:37 */
} tsdrgpu_pp_params_t;

/* What the reference leaves in dsp_postprocess_t / announces after a frame. */
typedef struct tsdrgpu_pp_frameinfo {
    float lastmin, lastmax;   /* dsp_autogain_t after this frame */
    int dx, vx, stripx;       /* sync.db_x: dx, vx, curr_stripsize */
    int dy, vy, stripy;       /* sync.db_y */
    int locked;               /* sync.state */
    int pll_fired;            /* frameratepll nudged the refresh rate */
    double avg_speed;         /* sync.avg_speed */
    double frameratediff;     /* amount subtracted from refreshrate (0 when !pll_fired) */
} tsdrgpu_pp_frameinfo_t;

int tsdrgpu_postproc_create(tsdrgpu_t *g, tsdrgpu_postproc_t **out);
void tsdrgpu_postproc_destroy(tsdrgpu_postproc_t *pp);
int tsdrgpu_postproc_reset(tsdrgpu_postproc_t *pp); /* dsp_post_process_init, dsp.c:112-132 */
/* Exact ties.  The sync detector's strips (column / row sums) are formed from f64 tile sums here and by f32 additions in
 * raster order in the reference (dsp.c:96-110); they agree to ~1e-6.  Frames whose strips hold exact ties (blank,
 * plateaus, periodic patterns, nearly flat) are always re-collapsed in the reference's order, so those decisions are
 * identical.  What remains is the rare decision whose margin over the runner-up is smaller than the reference's own
 * rounding (measured: ~3e-5 of the decisions on smooth repeated frames): with `on`, such toss-ups are detected (margin
 * test against the rounding of the reference's sums), the frames concerned get exact strips and the chain of the batch
 * is run again — the sync state is then identical there too.  Costs ~0.1 ms per batch plus ~0.3 ms for each batch that
 * holds a toss-up (about every batch on noisy rasters), on the chain's stream. */
int tsdrgpu_postproc_set_exact_ties(tsdrgpu_postproc_t *pp, int on);
/* Diagnostics of the last run (synchronises): how many of its 2*nframes sync decisions the first chain run marked as
 * toss-ups, how many strips were re-collapsed in the reference's order because of that, and how many had been flagged
 * for the literal collapse up front (ties / sentinels / flat strips). */
int tsdrgpu_postproc_redo_stats(tsdrgpu_postproc_t *pp, int *h_tossups, int *h_recollapsed, int *h_flagged_upfront);
int tsdrgpu_postproc_redo_raw(tsdrgpu_postproc_t *pp, int *h, int cap_ints, int *h_frames); /* the three flag arrays, 2*F ints each */
/* Runs `nframes` consecutive frames (d_frames: nframes*width*height floats,
 * raster order) through dsp_post_process in order, as one batch of launches.
 * d_out receives every frame's result (what the reference hands to the video
 * thread).  h_info (nframes entries, may be NULL): when given, the call
 * synchronises and fills it.  The refresh-rate PLL's effect on width/refresh
 * is the caller's job (apply h_info[i].frameratediff, TSDRLibrary.c:540-550),
 * so with params->pll the caller should pass one frame at a time.
 * Any width >= 1 and height >= 1 with width*height <= TSDRGPU_MAX_FRAME_PIXELS, one-row and one-column frames included (the
 * reference shows a line of pixels for those; so does this).  A strip — a frame's width or height — of up to TSDRGPU_MAX_STRIP
 * entries is blurred and scanned in LDS, a longer one in HBM (same results, slower: rasters of a few lines at a high rate). */
#define TSDRGPU_MAX_FRAME_PIXELS (4000 * 4000) /* MAX_ARR_SIZE, TSDRLibrary.c:31 */
#define TSDRGPU_MAX_STRIP 16384                /* longest strip the sync detector keeps in LDS; NOT a limit of the interface */
int tsdrgpu_postproc_run(tsdrgpu_postproc_t *pp, const float *d_frames, int nframes, int width,
                         int height, const tsdrgpu_pp_params_t *params, float *d_out,
                         tsdrgpu_pp_frameinfo_t *h_info);
/* The same run in two halves, for the default stage order (autogain, sync, low-pass): _begin queues the
 * frame statistics on the context's stream and the short frame-to-frame chain (autogain IIR, strip blur,
 * sync search, PLL: latency-bound, ~0.1 ms per batch during which the GPU is nearly idle) on its side
 * stream; _finish makes the main stream wait for the chain and queues the normalise / low-pass pass.
 * Work the caller queues on the context's stream in between (typically tsdrgpu_autocorr_run on the same
 * samples) overlaps the chain.  d_frames must stay untouched until _finish; results are bit-identical to
 * tsdrgpu_postproc_run.  With the other stage orders _begin only records its arguments and _finish runs
 * everything.  tsdrgpu_postproc_run / _begin return TSDRGPU_ESTATE while a split run is open. */
int tsdrgpu_postproc_begin(tsdrgpu_postproc_t *pp, const float *d_frames, int nframes, int width, int height,
                           const tsdrgpu_pp_params_t *params);
int tsdrgpu_postproc_finish(tsdrgpu_postproc_t *pp, float *d_out, tsdrgpu_pp_frameinfo_t *h_info);
/* Fused form of _begin for a caller that already holds each frame's min / max over its non-sentinel
 * pixels (tsdrgpu_resampler_frame_minmax; device arrays of nframes floats): with the default stage order
 * and autoshift off, the autogain IIR runs from those values and ONE trip over the raw frames normalises,
 * low-passes, writes d_out and gathers the row/column sums of the sync detector, which then runs (with the
 * green lines of syncdetector.c:209-223 patched in afterwards) on the side stream — the separate
 * statistics read of every frame (4 bytes/pixel of 16) disappears.  Results are bit-identical to
 * tsdrgpu_postproc_run.  Everything is queued here, so d_out is given now; _finish(pp, d_out, h_info) joins
 * the streams (the same d_out; the raw frames must stay untouched until then: _finish may redo the batch from
 * them).  Any other parameter combination silently takes the _begin path (min/max ignored).
 * With motion blur 0 and nframes >= 8 the trip is one FLAT kernel over (tile, frame) — the statistics kernel that
 * also stores the normalised pixels; a -0.0 or non-finite pixel or state raises a device flag and the literal pass,
 * queued by _finish and gated on that flag, redoes the batch.  On MI355X (1080p, 60-frame batches) that is the fast
 * form: 12P instead of 16P bytes per frame, +5 % on the whole pass when the caller lets a batch's sync detector run
 * beside the next batch's resampler (call _finish for batch k behind tsdrgpu_resample of batch k+1, with two pixel
 * buffers; bench.py does).  With motion blur > 0 the trip has to walk the frames tile by tile with the IIR state in
 * registers: a slower kernel, but in the same call order still 1-4 % ahead of _begin/_finish. */
int tsdrgpu_postproc_begin_minmax(tsdrgpu_postproc_t *pp, const float *d_frames, int nframes, int width, int height,
                                  const tsdrgpu_pp_params_t *params, const float *d_fmin, const float *d_fmax,
                                  float *d_out);
/* Row-band sharding of the frame path across GPUs: this rank holds rows [y0, y0+rows) of every
 * frame (d_band: nframes * rows * width floats), y0 a multiple of 32.  _band_begin computes the band's statistics and
 * exposes two small device buffers; the caller all-reduces them IN PLACE across the ranks — *d_xsum (n_xsum doubles)
 * with ncclSum, *d_xmax (n_xmax floats) with ncclMax: tsdrgpu_comm_allreduce_f64 / tsdrgpu_comm_allreduce_f32max —
 * and _band_finish runs the (replicated) autogain / sync-detector chain and the normalise / green-lines / IIR pass on
 * the band's rows into d_out_band.  Library-default stage order, no autoshift, no PLL.  With _band_finish frames are
 * bit-identical to tsdrgpu_postproc_run's in its fast mode (tsdrgpu_postproc_set_exact_ties(pp, 0)); with
 * _band_advance (below) to its default, contract-exact mode; a one-band "sharding" needs no exchange at all.
 * Reference: dsp.c:41-110, syncdetector.c:171-225. */
int tsdrgpu_postproc_band_begin(tsdrgpu_postproc_t *pp, const float *d_band, int nframes, int width, int height, int y0, int rows,
                                const tsdrgpu_pp_params_t *params, double **d_xsum, int64_t *n_xsum, float **d_xmax, int64_t *n_xmax);
int tsdrgpu_postproc_band_finish(tsdrgpu_postproc_t *pp, float *d_out_band, tsdrgpu_pp_frameinfo_t *h_info);
/* The contract-exact form of _band_finish (tsdrgpu_postproc_set_exact_ties, on by default): frames and per-frame
 * records bit-identical to tsdrgpu_postproc_run in ITS default mode, i.e. to the reference.  The literal collapse of
 * a strip (f32 additions in raster order, dsp.c:96-110) — needed where a strip holds exact ties and where a sync
 * decision is a toss-up — walks every column through all bands from the top, so the bands take turns: call
 *     do { tsdrgpu_postproc_band_advance(pp, d_out_band, band_index, nbands, &d_buf, &n, &more, h_info);
 *          if (more) <sum all-reduce of d_buf[0..n) over the ranks, in place: tsdrgpu_comm_allreduce_f64>; } while (more);
 * after the two all-reduces of _band_begin.  band_index = this rank's position in the order of the bands from the top
 * row down (0 .. nbands-1, one band per rank).  Whether and how often `more` is raised follows from the exchanged
 * strips alone, identically on every rank: not at all for an ordinary batch, nbands times for a batch with flagged
 * strips, again nbands times if a decision was a toss-up.  The two questions behind that ("does a strip hold ties?",
 * "was a decision a toss-up?") are SPECULATED: both halves of the chain are queued as if no strip held ties, both flag
 * arrays are copied out behind them and the host waits once — an ordinary batch then goes straight to the pass, a batch
 * with toss-ups only goes on with the second relay (what was queued is the literal run), and only a batch whose strips
 * hold ties is put back to the autogain / sync state it started from (saved on the device) and taken literally from the
 * start, so results never depend on the speculation (TSDRGPU_BAND_SPECULATE=0 in the environment: every batch literally,
 * two host round trips each).  Synchronises.  With exact ties off it is _band_finish.
 * Reference: syncdetector.c:26-153, dsp.c:96-110. */
int tsdrgpu_postproc_band_advance(tsdrgpu_postproc_t *pp, float *d_out_band, int band_index, int nbands, double **d_buf,
                                  int64_t *n_buf, int *h_more, tsdrgpu_pp_frameinfo_t *h_info);
/* The FUSED band run — the band form of tsdrgpu_postproc_begin_minmax: with frame tracking on (tsdrgpu_resampler_track_frames)
 * tsdrgpu_resample_band leaves every frame's min/max over THIS band's pixels (tsdrgpu_resampler_frame_minmax), the ranks
 * exchange that range BEFORE the band is read, and one trip over the raw band writes the normalised / IIR'd rows and gathers
 * the strip partials: 12 bytes per band pixel instead of the 16 of _band_begin + _band_advance.  Every rank alike:
 *     tsdrgpu_postproc_band_begin_minmax(pp, d_band, F, W, H, y0, rows, &prm, d_fmin, d_fmax, &d_xmax, &n);
 *     <max all-reduce of d_xmax[0..n) over the ranks, in place: tsdrgpu_comm_allreduce_f32max>
 *     tsdrgpu_postproc_band_fused(pp, d_out_band, &d_xsum, &m);          // the trip
 *     <sum all-reduce of d_xsum[0..m): tsdrgpu_comm_allreduce_f64>
 *     do { tsdrgpu_postproc_band_advance(pp, d_out_band, ...); if (more) <sum all-reduce>; } while (more);   // chain, relays, lines
 * Same restrictions as _band_begin (library-default order, no autoshift, no PLL, y0 a multiple of 32); d_out_band must not
 * overlap d_band (the relays and the painted lines read the raw band after the trip).  Frames and per-frame records are
 * bit-identical to _band_begin + _band_advance, i.e. to tsdrgpu_postproc_run in its default mode and to the reference.
 * Reference: dsp.c:41-110,134-239, syncdetector.c:171-225. */
int tsdrgpu_postproc_band_begin_minmax(tsdrgpu_postproc_t *pp, const float *d_band, int nframes, int width, int height, int y0, int rows,
                                       const tsdrgpu_pp_params_t *params, const float *d_fmin, const float *d_fmax, float **d_xmax,
                                       int64_t *n_xmax);
int tsdrgpu_postproc_band_fused(tsdrgpu_postproc_t *pp, float *d_out_band, double **d_xsum, int64_t *n_xsum);
/* how many band runs were speculated since the object was created, and how many of those had to be put back and taken literally */
int tsdrgpu_postproc_band_spec_stats(tsdrgpu_postproc_t *pp, uint64_t *runs, uint64_t *replays);
/* The band run in its GENERAL form: every stage order of dsp_post_process (dsp.c:134-239: PARAM_LOW_PASS_BEFORE_SYNC,
 * PARAM_AUTOGAIN_AFTER_PROCESSING — the GUI's default order among them), PARAM_INT_AUTOSHIFT (the 2-D roll,
 * syncdetector.c:187-207) and PARAM_INT_FRAMERATE_PLL (syncdetector.c:133-153; replicated: every rank computes the same
 * nudge, reported in h_info — pass one frame per run then, as with tsdrgpu_postproc_run).  `edges` = the row edges of all
 * nbands bands (nbands + 1 values from 0 to height, each band starting on a multiple of 32 rows), band_index = this rank's.
 *     tsdrgpu_postproc_band_open(pp, d_band, F, W, H, edges, nbands, my_band, &prm);
 *     do { tsdrgpu_postproc_band_step(pp, d_out_band, &x, h_info);
 *          switch (x.kind) {                       // every rank is asked for the same collective at the same step
 *          case TSDRGPU_BAND_SUM_F64:       tsdrgpu_comm_allreduce_f64(comm, x.d_buf, x.count, lane);            break;
 *          case TSDRGPU_BAND_MAX_F32:       tsdrgpu_comm_allreduce_f32max(comm, x.d_buf, x.count, lane);         break;
 *          case TSDRGPU_BAND_ALLGATHER_F32: tsdrgpu_comm_allgather_f32(comm, x.d_buf, x.count, lane);            break; }
 *     } while (x.kind != TSDRGPU_BAND_DONE);
 * The all-gather (count floats PER RANK, rank r's part at r * count) only occurs with autoshift: the roll moves rows
 * across bands, so every rank receives the frames once (F * height * width floats per batch over xGMI; 27 MB per frame at
 * 2962 x 2250).  Frames and per-frame records are bit-identical to tsdrgpu_postproc_run with the same parameters in its
 * default, contract-exact mode, i.e. to the reference; the library-default order without autoshift gives what
 * _band_begin / _band_advance give. */
#define TSDRGPU_BAND_DONE 0
#define TSDRGPU_BAND_SUM_F64 1
#define TSDRGPU_BAND_MAX_F32 2
#define TSDRGPU_BAND_ALLGATHER_F32 3
typedef struct tsdrgpu_band_exchange {
    int kind;      /* TSDRGPU_BAND_*: what the caller has to do with d_buf before the next step */
    void *d_buf;   /* device buffer, reduced / gathered IN PLACE over the ranks */
    int64_t count; /* elements (doubles or floats; per rank for the all-gather) */
} tsdrgpu_band_exchange_t;
int tsdrgpu_postproc_band_open(tsdrgpu_postproc_t *pp, const float *d_band, int nframes, int width, int height, const int *edges, int nbands,
                               int band_index, const tsdrgpu_pp_params_t *params);
int tsdrgpu_postproc_band_step(tsdrgpu_postproc_t *pp, float *d_out_band, tsdrgpu_band_exchange_t *x, tsdrgpu_pp_frameinfo_t *h_info);
/* The per-frame record of the last run, without a host synchronisation: packs nframes tsdrgpu_pp_frameinfo_t
 * into the caller's DEVICE buffer on the COMPUTE lane (download it on any lane behind an event). */
int tsdrgpu_postproc_info_pack(tsdrgpu_postproc_t *pp, tsdrgpu_pp_frameinfo_t *d_info, int nframes);
/* strips of the last frame run (after blur + markers), for stage-level tests */
int tsdrgpu_postproc_strips(tsdrgpu_postproc_t *pp, float *h_colsum, float *h_rowsum); /* syncs */

/* ---- a9..a12: FFT autocorrelation ---------------------------------------------- */
/* fft_perform, fft.c:96-176: in-place complex FFT of n = 2^m points on
 * interleaved float32 (forward scaled by 1/n, inverse unscaled).  Asynchronous like the rest. */
int tsdrgpu_fft(tsdrgpu_t *g, float *d_iq, uint32_t n, int inverse);
/* The same transform in the reference's own arithmetic (see tsdrgpu_autocorr_set_exact): results bit-identical
 * to fft_perform.  Keeps one twiddle table per context (rebuilt on the host when n changes). */
int tsdrgpu_fft_exact(tsdrgpu_t *g, float *d_iq, uint32_t n, int inverse);

/* frameratedetector_runontodata numerics (frameratedetector.c:87-126):
 * fft_autocorrelation (fft.c:49-64) of each capture window, then accummulate
 * (frameratedetector.c:34-62) into the frame- and line-lag plots. */
int tsdrgpu_autocorr_create(tsdrgpu_t *g, tsdrgpu_autocorr_t **out, uint32_t samplerate);
void tsdrgpu_autocorr_destroy(tsdrgpu_autocorr_t *ac);
int tsdrgpu_autocorr_reset(tsdrgpu_autocorr_t *ac); /* PARAM_AUTOCORR_PLOTS_RESET */
/* geometry: lags [frame_lo, frame_lo+frame_len), [line_lo, ...), capture size
 * (3.1*fs/55) and the FFT length n = largest 2^m <= capture size */
int tsdrgpu_autocorr_geometry(tsdrgpu_autocorr_t *ac, int32_t *frame_lo, int32_t *frame_len,
                              int32_t *line_lo, int32_t *line_len, uint32_t *capture,
                              uint32_t *fft_n);
/* d_in: magnitude samples (in_is_iq=0) or interleaved IQ (demod fused).
 * Window w starts at sample w*stride.  mode 0: running mean, bit-for-bit the
 * reference's recurrence; mode 1: plain sums (for sharded runs: all-reduce
 * the sums, then tsdrgpu_autocorr_finalize_sums). */
int tsdrgpu_autocorr_run(tsdrgpu_autocorr_t *ac, const float *d_in, int in_is_iq, int64_t stride,
                         int nwindows, int mode);
/* on != 0: this object's work is queued on the context's BACKGROUND lane (lowest priority), so it overlaps the
 * frame path and fills its gaps; each run still waits for everything queued on the COMPUTE lane before it.
 * tsdrgpu_sync() waits for the COMPUTE, SIDE and BACKGROUND lanes. */
int tsdrgpu_autocorr_set_async(tsdrgpu_autocorr_t *ac, int on);
/* Exact mode.  The default autocorrelation is a different FFT algorithm than the reference's and agrees with it
 * to ~1e-6 of the plot maximum; where a plot holds exact mathematical ties (R[j] == R[N-j] inside the 8 MS/s
 * frame-lag window) the reference's argmax is its own rounding noise.  With `on` the transforms are done in the
 * reference's arithmetic (fft.c:96-176: radix-2 DIT, f64 butterflies on f32 storage, the sequential twiddle
 * recurrence) and plots, argmax and tsdrgpu_autocorr_last_corr are BIT-IDENTICAL to the reference's.  About 7x
 * slower (0.24 ms per 2^22-sample window); builds a table of N-1 f64 twiddle pairs (64 MB at 100 MS/s) on first use. */
int tsdrgpu_autocorr_set_exact(tsdrgpu_autocorr_t *ac, int on);
/* Certified mode: the float32 transform with a GUARANTEE that the argmax of each plot is the reference's, and a
 * replay in the reference's arithmetic whenever that cannot be guaranteed.  SURVEY 8(d) asks of the plots <= 1e-4*max
 * per lag AND the identical argmax lag (what the host turns into frame rate and line count, Main.java:1301-1303,
 * PlotVisualizer.java:233-236).  The float32 transform keeps every accumulated value within
 * (TSDRGPU_AC_CERT_KAPPA / 2) * R0 of the reference's (R0 = the accumulated lag-0 value, the largest value a
 * correlation of magnitudes holds; measured distance ~1e-7 * R0, tests/test_gpu_certify.py asserts the bound on every
 * case it runs), so an argmax whose value exceeds the runner-up's — the largest value at any OTHER lag — by more than
 * TSDRGPU_AC_CERT_KAPPA * R0 is the reference's argmax.  The argmax kernels compute runner-up and R0 on the fly
 * (tsdrgpu_autocorr_certificate).  When the test fails — exact mathematical ties (R[j] == R[N-j] inside the 8 MS/s
 * frame-lag window), noise-like or flat plots — the windows of the epoch (everything run since the last
 * tsdrgpu_autocorr_reset) are replayed through the exact form (tsdrgpu_autocorr_promote): plots, argmax and
 * last correlation are then bit-identical to the reference's, and the rest of the epoch runs exact.
 *   mode 1: the library retains the windows — the first fft_n samples of each, 4 bytes a sample (magnitudes; from IQ input
 *           the sum of squares am_demod forms, left there by the transform's own first trip), in a ring of
 *           retain_bytes in HBM (0 = a quarter of the device's free memory at the time of the call, at most 32 GiB: 2048
 *           windows of 2^22 samples, 116 s of real-time signal at 100 MS/s), allocated in segments of >= 32 windows ahead of need
 *           by a thread of the library's own (tsdrgpu_autocorr_retention); the float32 transform reads the ring.  An
 *           epoch that outgrows the ring is promoted (one exact replay) and continues exact: such epochs cost what the
 *           exact form costs from then on, shorter ones (a sweep over a recording, a GUI that resets on every parameter
 *           change, two minutes of live signal) what the fast form costs.
 *   mode 2: the caller retains them: every buffer passed to tsdrgpu_autocorr_run since the last reset must stay valid
 *           and unchanged until the next reset (a recording resident in HBM).  No copy is made.
 *   mode 0: off (plain float32 form; the certificate is still computed and reported).
 * Switching modes mid-epoch resets the object.  tsdrgpu_autocorr_last_corr returns the reference's bits in either
 * certified mode (the last window is transformed once more in the exact form when the epoch is still fast).
 * THE PREMISE IS AN ERROR MODEL, NOT A THEOREM, AND IT IS CHECKED AT RUN TIME.  KAPPA / 2 = 4e-6 is ~40x the rounding
 * error c * eps * sqrt(log2 N) * R0 that the error model of DESIGN.md section 2 predicts and every measurement shows
 * (2e-8 .. 3.6e-7 * R0); a worst-case bound of the same form (eps * log2 N * R0 for the inverse transform alone) is
 * already larger than that, so no choice of KAPPA makes the certificate a proof.  Therefore every check_every-th plot
 * update (TSDRGPU_AC_CHECK_EVERY, default 16, and always the first one after this call) also runs the newest retained
 * window through the reference's arithmetic and compares the two on the device: max |float32 - exact| over the lag windows
 * must be <= (KAPPA / 2) * (the window's lag-0 value), otherwise that update's certificate FAILS like any other and the
 * epoch is replayed exactly.  tsdrgpu_ac_certificate_t reports the check; TSDR_GPU_AUTOCORR=exact (the engine) or
 * tsdrgpu_autocorr_set_exact remain for hosts that want the reference's bits unconditionally. */
#define TSDRGPU_AC_CERT_KAPPA 8e-6
int tsdrgpu_autocorr_set_certify(tsdrgpu_autocorr_t *ac, int mode, size_t retain_bytes);
/* mode 1's retention ring: its capacity in windows when fully allocated, the part of it that is allocated right now (the ring
 * grows in segments of >= 32 windows that a background thread allocates ahead of need: fresh device memory costs 40-80 ms per GiB at
 * first use, which neither the caller's thread nor the detector's lane should wait for), the position of the epoch's next
 * window in it, and whether the epoch has been promoted (or the object is in exact mode) — i.e. which transform the next
 * window will go through.  A window that would not fit into what is allocated promotes the epoch. */
int tsdrgpu_autocorr_retention(tsdrgpu_autocorr_t *ac, int *ring_windows, int *ring_ready, int *retained_windows, int *epoch_is_exact);
/* The allocator keeps two segments beyond the one in use ready.  A host that knows its epoch will hold `windows` windows (a
 * sweep over a recording) asks for that room at once and may wait up to wait_ms for it (0: just ask). */
int tsdrgpu_autocorr_retention_reserve(tsdrgpu_autocorr_t *ac, int windows, int wait_ms);
/* Replays the current epoch in the reference's arithmetic (no-op when it already is exact).  In a sharded run
 * (mode 1 sums + tsdrgpu_autocorr_allreduce) it leaves this rank's exact sums: repeat the all-reduce afterwards. */
int tsdrgpu_autocorr_promote(tsdrgpu_autocorr_t *ac);
/* The same replay in bounded steps, for a streaming host that must not stall its queue behind a long epoch's replay (2048
 * retained windows of 2^22 samples = 0.28 s of transforms): the first call opens it, every call replays up to max_windows
 * windows, *h_remaining = windows still to go.  While it is > 0 the plots are partial — tsdrgpu_autocorr_run and the
 * argmax calls return TSDRGPU_ESTATE — and the host skips its capture windows, as the reference's detector thread
 * does whenever it is busy (frameratedetector.c:128-187 takes a window only when it is idle). */
int tsdrgpu_autocorr_promote_step(tsdrgpu_autocorr_t *ac, int max_windows, int *h_remaining);
typedef struct tsdrgpu_ac_certificate {
    int frame_certified, line_certified; /* 1: this plot's argmax is provably the reference's */
    int exact_epoch;                      /* 1: the plots are in the reference's own arithmetic (bit-identical) */
    int promotions;                       /* epochs promoted by this object so far */
    double frame_best, frame_runner_up, line_best, line_runner_up; /* plot values */
    double r0;                            /* accumulated lag-0 value */
    double margin;                        /* TSDRGPU_AC_CERT_KAPPA * r0: what best - runner_up must exceed */
    /* the runtime check of the premise (see above): did this update carry one, did it hold, what it measured */
    int premise_checked, premise_ok;
    double premise_err, premise_r0;       /* max |float32 - exact| over the lags of the checked window; that window's lag-0 value */
    long premise_checks, premise_failures; /* of this object so far */
} tsdrgpu_ac_certificate_t;
/* the certificate that came with the last argmax collected (tsdrgpu_autocorr_argmax / _argmax_result) */
int tsdrgpu_autocorr_certificate(tsdrgpu_autocorr_t *ac, tsdrgpu_ac_certificate_t *out);
/* argmax; if a plot is not certified: promote, argmax again.  *h_promoted = 1 when that happened.  Synchronises. */
int tsdrgpu_autocorr_argmax_certified(tsdrgpu_autocorr_t *ac, int32_t *frame_idx, int32_t *line_idx, int *h_promoted);
/* Transform plan of the default (non-exact) form.  3 (default where it applies: capture windows of 2^17 ..
 * 2^23 samples, i.e. 2.4 .. 297 MS/s): the packed window of nh = N/2 complex points is treated as an
 * (nh/4096) x 4096 matrix — column DFTs, then the row pairs (k, nh-k) with the packed-real split, 1/N, the
 * magnitude and the inverse row DFTs fused in LDS, then column DFTs storing only the two lag windows: three
 * trips over HBM per window.  5: the Stockham radix-128 plan of round 1 (three passes each way, the middle two
 * fused), also what other sizes fall back to.  Same tolerance either way. */
int tsdrgpu_autocorr_set_plan(tsdrgpu_autocorr_t *ac, int trips);
int tsdrgpu_autocorr_plots(tsdrgpu_autocorr_t *ac, double *h_frame, double *h_line,
                           uint64_t *h_calls); /* syncs */
/* the same copies queued on the object's lane without waiting: h_* must be pinned and stay valid until an event
 * recorded on that lane (tsdrgpu_autocorr_lane) behind this call has completed */
int tsdrgpu_autocorr_plots_async(tsdrgpu_autocorr_t *ac, double *h_frame, double *h_line, uint64_t *h_calls);
/* a device-side copy (frame plot first, then the line plot) taken on the object's lane by a kernel; copy it home on
 * another lane once an event recorded behind this call has fired (the streaming engine's plot thread does) */
int tsdrgpu_autocorr_plots_snapshot(tsdrgpu_autocorr_t *ac, const double **d_snapshot, uint64_t *h_calls);
/* device plots: frame_len + line_len doubles, contiguous (frame first).  One more double sits behind them: the
 * accumulated lag-0 value R0, the scale of the argmax certificate; *count does not include it. */
int tsdrgpu_autocorr_device_plots(tsdrgpu_autocorr_t *ac, double **d_plots, int64_t *count);
/* The buffer a caller that runs its OWN collective (instead of tsdrgpu_autocorr_allreduce) must sum over the ranks after
 * tsdrgpu_autocorr_run(mode 1): the same pointer, *count = frame_len + line_len + 1 — the lags and R0.  Reducing only
 * the lags would leave R0 rank-local while tsdrgpu_autocorr_finalize_sums divides it by the global window count: the
 * certificate's margin KAPPA * R0 would come out `world` times too small. */
int tsdrgpu_autocorr_device_sums(tsdrgpu_autocorr_t *ac, double **d_sums, int64_t *count);
/* after the sum over the ranks: divides lags and R0 by the global window count; calls := total_windows */
int tsdrgpu_autocorr_finalize_sums(tsdrgpu_autocorr_t *ac, uint64_t total_windows);
/* argmax (lowest index wins ties, PlotVisualizer.java:233-236); syncs */
int tsdrgpu_autocorr_argmax(tsdrgpu_autocorr_t *ac, int32_t *frame_idx, int32_t *line_idx);
/* The same in two halves, so that the host does not have to wait for the device once per plot update:
 * _async queues the argmax and the copy of its two indices, _result waits for that copy only (work queued
 * after _async keeps the device busy meanwhile).  One outstanding request per object. */
int tsdrgpu_autocorr_argmax_async(tsdrgpu_autocorr_t *ac);
int tsdrgpu_autocorr_argmax_result(tsdrgpu_autocorr_t *ac, int32_t *frame_idx, int32_t *line_idx);
/* the raw correlation of the LAST window run (2*n floats), for stage tests */
int tsdrgpu_autocorr_last_corr(tsdrgpu_autocorr_t *ac, const float **d_corr, uint32_t *n);

int tsdrgpu_autocorr_lane(tsdrgpu_autocorr_t *ac); /* TSDRGPU_LANE_COMPUTE, or _BACKGROUND after tsdrgpu_autocorr_set_async */

/* ---- the sweep across GPUs — one process per GPU, one exchange over RCCL / xGMI ------------------
 * Capture windows are independent; the reference only forms their running mean (accummulate,
 * frameratedetector.c:51-60).  Rank r of `world` takes windows r, r+world, ... with tsdrgpu_autocorr_run(mode 1)
 * (plain per-lag sums) and tsdrgpu_autocorr_allreduce then leaves the global plots — sum over all ranks / total
 * window count — on every rank: ncclAllReduce(ncclDouble, ncclSum) of frame_len + line_len values queued on the
 * autocorrelation's own lane (no host synchronisation), then tsdrgpu_autocorr_finalize_sums.  Equal to the
 * single-rank running mean up to f64 rounding (1e-15 relative).
 * Bootstrap like any NCCL program: rank 0 calls tsdrgpu_rccl_unique_id and ships the 128 bytes to the other
 * ranks by whatever means the host has (MPI, a TCP store, torch.distributed's broadcast); every rank then calls
 * tsdrgpu_comm_create.  RCCL is dlopen'ed at first use: no link-time dependency, never loaded on one GPU. */
#define TSDRGPU_RCCL_ID_BYTES 128
typedef struct tsdrgpu_comm tsdrgpu_comm_t;
int tsdrgpu_rccl_unique_id(void *id128);
int tsdrgpu_comm_create(tsdrgpu_t *g, tsdrgpu_comm_t **out, int world, int rank, const void *id128);
void tsdrgpu_comm_destroy(tsdrgpu_comm_t *c);
int tsdrgpu_comm_count(tsdrgpu_comm_t *c, int *ranks, int *my_rank); /* ncclCommCount / ncclCommUserRank: what RCCL itself sees */
int tsdrgpu_comm_allreduce_f64(tsdrgpu_comm_t *c, double *d_buf, int64_t count, int lane); /* in place, ncclSum */
int tsdrgpu_comm_allreduce_f32max(tsdrgpu_comm_t *c, float *d_buf, int64_t count, int lane); /* in place, ncclMax */
int tsdrgpu_comm_broadcast_f32(tsdrgpu_comm_t *c, float *d_buf, int64_t count, int root, int lane);      /* in place */
int tsdrgpu_comm_allgather_f32(tsdrgpu_comm_t *c, float *d_buf, int64_t count_per_rank, int lane);       /* in place: rank r's part at r*count */
int tsdrgpu_autocorr_allreduce(tsdrgpu_autocorr_t *ac, tsdrgpu_comm_t *c, uint64_t total_windows);

/* ---- a13/a14: super-bandwidth stitch --------------------------------------------- */
/* superb_ondataready, superbandwidth.c:121-152 (complex_to_abs_diff :67-81,
 * superb_bestfit :83-119, fft_crosscorrelation fft.c:69-93).  d_hops: nhops
 * device buffers of 2*gathered floats.  d_out: nhops*2*n floats with n = largest
 * 2^m <= gathered.  h_offsets[nhops]: best offsets in floats.  Synchronises.
 * Four hops (the reference's SUPER_HOPS_TO_MAKE) of 2^16 .. 2^23 points take the THREE-TRIP plan (csrc/fft4step.h): the
 * hop buffers are read only.  Other shapes, and tsdrgpu_superb_set_plan(g, 0), take the pass-per-radix plan, which — like
 * the reference, superbandwidth.c:138-144 — leaves every hop buffer holding its spectrum (values the reference never
 * reads again: the next gather overwrites them, superbandwidth.c:225-236).  Same results either way: offsets identical,
 * the stitched signal within 1e-4 * max. */
int tsdrgpu_superb_stitch(tsdrgpu_t *g, float *const *d_hops, int nhops, int gathered,
                          int samples_in_frame, float *d_out, int32_t *h_offsets,
                          uint32_t *h_total);
/* trips = 3 (default): the three-trip plan where it applies; 0: the pass-per-radix plan always (A/B, and what the sharded
 * form below is bit-identical to). */
int tsdrgpu_superb_set_plan(tsdrgpu_t *g, int trips);
/* SURVEY 8(e) row 3 — the stitch with ONE HOP PER GPU (rank r holds hop r; superbandwidth.c:121-152).  Everything but
 * two steps is per hop; the two steps are exchanges the caller makes between the phases:
 *   _reference   hop 0's rank transforms its abs-diff signal; *d_ref (n floats) is broadcast from that rank
 *                (tsdrgpu_comm_broadcast_f32, root = the rank of hop 0)
 *   _spectrum    this rank's offset against hop 0 (cross-correlation peak), rotation and transform; its spectrum lands in
 *                slot my_hop of *d_spectra, which is all-gathered in place (tsdrgpu_comm_allgather_f32, n floats per hop);
 *                the hop buffer is left holding the spectrum like the reference's
 *   _finish      the inverse transform of the concatenated spectra (on every rank that calls it) -> d_out
 * The kernels and their order are those of tsdrgpu_superb_stitch's pass-per-radix plan (tsdrgpu_superb_set_plan(g, 0)), so
 * offsets and the stitched signal are bit-identical to the single-GPU call on that plan; against its three-trip plan the
 * offsets are identical and the signal agrees to rounding (within 1e-4 * max of the reference either way).  Measured worth: the exchanges (32 + 256 MB at 4 x 2^23 samples) cost more than
 * the 0.65 ms the whole stitch takes on one GPU (three-trip plan; 1.25 ms on the plan the shards run) — the form exists for hosts whose hops already live on different GPUs. */
typedef struct tsdrgpu_superb_shard tsdrgpu_superb_shard_t;
int tsdrgpu_superb_shard_create(tsdrgpu_t *g, tsdrgpu_superb_shard_t **out, int nhops, int my_hop, int gathered, int samples_in_frame);
void tsdrgpu_superb_shard_destroy(tsdrgpu_superb_shard_t *sh);
int tsdrgpu_superb_shard_reference(tsdrgpu_superb_shard_t *sh, const float *d_my_hop, float **d_ref, int64_t *n_floats);
int tsdrgpu_superb_shard_spectrum(tsdrgpu_superb_shard_t *sh, float *d_my_hop, float **d_spectra, int64_t *n_floats_per_hop,
                                  int32_t *h_my_offset);
int tsdrgpu_superb_shard_finish(tsdrgpu_superb_shard_t *sh, float *d_out, uint32_t *h_total);
/* The same stitch with every transform in the reference's own arithmetic (see tsdrgpu_autocorr_set_exact): hop
 * offsets and the stitched signal bit-identical to superb_ondataready. */
int tsdrgpu_superb_stitch_exact(tsdrgpu_t *g, float *const *d_hops, int nhops, int gathered, int samples_in_frame,
                                float *d_out, int32_t *h_offsets, uint32_t *h_total);

/* ---- the components next to the hot path ------------------------- */
/* f1: the RawFile plugin's sample formats decoded on the device
 * (TSDRPlugin_RawFile/src/TSDRPlugin_RawFile.c:241-261): type 0 float32 (copy), 1 int8 (/128.0),
 * 2 int16 (/32767.0), 3 uint8 ((v-128)/128.0), 4 uint16 ((v-32767)/32767.0); n = number of values
 * (2 per IQ sample).  Bit-exact with the plugin's double division + float store. */
int tsdrgpu_decode_samples(tsdrgpu_t *g, const void *d_raw, int type, float *d_out, int64_t n);

/* a3: dsp_autogain_t.snr of a frame as dsp_autogain_run would leave it (TempestSDR/src/dsp.c:69-93: mean over the
 * non-sentinel sum / all pixels, deviations over every pixel); f64 tree sums, ~1e-12 relative to the sequential loop.
 * The reference never reads the field (dsp.c:234 is commented out), so a post-processing run produces it only when asked:
 *   tsdrgpu_postproc_set_snr(pp, 1)  every tsdrgpu_postproc_run / _begin+_finish / _begin_minmax+_finish from now on also
 *                                    leaves one value per frame — of the frames autogain reads in that stage order (the
 *                                    input frames, or the low-passed / corrected ones when autogain runs after them):
 *                                    one more read of those frames, queued beside the sync chain.  Band runs do not.
 *   tsdrgpu_postproc_snr(pp, h, n)   the first n values of the last run (ESTATE if it produced fewer).  Synchronises.
 *   tsdrgpu_frame_snr                the same value for any frame on demand (bit-identical to the by-product).  Synchronises. */
int tsdrgpu_postproc_set_snr(tsdrgpu_postproc_t *pp, int on);
int tsdrgpu_postproc_snr(tsdrgpu_postproc_t *pp, float *h_snr, int nframes);
int tsdrgpu_frame_snr(tsdrgpu_t *g, const float *d_frame, int64_t npixels, float *h_snr);

/* f3: frame -> packed 0x00RRGGBB exactly like the JNI shim (JavaGUI/jni/TSDRLibraryNDK.c:222-276):
 * gray = (int)(v*255) for 0 < v <= 1, black/white outside, the PIXEL_SPECIAL_VALUE_* debug colours,
 * TRANSPARENT keeps the pixel d_rgb already holds; `inverted` as the GUI's invert option. */
int tsdrgpu_frame_to_rgb(tsdrgpu_t *g, const float *d_frame, int32_t *d_rgb, int64_t npixels, int inverted);

/* f2: mode detection from the two plots' argmax (the GUI's logic, PlotVisualizer.java:200-247,
 * Main.java:1233-1277,1301-1303,1346-1350, VideoMode.java:163-190): feed one (frame, line) argmax pair
 * per plot update; `accepted` becomes 1 once the same (fps, height) was seen 3 times before. */
typedef struct tsdrgpu_modedetect tsdrgpu_modedetect_t;
typedef struct tsdrgpu_detection {
    int frame_lag, line_lag;   /* offset + argmax index, in samples */
    double framerate;          /* samplerate / frame_lag */
    double linerate;           /* samplerate / line_lag */
    int height;                /* round(frame_lag / line_lag): total lines */
    double pixelrate;          /* width*height*framerate as tsdr_setresolution would derive it */
    int seen, accepted;
    int mode_id;               /* closest pre-registered video mode, -1 if none */
    char mode_name[48];
    int mode_width, mode_height;
    double mode_refresh;
} tsdrgpu_detection_t;
int tsdrgpu_modedetect_create(tsdrgpu_modedetect_t **out);
void tsdrgpu_modedetect_destroy(tsdrgpu_modedetect_t *d);
void tsdrgpu_modedetect_reset(tsdrgpu_modedetect_t *d);
int tsdrgpu_modedetect_feed(tsdrgpu_modedetect_t *d, int frame_offset, int frame_idx, int line_offset,
                            int line_idx, uint32_t samplerate, tsdrgpu_detection_t *out);

/* f4: plot decimation for display — what PlotVisualizer.populateData
 * (JavaGUI/src/martin/tempest/gui/PlotVisualizer.java:200-247) computes from a plot of `size` doubles for a
 * widget `nwidth` pixels wide: h_visdata[px] = maximum of the lags drawn in pixel column px (columns no lag
 * maps to repeat the previous column), before the y scaling of :245-246; lowest/highest = the pair handed to
 * scale_y (:243); max_index = getMaxIndex() (:226-229).  `scale` is the state of the widget's
 * ZoomableXScale (gui/scale/ZoomableXScale.java:133-149: value_to_pixel_absolute(v) =
 * (int)((v-min_value)*one_val_in_pixels) - offset_px, pixels_to_value_absolute(px) =
 * px*one_px_in_values + offset_val + min_value); NULL = the unzoomed scale (tsdrgpu_plotscale_default).
 * d_data is a device pointer, e.g. from tsdrgpu_autocorr_device_plots; only nwidth doubles cross PCIe. */
typedef struct tsdrgpu_plotscale {
    double one_val_in_pixels, one_px_in_values, offset_val, min_value;
    int offset_px;
} tsdrgpu_plotscale_t;
void tsdrgpu_plotscale_default(int size, int nwidth, tsdrgpu_plotscale_t *s);
int tsdrgpu_plot_columns(tsdrgpu_t *g, const double *d_data, int size, int nwidth, const tsdrgpu_plotscale_t *scale,
                         double *h_visdata, double *h_lowest, double *h_highest, int *h_max_index);

#ifdef __cplusplus
}
#endif
#endif /* TSDRGPU_H_ */
