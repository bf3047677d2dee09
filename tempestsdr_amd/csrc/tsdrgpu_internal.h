// tsdrgpu_internal.h — shared by the .hip translation units of libtsdrgpu.so
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/tsdrgpu.h"

#define TSDR_WAVE 64

// Debug aid, OFF unless the process sets TSDRGPU_REDZONES=1 (or =2: report and go on): every device allocation the library makes
// — its own scratch, plots, rings and the buffers behind tsdrgpu_alloc — then sits between two 4 KiB zones of 0xFF bytes (NaN as
// float32 / float64, -1 as an integer), and tsdrgpu_free / every internal free checks that both are untouched: a kernel that writes
// outside the scratch the library sized for it is reported on stderr with the allocation's size (and the process aborts unless =2),
// one that reads outside computes with NaN and fails the parity tests.  tests/conftest.py does the same for the buffers the TESTS
// allocate; this covers the ones they never see.  Unset, hipMalloc / hipFree below are the runtime's own, one branch away.
hipError_t tsdr_redzone_malloc(void **p, size_t n);
hipError_t tsdr_redzone_free(void *p);
template <class T>
static inline hipError_t tsdr_malloc_t(T **p, size_t n)
{
    return tsdr_redzone_malloc((void **)p, n);
}
#define hipMalloc(p_, n_) tsdr_malloc_t((p_), (n_))
#define hipFree(p_) tsdr_redzone_free((p_))

// stage ids for the optional event profiler (tsdrgpu_profile_*)
enum ProfStage {
    PROF_DEMOD = 0,
    PROF_RS_CARRY,
    PROF_RS_AREA,
    PROF_RS_NEAREST,
    PROF_FRAME_STATS,
    PROF_FRAME_REDUCE,
    PROF_CHAIN,
    PROF_FRAME_PASS,
    PROF_FFT_PASS,
    PROF_AC_SPLIT,
    PROF_AC_COLS,
    PROF_AC_ROWS,
    PROF_ACCUMULATE,
    PROF_SUPERB_MISC,
    PROF_ARGMAX,
    PROF_EXTRAS,
    PROF_COUNT
};

struct ProfSpan {
    int stage;
    hipEvent_t a, b;
};

struct tsdrgpu {
    int device;
    hipStream_t stream;
    hipStream_t stream2;  // side stream: the autocorrelation can run beside the frame path
    hipStream_t up, down; // copy lanes (tsdrgpu_upload_lane / tsdrgpu_download_lane)
    hipStream_t bg;       // background lane, lowest priority: an asynchronous autocorrelation fills the gaps of the frame path
    hipEvent_t fork;      // orders stream2 behind what is already queued on `stream`
    hipEvent_t t0, t1;
    hipDeviceProp_t prop;
    // scratch kept between calls of tsdrgpu_fft (grown on demand)
    void *fft_ws;
    size_t fft_ws_bytes;
    void *superb_ws;    // scratch of tsdrgpu_superb_stitch (grown on demand)
    size_t superb_ws_bytes;
    int *superb_h_off;  // pinned: the stitch's hop offsets, written by the peak kernel itself
    int superb_passes;  // tsdrgpu_superb_set_plan(0): the pass-per-radix plan even where the three-trip one applies
    void *fftx_tw;      // twiddle table of the last tsdrgpu_fft_exact size (double2[n-1])
    uint32_t fftx_n;
    // profiler
    int prof_on;
    ProfSpan *spans;
    int nspans, cap_spans;
};

// Profiling-aware launch: with the profiler on, the kernel's own dispatch carries a start and a
// stop event (hipExtLaunchKernelGGL), so per-kernel durations are measured without extra barrier
// packets between launches; with it off this is a plain launch.
void prof_pair(tsdrgpu_t *g, int stage, hipEvent_t *a, hipEvent_t *b);
hipStream_t tsdr_lane_stream(tsdrgpu_t *g, int lane);
#define TSDR_LAUNCH(g_, stage_, stream_, kernel_, grid_, block_, ...)                                        \
    do {                                                                                                    \
        hipEvent_t pa_ = nullptr, pb_ = nullptr;                                                            \
        prof_pair((g_), (stage_), &pa_, &pb_);                                                              \
        hipExtLaunchKernelGGL(kernel_, dim3(grid_), dim3(block_), 0, (stream_), pa_, pb_, 0, __VA_ARGS__);  \
    } while (0)

// Error text is kept per host thread (tsdrgpu_last_error returns the calling thread's): several threads may drive one
// context, each on its own lane, and a failing call on one must not garble what another is reading.
char *tsdr_errbuf(void);
static inline int tsdr_fail(tsdrgpu_t *g, int code, const char *what, const char *detail)
{
    if (g) snprintf(tsdr_errbuf(), 512, "%s: %s", what, detail ? detail : "");
    return code;
}

#define HIP_TRY(g, call)                                                              \
    do {                                                                              \
        hipError_t e__ = (call);                                                      \
        if (e__ != hipSuccess) return tsdr_fail((g), TSDRGPU_EHIP, #call, hipGetErrorString(e__)); \
    } while (0)

#define KERNEL_CHECK(g, name)                                                         \
    do {                                                                              \
        hipError_t e__ = hipGetLastError();                                           \
        if (e__ != hipSuccess) return tsdr_fail((g), TSDRGPU_EHIP, name, hipGetErrorString(e__)); \
    } while (0)

static inline unsigned ceil_div_u(unsigned long long a, unsigned long long b)
{
    return (unsigned)((a + b - 1) / b);
}

// A small pinned staging area for tables that kernels read (chunk tables, ...),
// recycled round-robin; an event per slot guards reuse.
struct StagingRing {
    static const int SLOTS = 4;
    void *h[SLOTS];
    void *d[SLOTS];
    size_t cap[SLOTS];
    hipEvent_t ev[SLOTS];
    bool used[SLOTS];
    int next;
};

int staging_init(tsdrgpu_t *g, StagingRing *r);
void staging_free(StagingRing *r);
// returns slot index or <0; after filling h[slot], call staging_push
int staging_acquire(tsdrgpu_t *g, StagingRing *r, size_t bytes);
int staging_push(tsdrgpu_t *g, StagingRing *r, int slot, size_t bytes);  // async H2D + nothing else
int staging_release(tsdrgpu_t *g, StagingRing *r, int slot);            // record event after the consumer launch

// FFT engine (tsdrgpu_fft.hip), used by autocorrelation and super-bandwidth
struct FftPlan;

// Exact replica of the reference's FFT arithmetic (tsdrgpu_fftx.hip), the optional exact mode of the autocorrelation
int fftx_build_table(tsdrgpu_t *g, uint32_t n, double2 **d_tw);
int fftx_autocorr(tsdrgpu_t *g, hipStream_t st, const float *d_in, int in_is_iq, long long stride, int cnt, uint32_t n,
                  const double2 *d_tw, float2 *z, float *mag, int frame_lo, int frame_len, int line_lo, int line_len, double *d_plots,
                  unsigned long long calls_before, int mode, int full_w);
int fftx_correlate(tsdrgpu_t *g, hipStream_t st, const float *d_in, int in_is_iq, long long stride, int cnt, uint32_t n,
                   const double2 *d_tw, float2 *z, float *mag);
int fftx_retain(tsdrgpu_t *g, hipStream_t st, const float *src, int is_iq, long long stride, int cnt, uint32_t n, float *dst);
// tsdrgpu_extras.hip: dsp_autogain_t.snr of F frames (dsp.c:69-93) queued on st; d_parts = tsdr_snr_part_doubles() doubles per frame
size_t tsdr_snr_part_doubles(void);
int tsdr_snr_batch(tsdrgpu_t *g, hipStream_t st, const float *d_frames, long long stride, long long npixels, int F, double *d_parts, float *d_snr);
int fftx_perform(tsdrgpu_t *g, hipStream_t st, const float2 *d_z, float2 *d_work, uint32_t n, const double2 *d_tw, int inverse);
