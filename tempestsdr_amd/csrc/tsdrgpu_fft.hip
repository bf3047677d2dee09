// tsdrgpu_fft.hip — FFT engine, autocorrelation and lag-window accumulation for
// gfx950: fft_perform / fft_autocorrelation (TempestSDR/src/fft.c:49-64,96-176)
// and accummulate (frameratedetector.c:34-62).
//
// Engine (round 1): out-of-place Stockham autosort, radix 16/8/4/2 butterflies
// held in registers, one pass over HBM per radix stage, ping-pong between two
// buffers; many windows per launch (blockIdx.y).  Twiddles come from
// sincospif() of an exactly representable dyadic fraction.  Data is float32
// complex like the reference's storage; the reference rounds to f32 after every
// radix-2 stage with f64 butterflies, so the two differ at the 1e-7 level of
// the largest term (tolerance stated in tests/test_gpu_autocorr.py).
#include "tsdrgpu_internal.h"
#include "fft4step.h"

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

// windows per launch of the float32 plan: 9 of 2^22 points or more (see ac_run_fast), 18 of shorter ones — 9 windows of 2^20
// points are a quarter of the workgroups, launches that are mostly ramp and drain.  Measured on bench.py --config 1 (25 MS/s,
// 70 windows of 2^20 per pass): 9 -> 79.0 GS/s, 18 -> 84.5, 36 -> 81.8, 64 -> 81.6 (longer launches starve the sync chain on
// the side lane, as at 2^22).  TSDRGPU_AC_SUBBATCH overrides for experiments: 1..64
static int ac_subbatch(uint32_t n)
{
    static const int forced = [] {
        const char *e = getenv("TSDRGPU_AC_SUBBATCH");
        const int v = e ? atoi(e) : 0;
        return v < 1 ? 0 : (v > 64 ? 64 : v);
    }();
    if (forced) return forced;
    return n < (1u << 22) ? 18 : 9;
}

// one tsdrgpu_autocorr_run call of the current epoch, as the certified mode remembers it for an exact replay
// (is_iq is the input KIND as the exact form's first trip takes it: 0 real magnitudes, 1 interleaved IQ, AC_KIND_SUMSQ the ring
// as k_ac_cols_retain fills it — am_demod's re*re + im*im, the root still to be taken; one float per sample like kind 0)
#define AC_KIND_SUMSQ 3
#define AC_KIND_FLOATS(kind_) ((kind_) == 1 ? 2 : 1)
struct AcLogRec {
    const float *src;
    int is_iq;
    long long stride;
    int nwindows;
    int mode;
};

// what k_argmax_final leaves in pinned memory: the argmax pair and its certificate
struct AcArgHost {
    int idx[2];
    int certified[2];
    double best[2], second[2];
    double r0, margin;
    int premise_checked;   // this update carried a runtime check of the certificate's premise (ac_premise_check)
    int premise_ok;        // ... and the float32 window lay within (KAPPA / 2) * R0 of the reference's arithmetic
    double premise_err, premise_r0;  // max |fast - exact| over the lag windows of the checked window, its lag-0 value
};

// The retention ring of the certified mode (tsdrgpu_autocorr_set_certify mode 1) in SEGMENTS that a background thread
// allocates ahead of need.  Fresh device memory costs 40-80 ms per GiB on MI355X the first time it is used (measured,
// scripts/micro/malloc_bench.hip: hipMalloc of 32 GiB takes 1.4-2.6 s) — in one piece at detector start that was a
// multi-second stall of the caller's thread, and as lazy pieces on the detector's own lane a hiccup every 64 windows.  So:
// segment 0 is there when set_certify returns, segment k + 1 is allocated and touched (on a stream of the thread's own)
// while the detector fills segment k; a consumer that outruns the allocator finds no room and promotes the epoch, like one
// that outgrows the whole ring.
struct AcRing {
    tsdrgpu_t *g = nullptr;
    size_t seg_bytes = 0;
    int seg_windows = 0, nseg_max = 0;
    std::vector<float *> seg;
    std::atomic<int> ready{0};
    int want = 0;
    bool quit = false, failed = false;
    std::mutex m;
    std::condition_variable cv;
    std::thread th;

    bool alloc_one(hipStream_t st)
    {
        float *p = nullptr;
        if (hipMalloc(&p, seg_bytes) != hipSuccess) { (void)hipGetLastError(); return false; }
        // the first use of fresh memory is what costs: paid here
        if (hipMemsetAsync(p, 0, seg_bytes, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { (void)hipFree(p); return false; }
        seg[ready.load(std::memory_order_relaxed)] = p;
        ready.fetch_add(1, std::memory_order_release);
        return true;
    }
    void run()
    {
        (void)hipSetDevice(g->device);
        hipStream_t st = nullptr;
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) st = nullptr;
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv.wait(lk, [&] { return quit || (!failed && want > ready.load() && ready.load() < nseg_max); });
            if (quit) break;
            lk.unlock();
            const bool ok = alloc_one(st);
            lk.lock();
            if (!ok) failed = true;  // no more room on the device: the ring stays as big as it got
        }
        lk.unlock();
        if (st) (void)hipStreamDestroy(st);
    }
    void ask(int nseg)
    {
        std::lock_guard<std::mutex> lk(m);
        if (nseg > want) { want = nseg; cv.notify_one(); }
    }
    void stop()
    {
        { std::lock_guard<std::mutex> lk(m); quit = true; cv.notify_one(); }
        if (th.joinable()) th.join();
        for (int i = 0; i < ready.load(); i++) (void)hipFree(seg[i]);
        seg.clear();
        ready.store(0);
    }
};

struct tsdrgpu_autocorr {
    tsdrgpu_t *g;
    uint32_t samplerate;
    int32_t frame_lo, frame_len, line_lo, line_len;
    uint32_t capture, n;
    uint64_t calls;
    double *d_plots;   // frame_len + line_len, + 1: the accumulated lag-0 value (scale of the certificate)
    double *d_snapshot;  // tsdrgpu_autocorr_plots_snapshot
    float2 *d_a, *d_b; // ping-pong work buffers, cap_windows * n/2 complex each (packed real transform)
    int cap_windows;
    float2 *d_last;    // packed correlation of the last window run (n/2 complex = n reals), or n complex (last_exact)
    int last_exact;
    float2 *d_expand;  // the same unpacked to n complex values, made on demand
    AcArgHost *d_arg;
    AcArgHost *h_arg;
    AcArgHost res;      // the last collected result
    hipEvent_t ev_arg;  // recorded behind the kernel that writes h_arg
    int arg_pending;
    double *d_pval, *d_psec;  // argmax partials: best and runner-up values
    int *d_pidx;
    hipStream_t st;  // g->stream, or g->bg (the background lane) when set asynchronous
    int plan5;       // tsdrgpu_autocorr_set_plan: 1 = the five-trip Stockham plan even where the three-trip one applies
    // exact mode (tsdrgpu_autocorr_set_exact, tsdrgpu_fftx.hip)
    int exact;
    double2 *d_tw;   // the reference's twiddle recurrence values, n-1 entries
    float2 *d_xz;    // AC_XBATCH windows of n complex points
    float *d_xmag;   // and as many of scratch (the forward transform's trips; the magnitudes in the six-trip form)
    // certified mode (tsdrgpu_autocorr_set_certify)
    int certify;       // 0 off, 1 windows retained by the library (ring), 2 retained by the caller
    int epoch_exact;   // this epoch (since the last reset) was promoted: its windows run in the exact form
    int promotions;    // epochs promoted so far (diagnostics)
    AcLogRec *log;
    int log_count, log_cap;
    AcRing *ring;      // certify == 1: segments of seg_windows windows of n magnitudes each
    int ring_cap, ring_count;  // capacity when every segment is there / position of the next window (windows)
    // runtime check of the certificate's premise (ac_premise_check)
    int check_every;       // every n-th plot update of a float32 epoch (0: never); the first update after set_certify always
    int since_check;       // plot updates since the last check (-1: none yet)
    unsigned long long *d_check;  // [0] max |fast - exact| as the bits of a non-negative double, [1] the checked window's lag-0 value
    long premise_checks, premise_failures;
    int premise_broken;    // a check of this epoch failed: no plot of it is certified any more (until it is promoted or reset)
    int recheck_next;      // ... and the next float32 epoch is checked at its first update
    // incremental promotion (tsdrgpu_autocorr_promote_step)
    int replay_rec, replay_win;  // next log record / window inside it to replay; replay_rec < 0: no replay in progress
};
// windows per launch of the exact form (TSDRGPU_XBATCH overrides: 1..16)
static int ac_xbatch()
{
    static const int v = [] {
        const char *e = getenv("TSDRGPU_XBATCH");
        const int n = e ? atoi(e) : 4;
        return n < 1 ? 1 : (n > 16 ? 16 : n);
    }();
    return v;
}
#define AC_XBATCH (ac_xbatch())

// (register DFTs, twiddle helpers and the three-trip autocorrelation kernels: fft4step.h)
// |z|^2 and |z| exactly as complex_to_abs_diff forms them (superbandwidth.c:67-81): one definition for the stand-alone kernel
// and for the transform's fused load, so that both give the same bits
__device__ __forceinline__ float sb_mag2(float2 c) { return c.x * c.x + c.y * c.y; }
__device__ __forceinline__ float sb_absdiff(const float2 *__restrict__ z, long long i)
{
    const float2 c = z[i];
    const float cur = sqrtf(sb_mag2(c));
    const float prev = (i == 0) ? sb_mag2(c) : sqrtf(sb_mag2(z[i - 1]));
    return cur - prev;
}

template <int IN_MODE>
__device__ __forceinline__ float2 fft_load(const void *__restrict__ xin, long long base, long long at, unsigned n, const int *__restrict__ aux)
{
    if (IN_MODE == 0) return ((const float2 *)xin + base)[at];
    if (IN_MODE == 1) return make_float2(((const float *)xin + base)[at], 0.f);
    if (IN_MODE == 2) {
        const float2 s = ((const float2 *)xin + base)[at];
        return make_float2(sqrtf(s.x * s.x + s.y * s.y), 0.f);
    }
    if (IN_MODE == 3) {
        const float *x = (const float *)xin + base;
        return make_float2(x[2 * at], x[2 * at + 1]);
    }
    if (IN_MODE == 5) return make_float2(sb_absdiff((const float2 *)xin + base, at), 0.f);  // complex_to_abs_diff of the input
    if (IN_MODE == 6) {  // the input rotated left by *aux floats (superbandwidth.c:135-137)
        const float *x = (const float *)((const float2 *)xin + base);
        const unsigned nfl = 2u * n, off = (unsigned)*aux;
        unsigned s0 = 2u * (unsigned)at + off;
        if (s0 >= nfl) s0 -= nfl;
        unsigned s1 = 2u * (unsigned)at + 1u + off;
        if (s1 >= nfl) s1 -= nfl;
        return make_float2(x[s0], x[s1]);
    }
    const float2 *x = (const float2 *)xin + base;
    const float2 a = x[2 * at], b = x[2 * at + 1];
    return make_float2(sqrtf(a.x * a.x + a.y * a.y), sqrtf(b.x * b.x + b.y * b.y));
}

// ---------------------------------------------------------------------------
// one Stockham pass of radix R over `batch` transforms of n points
//   thread j (< n/R): k = j mod Ns; reads x[j + t*n/R], multiplies by
//   w_{Ns*R}^{t*k}, R-point DFT, writes y[(j-k)*R + k + u*Ns]
// IN_MODE 0: complex input; 1: real float input (imag = 0); 2: interleaved IQ,
// magnitude taken on the fly (am_demod fused, TSDRLibrary.c:244-262);
// 3: real input packed two samples per complex point, z[m] = x[2m] + i x[2m+1];
// 4: the same packing with the samples demodulated from interleaved IQ
// OUT_MAG: store (|v|*scale, 0) — fft_complex_to_absolute_complex, fft.c:34-45
// ---------------------------------------------------------------------------
template <int R, int IN_MODE, bool OUT_MAG>
__global__ __launch_bounds__(256) void k_fft_pass(const void *__restrict__ xin, long long in_stride, float2 *__restrict__ y,
                                                  unsigned n, unsigned Ns, int conj_in, int conj_out, float scale, const int *__restrict__ aux)
{
    const unsigned T = n / R;
    const unsigned j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= T) return;
    const unsigned b = blockIdx.y;
    float2 v[R];
#pragma unroll
    for (int t = 0; t < R; t++) v[t] = fft_load<IN_MODE>(xin, (long long)b * in_stride, (long long)j + (long long)t * T, n, aux);
    if (conj_in) {
#pragma unroll
        for (int t = 0; t < R; t++) v[t].y = -v[t].y;
    }
    const unsigned k = j & (Ns - 1);
    if (Ns > 1) {
        const unsigned span = Ns * R;  // power of two
        const float inv = -2.0f / (float)span;
#pragma unroll
        for (int t = 1; t < R; t++) {
            const unsigned m = (t * k) & (span - 1);
            float s, c;
            sincospif((float)m * inv, &s, &c);
            v[t] = cmul(v[t], make_float2(c, s));
        }
    }
    dft_reg<R>(v);
    float2 *yo = y + (long long)b * n + ((j - k) * R + k);
#pragma unroll
    for (int u = 0; u < R; u++) {
        float2 o = v[u];
        if (OUT_MAG) {
            o.x *= scale;
            o.y *= scale;
            o = make_float2(sqrtf(o.x * o.x + o.y * o.y), 0.f);
        } else {
            if (conj_out) o.y = -o.y;
            o.x *= scale;
            o.y *= scale;
        }
        yo[(long long)u * Ns] = o;
    }
}

// ---------------------------------------------------------------------------
// LDS pass: one Stockham pass of radix R = 16*R1 (16..256) where the R-point
// DFT is done on chip as DFT-R1 (registers) -> LDS exchange -> DFT-16
// (registers).  A workgroup of 256 threads owns a tile of C = 256/R1
// neighbouring columns j (4096 points, 32 KiB of LDS), so every global access is
// a run of C*8 >= 128 contiguous bytes.  With it a 2^22-point transform takes 3
// trips over memory instead of 6.
//   n = t0 + 16*i (t0 < 16, i < R1), k = k1 + R1*k2 (k1 < R1, k2 < 16):
//   X[k] = sum_t0 w16^{t0 k2} [ w_R^{t0 k1} sum_i x[t0+16 i] w_R1^{i k1} ]
// ---------------------------------------------------------------------------
// Outer twiddles of an LDS pass, w_span^(nidx*k) for the 16 elements nidx = q*G + a + 16*i of a thread.
// The exponent factors as (q*G)*k + a*k + (16*i)*k, so 1 + log2(G) + log2(R1) = 5 accurately evaluated
// sincospif's (powers of two of each factor) and a few complex products replace 16 evaluations; a
// product chain is at most 4 deep, i.e. a few f32 ulps.  (The passes were VALU-bound on sincospif.)
template <int R1>
__device__ __forceinline__ void outer_twiddles(float2 (&v)[16], unsigned q, unsigned k, unsigned span)
{
    constexpr int G = 16 / R1;
    const unsigned mask = span - 1;
    const float inv = -2.0f / (float)span;
    float2 pw[R1], pa[G];
    tw_powers<R1>(pw, 16u * k, mask, inv);
    tw_powers<G>(pa, k, mask, inv);
    const float2 bq = tw_exact(q * G * k, mask, inv);
#pragma unroll
    for (int a = 0; a < G; a++) {
        const float2 ba = a ? cmul(bq, pa[a]) : bq;
#pragma unroll
        for (int i = 0; i < R1; i++) v[a * R1 + i] = cmul(v[a * R1 + i], i ? cmul(ba, pw[i]) : ba);
    }
}

static const FftKeep KEEP_ALL = {0, -1, 0u, 0u, 0u, 0u};

template <int R1, int IN_MODE, bool OUT_MAG>
__global__ __launch_bounds__(256) void k_fft_lds(const void *__restrict__ xin, long long in_stride, float2 *__restrict__ y,
                                                 unsigned n, unsigned Ns, int conj_in, int conj_out, float scale, FftKeep keep,
                                                 const int *__restrict__ aux)
{
    constexpr int R = 16 * R1;
    constexpr int C = 256 / R1;
    constexpr int G = 16 / R1;  // DFT-R1's per thread in stage 1
    __shared__ float2 lds[4096 + 256];
    __shared__ float2 tw[256];
    const unsigned T = n / R;
    const unsigned tid = threadIdx.x;
    const unsigned c = tid % C, q = tid / C;
    const unsigned j = blockIdx.x * C + c;
    const unsigned b = blockIdx.y;
    {
        float sn, cs;
        sincospif(-(float)tid * (1.0f / 128.0f), &sn, &cs);
        tw[tid] = make_float2(cs, sn);
    }
    float2 v[16];
#pragma unroll
    for (int a = 0; a < G; a++) {
#pragma unroll
        for (int i = 0; i < R1; i++) {
            const unsigned nidx = q * G + a + 16 * i;
            const long long at = (long long)j + (long long)nidx * T;
            float2 val = fft_load<IN_MODE>(xin, (long long)b * in_stride, at, n, aux);
            if (conj_in) val.y = -val.y;
            v[a * R1 + i] = val;
        }
    }
    const unsigned k = j & (Ns - 1);
    if (Ns > 1) outer_twiddles<R1>(v, q, k, Ns * R);
    __syncthreads();  // tw[] ready
#pragma unroll
    for (int a = 0; a < G; a++) {
        dft_reg<R1>(*reinterpret_cast<float2(*)[R1]>(&v[a * R1]));
        const unsigned t0 = q * G + a;
#pragma unroll
        for (int k1 = 0; k1 < R1; k1++) {
            float2 val = v[a * R1 + k1];
            if (R1 > 1 && k1 > 0) val = cmul(val, tw[(t0 * k1 * (256 / R)) & 255]);
            lds[(k1 * 16 + t0) * C + c] = val;
        }
    }
    __syncthreads();
    // stage 2: this thread is (k1 = q, column c)
    float2 w[16];
#pragma unroll
    for (int t0 = 0; t0 < 16; t0++) w[t0] = lds[(q * 16 + t0) * C + c];
    dft_reg<16>(w);
#pragma unroll
    for (int k2 = 0; k2 < 16; k2++) {
        float2 o = w[k2];
        if (OUT_MAG) {
            o.x *= scale;
            o.y *= scale;
            o = make_float2(sqrtf(o.x * o.x + o.y * o.y), 0.f);
        } else {
            if (conj_out) o.y = -o.y;
            o.x *= scale;
            o.y *= scale;
        }
        w[k2] = o;
    }
    float2 *yb = y + (long long)b * n;
    if (Ns == 1) {
        // outputs of the tile are the contiguous block [jb*R, jb*R + 4096): stage through LDS
        __syncthreads();
#pragma unroll
        for (int k2 = 0; k2 < 16; k2++) lds[c * (R + 1) + q + R1 * k2] = w[k2];
        __syncthreads();
        float2 *dst = yb + (long long)blockIdx.x * C * R;
#pragma unroll
        for (int it = 0; it < 16; it++) {
            const unsigned idx = tid + 256 * it;
            dst[idx] = lds[(idx / R) * (R + 1) + (idx % R)];
        }
    } else {
        const unsigned o0 = (j - k) * R + k;
        float2 *yo = yb + o0;
        if (keep.on && (int)b != keep.full_b) {
#pragma unroll
            for (int k2 = 0; k2 < 16; k2++) {
                const unsigned o = o0 + (q + R1 * k2) * Ns;
                if ((o >= keep.lo0 && o < keep.hi0) || (o >= keep.lo1 && o < keep.hi1) || o == 0u) yo[(long long)(q + R1 * k2) * Ns] = w[k2];
            }
        } else {
#pragma unroll
            for (int k2 = 0; k2 < 16; k2++) yo[(long long)(q + R1 * k2) * Ns] = w[k2];
        }
    }
}

struct PassPlan {
    int radix[32];
    int count;
};

static PassPlan plan_passes(uint32_t n)
{
    PassPlan p;
    p.count = 0;
    int m = 0;
    while ((1u << m) < n) m++;
    if (m >= 12) {
        // LDS passes: split log2(n) into ceil(m/8) radices of 2^4..2^8, smallest first (the
        // first pass reads the narrowest elements, so it gets the widest tile)
        const int parts = (m + 7) / 8;
        const int base = m / parts, rem = m % parts;
        for (int i = 0; i < parts; i++) p.radix[p.count++] = 1 << (base + (i >= parts - rem ? 1 : 0));
        return p;
    }
    // small transforms: register-only radix 16/8/4/2 passes
    int rem = m % 4;
    if (rem) p.radix[p.count++] = 1 << rem;
    for (int i = 0; i < m / 4; i++) p.radix[p.count++] = 16;
    if (p.count == 0) p.radix[p.count++] = 1;  // n == 1
    return p;
}

template <int IN_MODE, bool OUT_MAG>
static void launch_pass(tsdrgpu_t *g, hipStream_t st, int R, const void *x, long long in_stride, float2 *y, unsigned n, unsigned Ns, int batch,
                        int conj_in, int conj_out, float scale, const FftKeep &keep = KEEP_ALL, const int *aux = nullptr)
{
    if (n >= 4096) {
        const int R1 = R / 16;
        dim3 grid(n / 4096, batch);  // (n/R)/C tiles, R*C = 4096
        switch (R1) {
            case 1: TSDR_LAUNCH(g, PROF_FFT_PASS, st, (k_fft_lds<1, IN_MODE, OUT_MAG>), grid, 256, x, in_stride, y, n, Ns, conj_in, conj_out, scale, keep, aux); break;
            case 2: TSDR_LAUNCH(g, PROF_FFT_PASS, st, (k_fft_lds<2, IN_MODE, OUT_MAG>), grid, 256, x, in_stride, y, n, Ns, conj_in, conj_out, scale, keep, aux); break;
            case 4: TSDR_LAUNCH(g, PROF_FFT_PASS, st, (k_fft_lds<4, IN_MODE, OUT_MAG>), grid, 256, x, in_stride, y, n, Ns, conj_in, conj_out, scale, keep, aux); break;
            case 8: TSDR_LAUNCH(g, PROF_FFT_PASS, st, (k_fft_lds<8, IN_MODE, OUT_MAG>), grid, 256, x, in_stride, y, n, Ns, conj_in, conj_out, scale, keep, aux); break;
            default: TSDR_LAUNCH(g, PROF_FFT_PASS, st, (k_fft_lds<16, IN_MODE, OUT_MAG>), grid, 256, x, in_stride, y, n, Ns, conj_in, conj_out, scale, keep, aux); break;
        }
        return;
    }
    const unsigned T = n / R;
    dim3 grid((T + 255) / 256, batch);
    switch (R) {
        case 2: TSDR_LAUNCH(g, PROF_FFT_PASS, st, (k_fft_pass<2, IN_MODE, OUT_MAG>), grid, 256, x, in_stride, y, n, Ns, conj_in, conj_out, scale, aux); break;
        case 4: TSDR_LAUNCH(g, PROF_FFT_PASS, st, (k_fft_pass<4, IN_MODE, OUT_MAG>), grid, 256, x, in_stride, y, n, Ns, conj_in, conj_out, scale, aux); break;
        case 8: TSDR_LAUNCH(g, PROF_FFT_PASS, st, (k_fft_pass<8, IN_MODE, OUT_MAG>), grid, 256, x, in_stride, y, n, Ns, conj_in, conj_out, scale, aux); break;
        default: TSDR_LAUNCH(g, PROF_FFT_PASS, st, (k_fft_pass<16, IN_MODE, OUT_MAG>), grid, 256, x, in_stride, y, n, Ns, conj_in, conj_out, scale, aux); break;
    }
}

// Runs all passes of `batch` n-point transforms.  Input: `in` (mode per
// in_mode, window b at in + b*in_stride elements).  Work buffers a, b (batch*n
// float2 each).  Returns the buffer that holds the result.
// Runs passes [pbegin, pend) of the plan `radix[0..count)`; Ns0 = product of the radices before
// pbegin.  conj_first / conj_last: conjugate the input of pass 0 / the output of pass count-1 (the
// inverse transform as conj(FFT(conj(x)))); scale / mag_out apply to pass count-1.
static float2 *run_fft_range(tsdrgpu_t *g, const void *in, int in_mode, long long in_stride, float2 *a, float2 *b, uint32_t n,
                             int batch, const int *radix, int count, int pbegin, int pend, unsigned Ns0, int conj_first,
                             int conj_last, bool mag_out, float scale, hipStream_t st, const FftKeep &keep = KEEP_ALL,
                             const int *aux = nullptr)
{
    unsigned Ns = Ns0;
    const void *src = in;
    long long sstride = in_stride;
    int smode = in_mode;
    float2 *dst = (in == (const void *)a) ? b : a;
    for (int i = pbegin; i < pend; i++) {
        const int R = radix[i];
        const bool first = i == 0, last = i == count - 1;
        const int cin = (conj_first && first) ? 1 : 0, cout = (conj_last && last) ? 1 : 0;
        const float sc = last ? scale : 1.0f;
        if (R == 1) {  // n == 1: copy
            (void)hipMemcpyAsync(dst, src, sizeof(float2) * batch, hipMemcpyDeviceToDevice, st);
        } else if (last && mag_out) {
            if (smode == 0) launch_pass<0, true>(g, st, R, src, sstride, dst, n, Ns, batch, cin, cout, sc);
            else if (smode == 1) launch_pass<1, true>(g, st, R, src, sstride, dst, n, Ns, batch, cin, cout, sc);
            else launch_pass<2, true>(g, st, R, src, sstride, dst, n, Ns, batch, cin, cout, sc);
        } else {
            switch (smode) {
                case 0: launch_pass<0, false>(g, st, R, src, sstride, dst, n, Ns, batch, cin, cout, sc, last ? keep : KEEP_ALL); break;
                case 1: launch_pass<1, false>(g, st, R, src, sstride, dst, n, Ns, batch, cin, cout, sc); break;
                case 2: launch_pass<2, false>(g, st, R, src, sstride, dst, n, Ns, batch, cin, cout, sc); break;
                case 3: launch_pass<3, false>(g, st, R, src, sstride, dst, n, Ns, batch, cin, cout, sc); break;
                case 5: launch_pass<5, false>(g, st, R, src, sstride, dst, n, Ns, batch, cin, cout, sc, last ? keep : KEEP_ALL); break;
                case 6: launch_pass<6, false>(g, st, R, src, sstride, dst, n, Ns, batch, cin, cout, sc, last ? keep : KEEP_ALL, aux); break;
                default: launch_pass<4, false>(g, st, R, src, sstride, dst, n, Ns, batch, cin, cout, sc); break;
            }
        }
        Ns *= R;
        src = dst;
        sstride = n;
        smode = 0;
        dst = (dst == a) ? b : a;
    }
    return (float2 *)src;
}

static float2 *run_fft(tsdrgpu_t *g, const void *in, int in_mode, long long in_stride, float2 *a, float2 *b, uint32_t n, int batch,
                       int inverse, bool mag_out, float scale, hipStream_t st = nullptr)
{
    if (!st) st = g->stream;
    const PassPlan p = plan_passes(n);
    return run_fft_range(g, in, in_mode, in_stride, a, b, n, batch, p.radix, p.count, 0, p.count, 1, inverse, inverse, mag_out,
                         scale, st);
}

// ---------------------------------------------------------------------------
// tsdrgpu_fft: fft_perform (fft.c:96-176) on a caller buffer
// ---------------------------------------------------------------------------
extern "C" int tsdrgpu_fft(tsdrgpu_t *g, float *d_iq, uint32_t n, int inverse)
{
    if (!g || !d_iq || n == 0 || (n & (n - 1))) return g ? tsdr_fail(g, TSDRGPU_EINVAL, "tsdrgpu_fft", "n must be a power of two") : TSDRGPU_EINVAL;
    if (n == 1) return TSDRGPU_OK;
    const size_t need = sizeof(float2) * (size_t)n * 2;
    if (g->fft_ws_bytes < need) {  // the context keeps the ping-pong scratch between calls
        HIP_TRY(g, hipStreamSynchronize(g->stream));
        if (g->fft_ws) (void)hipFree(g->fft_ws);
        g->fft_ws = nullptr;
        g->fft_ws_bytes = 0;
        if (hipMalloc(&g->fft_ws, need) != hipSuccess) return tsdr_fail(g, TSDRGPU_ENOMEM, "tsdrgpu_fft", "work buffer");
        g->fft_ws_bytes = need;
    }
    float2 *tmp = (float2 *)g->fft_ws;
    // pass 1 reads the caller's buffer, the rest ping-pongs inside the scratch
    float2 *res = run_fft(g, d_iq, 0, n, tmp, tmp + n, n, 1, inverse, false, inverse ? 1.0f : 1.0f / (float)n);
    KERNEL_CHECK(g, "fft passes");
    HIP_TRY(g, hipMemcpyAsync(d_iq, res, sizeof(float2) * (size_t)n, hipMemcpyDeviceToDevice, g->stream));
    return TSDRGPU_OK;
}

// fft_perform in the reference's own arithmetic (tsdrgpu_fftx.hip): bit-identical, slower
extern "C" int tsdrgpu_fft_exact(tsdrgpu_t *g, float *d_iq, uint32_t n, int inverse)
{
    if (!g || !d_iq || n == 0 || (n & (n - 1))) return g ? tsdr_fail(g, TSDRGPU_EINVAL, "tsdrgpu_fft_exact", "n must be a power of two") : TSDRGPU_EINVAL;
    if (n == 1) return TSDRGPU_OK;
    HIP_TRY(g, hipStreamSynchronize(g->stream));
    if (g->fftx_n != n) {
        if (g->fftx_tw) (void)hipFree(g->fftx_tw);
        g->fftx_tw = nullptr;
        g->fftx_n = 0;
        double2 *tw = nullptr;
        const int rc = fftx_build_table(g, n, &tw);
        if (rc) return rc;
        g->fftx_tw = tw;
        g->fftx_n = n;
    }
    const size_t need = sizeof(float2) * (size_t)n;
    if (g->fft_ws_bytes < need) {
        if (g->fft_ws) (void)hipFree(g->fft_ws);
        g->fft_ws = nullptr;
        g->fft_ws_bytes = 0;
        if (hipMalloc(&g->fft_ws, need) != hipSuccess) return tsdr_fail(g, TSDRGPU_ENOMEM, "tsdrgpu_fft_exact", "work buffer");
        g->fft_ws_bytes = need;
    }
    const int rc = fftx_perform(g, g->stream, (const float2 *)d_iq, (float2 *)g->fft_ws, n, (const double2 *)g->fftx_tw, inverse);
    if (rc) return rc;
    HIP_TRY(g, hipMemcpyAsync(d_iq, g->fft_ws, need, hipMemcpyDeviceToDevice, g->stream));
    return TSDRGPU_OK;
}

// ---------------------------------------------------------------------------
// Real-input trick for the autocorrelation.  The capture window is real, so it is
// transformed as nh = n/2 complex points z[m] = x[2m] + i x[2m+1]; with
// Z = FFT_nh(z) (unscaled) and w = exp(-2 pi i / n):
//     X[k]   = (A + B)/2 - (i/2) w^k (A - B),   A = Z[k], B = conj(Z[nh-k])
//     M[k]   = |X[k]| / n          (fft.c:167-175 scale, fft.c:34-45 magnitude; M is real and even)
// and the unscaled inverse r = IFFT_n(M) comes out of one nh-point inverse FFT of
//     Zin[k] = (M[k] + M[nh-k]) + i w^-k (M[k] - M[nh-k])
// packed as zout[m] = r[2m] + i r[2m+1].  k_ac_split does the Z -> Zin step in
// place, one thread per (k, nh-k) pair.  Memory traffic per window drops from
// 96 n to 56 n bytes.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ac_split(float2 *__restrict__ z, unsigned nh)
{
    const unsigned k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > nh / 2) return;
    float2 *zb = z + (long long)blockIdx.y * nh;
    const float inv_n = 1.0f / (float)(2 * nh);
    if (k == 0) {
        const float2 z0 = zb[0];
        const float m0 = fabsf(z0.x + z0.y) * inv_n;   // X[0]  = Re Z0 + Im Z0
        const float mh = fabsf(z0.x - z0.y) * inv_n;   // X[nh] = Re Z0 - Im Z0
        zb[0] = make_float2(m0 + mh, m0 - mh);
        return;
    }
    const unsigned km = nh - k;
    const float2 a = zb[k];
    const float2 bm = zb[km];
    const float2 b = make_float2(bm.x, -bm.y);
    float sn, cs;
    sincospif(-(float)k * (1.0f / (float)nh), &sn, &cs);  // w^k, w = exp(-2 pi i / n), n = 2 nh (nh = 2^m: exact)
    const float2 wk = make_float2(cs, sn);
    // X[k] = (A+B)/2 - (i/2) w^k (A-B)
    const float2 sum = make_float2(0.5f * (a.x + b.x), 0.5f * (a.y + b.y));
    const float2 dif = make_float2(0.5f * (a.x - b.x), 0.5f * (a.y - b.y));
    const float2 t = cmul(wk, dif);                  // w^k (A-B)/2
    const float2 xk = make_float2(sum.x + t.y, sum.y - t.x);  // sum - i t
    // X[nh-k] = conj( (A+B)/2 + (i/2) w^k (A-B) )   (Hermitian partner, w^{nh-k} = -conj(w^k))
    const float2 xm = make_float2(sum.x - t.y, -(sum.y + t.x));
    const float mk = sqrtf(xk.x * xk.x + xk.y * xk.y) * inv_n;
    const float mm = sqrtf(xm.x * xm.x + xm.y * xm.y) * inv_n;
    // Zin[k] = (M[k]+M[nh-k]) + i w^-k (M[k]-M[nh-k]),  w^-k = conj(w^k)
    const float s = mk + mm, d = mk - mm;
    // i * conj(wk) * d = i (cs - i sn) d = (sn d) + i (cs d)
    zb[k] = make_float2(s + sn * d, cs * d);
    if (km != k) {
        // Zin[nh-k] = (M[nh-k]+M[k]) + i w^-(nh-k) (M[nh-k]-M[k]); w^-(nh-k) = -w^k
        // i * (-wk) * (-d) = i wk d = i (cs + i sn) d = (-sn d) + i (cs d)
        zb[km] = make_float2(s - sn * d, cs * d);
    }
}

// ---------------------------------------------------------------------------
// k_ac_mid: the middle of the autocorrelation in one trip over memory.
// The last forward pass (radix R, Ns = nh/R) produces, for a tile of C columns k, the spectrum
// entries Z[k + u*Ns]; the split needs Z[idx] together with Z[nh-idx], which lies in the mirrored
// column Ns-k at row R-1-u; and the first inverse pass (same radix, Ns = 1) consumes exactly the
// entries of one tile again.  So a workgroup takes a tile A = columns [1+C*b, 1+C*b+C) and its
// mirror B = columns [Ns-C-C*b, Ns-C*b) (C = 128/R1 columns each), runs the forward pass for both into LDS, does the split
// in LDS, runs the first inverse pass from LDS and stores the two contiguous output blocks.
// Column 0 mirrors onto itself and is handled by workgroup 0 (ac_mid_col0).  Saves writing the spectrum,
// the k_ac_split round trip and re-reading it: 64 of 224 MB per 2^22-sample window.
// ---------------------------------------------------------------------------
template <int R>
__device__ __forceinline__ void ac_mid_col0(const float2 *__restrict__ xb, float2 *__restrict__ yb, unsigned nh, float2 *col,
                                            float2 *tw)
{
    // called by all 256 threads of one workgroup; threads u < R each own one row of column 0
    const unsigned Ns = nh / R;
    const unsigned u = threadIdx.x;
    const bool act = u < R;
    if (act) {
        float sn, cs;
        sincospif(-2.0f * (float)u / (float)R, &sn, &cs);
        tw[u] = make_float2(cs, sn);
        col[u] = xb[(long long)u * Ns];  // k = 0: no outer twiddle
    }
    __syncthreads();
    float2 acc = make_float2(0.f, 0.f);
    if (act)
        for (unsigned t = 0; t < R; t++) {  // direct DFT: X[u*Ns] = sum_t x[t*Ns] w_R^(t u)
            const float2 p = cmul(col[t], tw[(t * u) & (R - 1)]);
            acc.x += p.x;
            acc.y += p.y;
        }
    __syncthreads();
    if (act) col[u] = acc;
    __syncthreads();
    float2 zin = make_float2(0.f, 0.f);
    if (act) {
        if (u == 0) {
            const float inv_n = 1.0f / (float)(2 * nh);
            const float2 z0 = col[0];
            const float m0 = fabsf(z0.x + z0.y) * inv_n, mh = fabsf(z0.x - z0.y) * inv_n;
            zin = make_float2(m0 + mh, m0 - mh);
        } else {
            float2 zk, zkm;
            float sn, cs;
            sincospif(-(float)u * (1.0f / (float)R), &sn, &cs);  // k = u*Ns: exp(-i pi k/nh) = exp(-i pi u/R)
            ac_split_pair(col[u], col[R - u], make_float2(cs, sn), nh, &zk, &zkm);
            zin = zk;
        }
    }
    __syncthreads();
    if (act) col[u] = make_float2(zin.x, -zin.y);  // conjugated input of the inverse transform
    __syncthreads();
    if (act) {
        acc = make_float2(0.f, 0.f);
        for (unsigned t = 0; t < R; t++) {
            const float2 p = cmul(col[t], tw[(t * u) & (R - 1)]);
            acc.x += p.x;
            acc.y += p.y;
        }
        yb[u] = acc;  // y[0*R + u]
    }
}

template <int R1>
__global__ __launch_bounds__(256, 4) void k_ac_mid(const float2 *__restrict__ x, float2 *__restrict__ y, unsigned nh)
{
    // 128 threads per tile (tile A: threads 0..127, its mirror B: 128..255); a tile is C2 = 128/R1
    // columns x R rows = 2048 points, so the LDS footprint and the per-thread work equal k_fft_lds's
    constexpr int R = 16 * R1;
    constexpr int C2 = 128 / R1;
    constexpr int G = 16 / R1;
    constexpr int TILE = 2048 + 128;  // + room for the padded [c][R+1] staging of the stores
    __shared__ float2 spec[2][TILE];
    __shared__ float2 tw[256];
    const unsigned Ns = nh / R;  // columns; also the T of both passes
    const unsigned tid = threadIdx.x;
    const unsigned tile = tid >> 7, t = tid & 127;
    const unsigned c = t % C2, q = t / C2;
    const float2 *xb = x + (long long)blockIdx.y * nh;
    float2 *yb = y + (long long)blockIdx.y * nh;
    if (blockIdx.x == 0) {  // workgroup 0: column 0, which mirrors onto itself (a direct R-point DFT each way)
        ac_mid_col0<R>(xb, yb, nh, spec[0], spec[1]);
        return;
    }
    // Tile A starts at column 1, so its 128-byte row segments straddle two cache lines, the second of
    // which is the first of the next tile: workgroups are dispatched round-robin over the 8 XCDs, so
    // consecutive tiles are given to the SAME XCD (ids x, x+8, x+16 ... are neighbours in time there)
    // and the shared line is an L2 hit instead of a second HBM fetch (PMC: 142 -> ~100 MB per launch).
    // (ids with equal l mod 8 still share an XCD although workgroup 0 shifts everything by one)
    const unsigned gx = gridDim.x - 1, l = blockIdx.x - 1;
    const unsigned bx = (gx % 8u == 0u) ? (l % 8u) * (gx / 8u) + l / 8u : l;
    const unsigned colA = 1u + C2 * bx, colB = Ns - C2 - C2 * bx;
    const unsigned col0 = tile ? colB : colA;
    float2 *L = spec[tile];
    {
        float sn, cs;
        sincospif(-(float)tid * (1.0f / 128.0f), &sn, &cs);
        tw[tid] = make_float2(cs, sn);
    }
    // ---- last forward pass -> L[u*C2 + c]
    const unsigned j = col0 + c;
    float2 v[16];
#pragma unroll
    for (int a = 0; a < G; a++)
#pragma unroll
        for (int i = 0; i < R1; i++) {
            const unsigned nidx = q * G + a + 16 * i;
            v[a * R1 + i] = xb[(long long)j + (long long)nidx * Ns];
        }
    outer_twiddles<R1>(v, q, j, nh);  // w_nh^(nidx*k), k = j
    __syncthreads();  // tw[] ready
#pragma unroll
    for (int a = 0; a < G; a++) {
        dft_reg<R1>(*reinterpret_cast<float2(*)[R1]>(&v[a * R1]));
        const unsigned t0 = q * G + a;
#pragma unroll
        for (int k1 = 0; k1 < R1; k1++) {
            float2 val = v[a * R1 + k1];
            if (R1 > 1 && k1 > 0) val = cmul(val, tw[(t0 * k1 * (256 / R)) & 255]);
            L[(k1 * 16 + t0) * C2 + c] = val;
        }
    }
    __syncthreads();
    float2 w[16];
#pragma unroll
    for (int t0 = 0; t0 < 16; t0++) w[t0] = L[(q * 16 + t0) * C2 + c];
    dft_reg<16>(w);
    __syncthreads();
#pragma unroll
    for (int k2 = 0; k2 < 16; k2++) L[(q + R1 * k2) * C2 + c] = w[k2];  // row u = q + R1*k2
    __syncthreads();
    // ---- split: element (u, c) of A pairs with (R-1-u, C2-1-c) of B; 2048 pairs, 8 per thread
    // pair `it` of a thread sits 256/C2 rows = (256/C2)*Ns spectrum entries after pair it-1, so its
    // twiddle exp(-i pi k/nh) is the first one times exp(-i pi it/8) = tw[16 it] (C2*R = 2048)
    {
        const unsigned u0 = tid / C2, cc = tid % C2;
        float sn0, cs0;
        sincospif(-(float)((colA + cc) + u0 * Ns) * (1.0f / (float)nh), &sn0, &cs0);
        const float2 w0 = make_float2(cs0, sn0);
#pragma unroll
        for (int it = 0; it < 8; it++) {
            const unsigned u = u0 + (256 / C2) * it;
            const unsigned pa = u * C2 + cc, pb = (R - 1 - u) * C2 + (C2 - 1 - cc);
            const float2 wk = it ? cmul(w0, tw[16 * it]) : w0;
            float2 zk, zkm;
            ac_split_pair(spec[0][pa], spec[1][pb], wk, nh, &zk, &zkm);
            spec[0][pa] = zk;
            spec[1][pb] = zkm;
        }
    }
    __syncthreads();
    // ---- first inverse pass (Ns = 1: no outer twiddles; input conjugated)
#pragma unroll
    for (int a = 0; a < G; a++)
#pragma unroll
        for (int i = 0; i < R1; i++) {
            const unsigned nidx = q * G + a + 16 * i;
            float2 val = L[nidx * C2 + c];
            val.y = -val.y;
            v[a * R1 + i] = val;
        }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < G; a++) {
        dft_reg<R1>(*reinterpret_cast<float2(*)[R1]>(&v[a * R1]));
        const unsigned t0 = q * G + a;
#pragma unroll
        for (int k1 = 0; k1 < R1; k1++) {
            float2 val = v[a * R1 + k1];
            if (R1 > 1 && k1 > 0) val = cmul(val, tw[(t0 * k1 * (256 / R)) & 255]);
            L[(k1 * 16 + t0) * C2 + c] = val;
        }
    }
    __syncthreads();
#pragma unroll
    for (int t0 = 0; t0 < 16; t0++) w[t0] = L[(q * 16 + t0) * C2 + c];
    dft_reg<16>(w);
    __syncthreads();
    // outputs of a tile: y[(col0+c)*R + u], a contiguous block of 2048: stage as [c][R+1] and copy out
#pragma unroll
    for (int k2 = 0; k2 < 16; k2++) L[c * (R + 1) + q + R1 * k2] = w[k2];
    __syncthreads();
#pragma unroll
    for (int tl = 0; tl < 2; tl++) {
        float2 *dst = yb + (long long)(tl ? colB : colA) * R;
        // the self-mirrored column Ns/2 is the last of tile A and the first of tile B in the last
        // workgroup; both copies are complete, A's is the one stored
        const bool dup = tl == 1 && colB == Ns / 2;
#pragma unroll
        for (int it = 0; it < 8; it++) {
            const unsigned idx = tid + 256 * it;
            if (dup && idx < R) continue;
            dst[idx] = spec[tl][(idx / R) * (R + 1) + (idx % R)];
        }
    }
}

// column 0 of the same step: Z[u*Ns] <-> Z[(R-u)*Ns]; one workgroup of R threads per window

// unpack zout (r[2m] + i r[2m+1]) into the reference's layout: complex, imaginary part 0
__global__ __launch_bounds__(256) void k_ac_expand(const float2 *__restrict__ zout, float2 *__restrict__ corr, unsigned nh)
{
    const unsigned m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= nh) return;
    const float2 v = zout[m];
    corr[2 * m] = make_float2(v.x, 0.f);
    corr[2 * m + 1] = make_float2(v.y, 0.f);
}

// ---------------------------------------------------------------------------
// accummulate (frameratedetector.c:34-62) over a batch of windows, in window
// order, so the running mean's f64 rounding matches the reference's recurrence.
// ---------------------------------------------------------------------------
// (Round 5 tried requesting all of a launch's windows' values before the sequential recurrence — on the theory that nine
// load-wait-divide round trips are what its 12 us per launch of nine windows are: 13.1 us, no gain; the kernel is at what 36 MB in
// 2 600 short workgroups behind a launch ramp cost.)
__global__ __launch_bounds__(256) void k_accumulate(const float *__restrict__ corr, unsigned n, int nwindows, int frame_lo,
                                                    int frame_len, int line_lo, int line_len, double *__restrict__ plots,
                                                    unsigned long long calls_before, int mode)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > frame_len + line_len) return;
    // entry frame_len + line_len accumulates lag 0 the same way: the scale of the argmax certificate
    const int lag = (i < frame_len) ? (frame_lo + i) : (i < frame_len + line_len ? line_lo + (i - frame_len) : 0);
    double acc = plots[i];
    for (int w = 0; w < nwindows; w++) {
        // r[lag] of window w (real: the packed inverse transform has no imaginary residue)
        // sqrt(re*re + im*im) with im == 0 (frameratedetector.c:41-44): the square of a float is exact in f64 and the
        // correctly rounded root of an exact square is exact, so this is |re| bit for bit
        const double now = fabs((double)corr[(long long)w * n + lag]);
        if (mode == 0) {
            const unsigned long long calls = calls_before + w + 1;
            acc = (acc * (double)(calls - 1) + now) / (double)calls;
        } else {
            acc += now;
        }
    }
    plots[i] = acc;
}

__global__ void k_scale_plots(double *plots, int count, double divisor)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) plots[i] = plots[i] / divisor;
}

// argmax with lowest-index tie-break (PlotVisualizer.java:233-236), two stages:
// ARGMAX_BLOCKS workgroups per plot (enough to pull 5 MB of lags at memory speed), then one wave per plot.
// Beside the maximum the reduction carries the RUNNER-UP value (the largest value at any other lag; equal to the
// maximum when the maximum is attained twice): the certificate of tsdrgpu_autocorr_certificate compares their
// distance with the bound on what separates this plot from the reference's.
#define ARGMAX_BLOCKS 512
struct ArgTop {
    double best, second;
    int at;
};
__device__ __forceinline__ void argtop_merge(ArgTop &a, double ob, double os, int oi)
{
    const double lo = a.best < ob ? a.best : ob;  // the loser of the two maxima is a runner-up candidate
    double sec = a.second > os ? a.second : os;
    sec = sec > lo ? sec : lo;
    if (ob > a.best || (ob == a.best && oi < a.at)) { a.best = ob; a.at = oi; }
    a.second = sec;
}
__device__ __forceinline__ void argtop_wave(ArgTop &a)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double ob = __shfl_down(a.best, o, 64);
        const double os = __shfl_down(a.second, o, 64);
        const int oi = __shfl_down(a.at, o, 64);
        argtop_merge(a, ob, os, oi);
    }
}

__global__ __launch_bounds__(256) void k_argmax_partial(const double *__restrict__ plots, int frame_len, int line_len,
                                                        double *__restrict__ pval, double *__restrict__ psec, int *__restrict__ pidx)
{
    const int plot = blockIdx.y;
    const double *p = plot == 0 ? plots : plots + frame_len;
    const int len = plot == 0 ? frame_len : line_len;
    ArgTop a = {-1.0, -1.0, 0x7fffffff};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < len; i += gridDim.x * blockDim.x) {
        const double v = p[i];
        if (v > a.best) { a.second = a.best; a.best = v; a.at = i; }  // i ascending per thread
        else if (v > a.second) a.second = v;
    }
    __shared__ double sb[4], ss[4];
    __shared__ int si[4];
    argtop_wave(a);
    if ((threadIdx.x & 63) == 0) { sb[threadIdx.x >> 6] = a.best; ss[threadIdx.x >> 6] = a.second; si[threadIdx.x >> 6] = a.at; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) argtop_merge(a, sb[w], ss[w], si[w]);
        pval[plot * ARGMAX_BLOCKS + blockIdx.x] = a.best;
        psec[plot * ARGMAX_BLOCKS + blockIdx.x] = a.second;
        pidx[plot * ARGMAX_BLOCKS + blockIdx.x] = a.at;
    }
}

// kappa: see TSDRGPU_AC_CERT_KAPPA (include/tsdrgpu.h); exact_epoch: the plots are the reference's own bits
__global__ __launch_bounds__(64) void k_argmax_final(const double *__restrict__ pval, const double *__restrict__ psec,
                                                     const int *__restrict__ pidx, const double *__restrict__ plots, int frame_len,
                                                     int line_len, double kappa, int exact_epoch, AcArgHost *__restrict__ out,
                                                     AcArgHost *__restrict__ h_out, const unsigned long long *__restrict__ check)
{
    const int plot = blockIdx.x;
    ArgTop a = {-1.0, -1.0, 0x7fffffff};
    for (int b = threadIdx.x; b < ARGMAX_BLOCKS; b += 64)
        argtop_merge(a, pval[plot * ARGMAX_BLOCKS + b], psec[plot * ARGMAX_BLOCKS + b], pidx[plot * ARGMAX_BLOCKS + b]);
    argtop_wave(a);
    if (threadIdx.x == 0) {
        const int len = plot == 0 ? frame_len : line_len;
        // The host's maximum search starts from lag 0 of the plot and moves on a LARGER value only (PlotVisualizer.java:203-206,236-239:
        // max_val = data[0]; if (val > max_val) ...): a NaN at lag 0 — a window that held a non-finite sample makes every lag NaN,
        // fft.c:49-64 — is never replaced, and a plot without any value that compares larger keeps index 0.
        const double first = len > 0 ? (plot == 0 ? plots[0] : plots[frame_len]) : 0.0;
        const int r = len > 0 ? ((first != first || a.at == 0x7fffffff) ? 0 : a.at) : -1;
        const double r0 = plots[frame_len + line_len];
        const double margin = kappa * r0;
        // a plot of one lag has no runner-up (second stays -1); NaNs compare false and leave the plot uncertified
        // the premise, when this update carries a check of it: one retained window went through the reference's arithmetic
        // as well, and its float32 lags must lie within (kappa / 2) * (its own lag-0 value) of those (NaNs compare false)
        int p_ok = 1;
        double p_err = 0.0, p_r0 = 0.0;
        if (check) {
            p_err = __longlong_as_double((long long)check[0]);
            p_r0 = __longlong_as_double((long long)check[1]);
            p_ok = (p_err <= 0.5 * kappa * p_r0) ? 1 : 0;
        }
        // exact_epoch: 1 = the plots are the reference's bits; -1 = an earlier check of this (float32) epoch failed: nothing
        // of it is certified any more
        const int ok = exact_epoch > 0 || (exact_epoch == 0 && p_ok && (len <= 1 || (a.best - a.second > margin)));
        if (plot == 0) {
            out->premise_checked = check ? 1 : 0; out->premise_ok = p_ok; out->premise_err = p_err; out->premise_r0 = p_r0;
            if (h_out) { h_out->premise_checked = check ? 1 : 0; h_out->premise_ok = p_ok; h_out->premise_err = p_err; h_out->premise_r0 = p_r0; }
        }
        out->idx[plot] = r;
        out->certified[plot] = ok;
        out->best[plot] = a.best;
        out->second[plot] = a.second;
        if (plot == 0) { out->r0 = r0; out->margin = margin; }
        if (h_out) {  // pinned host memory: the result needs no copy engine, the lane no wait for one
            h_out->idx[plot] = r;
            h_out->certified[plot] = ok;
            h_out->best[plot] = a.best;
            h_out->second[plot] = a.second;
            if (plot == 0) { h_out->r0 = r0; h_out->margin = margin; }
        }
    }
}

// ---------------------------------------------------------------------------
// three-trip plan (fft4step.h): nh = N1 * 4096 with N1 = 16 .. 1024
// ---------------------------------------------------------------------------
static bool ac4_supported(uint32_t nh) { return nh >= 4096u * 16u && nh <= 4096u * 1024u; }

template <int LOGN1>
static void launch_ac4_n1(tsdrgpu_t *g, hipStream_t st, const float *src, int in_is_iq, long long stride, int cnt, uint32_t nh, float2 *work,
                          float2 *out, const FftKeep &keep, float *retain)
{
    typedef ColGeom<LOGN1> G;
    const dim3 cgrid(AC4_ROW / G::C, cnt);
    // retain (certified mode, library retention, IQ input): trip 1 also leaves the demodulated windows in the ring
    if (in_is_iq && retain) TSDR_LAUNCH(g, PROF_AC_COLS, st, (k_ac_cols_retain<LOGN1>), cgrid, G::NT, (const void *)src, stride, work, nh, retain);
    else if (in_is_iq) TSDR_LAUNCH(g, PROF_AC_COLS, st, (k_ac_cols<LOGN1, 4, false>), cgrid, G::NT, (const void *)src, stride, work, nh, KEEP_ALL);
    else TSDR_LAUNCH(g, PROF_AC_COLS, st, (k_ac_cols<LOGN1, 3, false>), cgrid, G::NT, (const void *)src, stride, work, nh, KEEP_ALL);
    TSDR_LAUNCH(g, PROF_AC_ROWS, st, k_ac_rows, dim3((1u << LOGN1) / 2u, cnt), 512, work, nh);
    TSDR_LAUNCH(g, PROF_AC_COLS, st, (k_ac_cols<LOGN1, 0, true>), cgrid, G::NT, (const void *)work, (long long)nh, out, nh, keep);
}

static void launch_ac4(tsdrgpu_t *g, hipStream_t st, const float *src, int in_is_iq, long long stride, int cnt, uint32_t nh, float2 *work,
                       float2 *out, const FftKeep &keep, float *retain = nullptr)
{
    int logn1 = 0;
    while ((4096u << logn1) < nh) logn1++;
    switch (logn1) {
        case 4: launch_ac4_n1<4>(g, st, src, in_is_iq, stride, cnt, nh, work, out, keep, retain); break;
        case 5: launch_ac4_n1<5>(g, st, src, in_is_iq, stride, cnt, nh, work, out, keep, retain); break;
        case 6: launch_ac4_n1<6>(g, st, src, in_is_iq, stride, cnt, nh, work, out, keep, retain); break;
        case 7: launch_ac4_n1<7>(g, st, src, in_is_iq, stride, cnt, nh, work, out, keep, retain); break;
        case 8: launch_ac4_n1<8>(g, st, src, in_is_iq, stride, cnt, nh, work, out, keep, retain); break;
        case 9: launch_ac4_n1<9>(g, st, src, in_is_iq, stride, cnt, nh, work, out, keep, retain); break;
        default: launch_ac4_n1<10>(g, st, src, in_is_iq, stride, cnt, nh, work, out, keep, retain); break;
    }
}

extern "C" int tsdrgpu_autocorr_create(tsdrgpu_t *g, tsdrgpu_autocorr_t **out, uint32_t samplerate)
{
    if (!g || !out || samplerate == 0) return TSDRGPU_EINVAL;
    tsdrgpu_autocorr_t *ac = (tsdrgpu_autocorr_t *)calloc(1, sizeof(*ac));
    if (!ac) return TSDRGPU_ENOMEM;
    ac->g = g;
    ac->samplerate = samplerate;
    // frameratedetector.c:20-24,91-95,160
    const int maxlength = samplerate / (double)(55);
    const int minlength = samplerate / (double)(87);
    const int height_maxlength = samplerate / (double)(590 * 55);
    const int height_minlength = samplerate / (double)(1500 * 87);
    ac->frame_lo = minlength;
    ac->frame_len = maxlength - minlength;
    ac->line_lo = height_minlength;
    ac->line_len = height_maxlength - height_minlength;
    ac->capture = (uint32_t)(3.1 * samplerate / (double)(55));
    uint32_t m = 0, sz = ac->capture;  // fft_getrealsize, fft.c:5-11
    while ((sz /= 2) != 0) m++;
    ac->n = 1u << m;
    const size_t L = (size_t)ac->frame_len + ac->line_len;
    if (ac->frame_len <= 0 || ac->line_len <= 0 || (uint32_t)(ac->frame_lo + ac->frame_len) > ac->n) {
        free(ac);
        return tsdr_fail(g, TSDRGPU_EINVAL, "tsdrgpu_autocorr_create", "sample rate too low for the lag windows");
    }
    if (hipMalloc(&ac->d_plots, sizeof(double) * (L + 1)) != hipSuccess || hipMalloc(&ac->d_arg, sizeof(AcArgHost)) != hipSuccess ||
        hipMalloc(&ac->d_pval, 2 * ARGMAX_BLOCKS * sizeof(double)) != hipSuccess ||
        hipMalloc(&ac->d_psec, 2 * ARGMAX_BLOCKS * sizeof(double)) != hipSuccess ||
        hipMalloc(&ac->d_pidx, 2 * ARGMAX_BLOCKS * sizeof(int)) != hipSuccess ||
        hipHostMalloc(&ac->h_arg, sizeof(AcArgHost), hipHostMallocDefault) != hipSuccess ||
        hipEventCreateWithFlags(&ac->ev_arg, hipEventDisableTiming) != hipSuccess) {
        (void)hipFree(ac->d_plots);
        (void)hipFree(ac->d_arg);
        (void)hipFree(ac->d_pval);
        (void)hipFree(ac->d_psec);
        (void)hipFree(ac->d_pidx);
        if (ac->h_arg) (void)hipHostFree(ac->h_arg);
        free(ac);
        return tsdr_fail(g, TSDRGPU_ENOMEM, "tsdrgpu_autocorr_create", "plots");
    }
    memset(ac->h_arg, 0, sizeof(AcArgHost));
    ac->st = g->stream;
    ac->replay_rec = -1;
    *out = ac;
    return tsdrgpu_autocorr_reset(ac);
}

extern "C" void tsdrgpu_autocorr_destroy(tsdrgpu_autocorr_t *ac)
{
    if (!ac) return;
    (void)hipStreamSynchronize(ac->g->stream);
    (void)hipStreamSynchronize(ac->g->stream2);
    (void)hipStreamSynchronize(ac->g->bg);
    (void)hipFree(ac->d_plots);
    (void)hipFree(ac->d_snapshot);
    (void)hipFree(ac->d_a);
    (void)hipFree(ac->d_b);
    (void)hipFree(ac->d_expand);
    (void)hipFree(ac->d_arg);
    (void)hipFree(ac->d_pval);
    (void)hipFree(ac->d_psec);
    (void)hipFree(ac->d_pidx);
    (void)hipHostFree(ac->h_arg);
    (void)hipEventDestroy(ac->ev_arg);
    (void)hipFree(ac->d_tw);
    (void)hipFree(ac->d_xz);
    (void)hipFree(ac->d_xmag);
    if (ac->ring) { ac->ring->stop(); delete ac->ring; }
    (void)hipFree(ac->d_check);
    free(ac->log);
    free(ac);
}

extern "C" int tsdrgpu_autocorr_reset(tsdrgpu_autocorr_t *ac)
{
    if (!ac) return TSDRGPU_EINVAL;
    tsdrgpu_t *g = ac->g;
    ac->calls = 0;  // extbuffer "cleartozero" semantics, extbuffer.c:68-81
    // a new epoch: back to the float32 transform, nothing retained
    ac->epoch_exact = 0;
    ac->log_count = 0;
    ac->ring_count = 0;
    ac->replay_rec = -1;
    if (ac->recheck_next) ac->since_check = -1;  // the last epoch broke the premise: this one is checked at its first update
    ac->recheck_next = 0;
    ac->premise_broken = 0;
    HIP_TRY(g, hipMemsetAsync(ac->d_plots, 0, sizeof(double) * ((size_t)ac->frame_len + ac->line_len + 1), ac->st));
    return TSDRGPU_OK;
}

extern "C" int tsdrgpu_autocorr_geometry(tsdrgpu_autocorr_t *ac, int32_t *frame_lo, int32_t *frame_len, int32_t *line_lo,
                                         int32_t *line_len, uint32_t *capture, uint32_t *fft_n)
{
    if (!ac) return TSDRGPU_EINVAL;
    if (frame_lo) *frame_lo = ac->frame_lo;
    if (frame_len) *frame_len = ac->frame_len;
    if (line_lo) *line_lo = ac->line_lo;
    if (line_len) *line_len = ac->line_len;
    if (capture) *capture = ac->capture;
    if (fft_n) *fft_n = ac->n;
    return TSDRGPU_OK;
}

// the exact form's resources: the twiddle table (built on the host, once per object) and AC_XBATCH windows of work space
static int ac_ensure_exact(tsdrgpu_autocorr_t *ac)
{
    tsdrgpu_t *g = ac->g;
    if (ac->d_tw) return TSDRGPU_OK;
    int rc = fftx_build_table(g, ac->n, &ac->d_tw);
    if (rc) return rc;
    if (hipMalloc(&ac->d_xz, sizeof(float2) * (size_t)ac->n * AC_XBATCH) != hipSuccess ||
        hipMalloc(&ac->d_xmag, sizeof(float2) * (size_t)ac->n * AC_XBATCH) != hipSuccess) {
        (void)hipFree(ac->d_xz);
        (void)hipFree(ac->d_tw);
        ac->d_xz = nullptr;
        ac->d_tw = nullptr;
        return tsdr_fail(g, TSDRGPU_ENOMEM, "tsdrgpu_autocorr", "work buffers of the exact form");
    }
    return TSDRGPU_OK;
}

// `nwindows` windows in the reference's own arithmetic (tsdrgpu_fftx.hip), accumulated into the plots
static int ac_run_exact(tsdrgpu_autocorr_t *ac, const float *d_in, int in_is_iq, long long stride, int nwindows, int mode)
{
    tsdrgpu_t *g = ac->g;
    int rc = ac_ensure_exact(ac);
    if (rc) return rc;
    for (int w0 = 0; w0 < nwindows; w0 += AC_XBATCH) {
        const int cnt = nwindows - w0 < AC_XBATCH ? nwindows - w0 : AC_XBATCH;
        const float *src = d_in + (size_t)w0 * (size_t)stride * AC_KIND_FLOATS(in_is_iq);
        // (only the call's final window is stored whole: tsdrgpu_autocorr_last_corr; the others keep their lag windows)
        rc = fftx_autocorr(g, ac->st, src, in_is_iq, stride, cnt, ac->n, ac->d_tw, ac->d_xz, ac->d_xmag, ac->frame_lo, ac->frame_len,
                           ac->line_lo, ac->line_len, ac->d_plots, (unsigned long long)(ac->calls + w0), mode,
                           (w0 + cnt == nwindows) ? cnt - 1 : -1);
        if (rc) return rc;
        ac->d_last = ac->d_xz + (size_t)(cnt - 1) * ac->n;  // the whole complex correlation of the last window
        ac->last_exact = 1;
    }
    ac->calls += (uint64_t)nwindows;
    return TSDRGPU_OK;
}

// `nwindows` windows through the float32 transform (three-trip plan where it applies), accumulated into the plots
// retain_to (IQ input on the three-trip plan only, see ac_retain_fused): where trip 1 leaves the windows' demodulated samples,
// n floats per window
static bool ac_retain_fused(const tsdrgpu_autocorr_t *ac, int in_is_iq);
static int ac_run_fast(tsdrgpu_autocorr_t *ac, const float *d_in, int in_is_iq, long long stride, int nwindows, int mode, float *retain_to = nullptr)
{
    tsdrgpu_t *g = ac->g;
    // windows are transformed at most AC_SUBBATCH at a time.  A launch of the three-trip plan is only ~3 rounds of
    // resident workgroups per 6 windows, so every launch pays a ramp and a drain: measured at 17 windows per pass,
    // 6+6+5 -> group at 0.565 of the roofline, 9+8 -> 0.585, one launch of 17 -> 0.62 — but a caller's small kernels on
    // another lane (the sync chain in bench.py's split run) then find free CUs less often (0.31 -> 0.49 ms) and become
    // the critical path; 9 is where the whole pass is fastest
    // TSDRGPU_AC_SPLIT="8,8,1": the sub-batches of a call, spelled out (A/B runs; used when they add up to the call's windows)
    static const std::vector<int> split_env = [] {
        std::vector<int> v;
        const char *e = getenv("TSDRGPU_AC_SPLIT");
        while (e && *e) {
            const int k = atoi(e);
            if (k > 0 && k <= 64) v.push_back(k);
            while (*e && *e != ',') e++;
            if (*e == ',') e++;
        }
        return v;
    }();
    int split_sum = 0, split_max = 0;
    for (int k : split_env) { split_sum += k; split_max = k > split_max ? k : split_max; }
    const bool use_split = !split_env.empty() && split_sum == nwindows;
    const int AC_SUBBATCH = use_split ? split_max : ac_subbatch(ac->n);
    const int sub = nwindows < AC_SUBBATCH ? nwindows : AC_SUBBATCH;
    if (ac->cap_windows < sub) {
        (void)hipStreamSynchronize(g->stream);
        (void)hipStreamSynchronize(g->stream2);
        (void)hipStreamSynchronize(g->bg);
        (void)hipFree(ac->d_a);
        (void)hipFree(ac->d_b);
        ac->d_a = ac->d_b = nullptr;
        ac->cap_windows = 0;
        if (hipMalloc(&ac->d_a, sizeof(float2) * (size_t)(ac->n / 2) * AC_SUBBATCH) != hipSuccess ||
            hipMalloc(&ac->d_b, sizeof(float2) * (size_t)(ac->n / 2) * AC_SUBBATCH) != hipSuccess)
            return tsdr_fail(g, TSDRGPU_ENOMEM, "tsdrgpu_autocorr_run", "work buffers");
        ac->cap_windows = AC_SUBBATCH;
    }
    const uint32_t nh = ac->n / 2;
    const int L = ac->frame_len + ac->line_len + 1;  // + the lag-0 entry
    float2 *corr = nullptr;
    int last_count = 0;
    int parts = (nwindows + AC_SUBBATCH - 1) / AC_SUBBATCH;
    // (A/B switch.  Re-measured in round 5 with the fused frame path, same box: 9 + 8 -> 91.8 / 92.0 GS/s, group 0.3926 / 0.3912 ms;
    // one launch of 17 -> 92.4 / 91.9, 0.3898 / 0.3893 (rows 0.118 instead of 0.127 ms, columns 0.250 instead of 0.243);
    // 6 + 6 + 5 -> 90.3, 0.403; 12 + 5 -> 91.9, 0.395: nothing to choose between 9 + 8 and 17, so 9 + 8 stays)
    static const bool one_launch = [] { const char *e = getenv("TSDRGPU_AC_ONE_LAUNCH"); return e && e[0] == '1'; }();
    if (parts < 2 && nwindows > 9 && !one_launch) parts = 2;  // a pass of 17 short windows stays 9 + 8: one launch of 17 measured 7 % slower for the pass
    const int per_part = (nwindows + parts - 1) / parts;  // equal sub-batches (17 -> 9,8)
    size_t split_i = 0;
    for (int w0 = 0, step = 0; w0 < nwindows; w0 += step) {
        const int cnt = use_split ? split_env[split_i++] : ((nwindows - w0 < per_part) ? (nwindows - w0) : per_part);
        step = cnt;
        const float *src = d_in + (size_t)w0 * (size_t)stride * (in_is_iq ? 2 : 1);
        // fft_autocorrelation (fft.c:49-64) on the real window, packed two samples per complex point
        const PassPlan plan = plan_passes(nh);
        const int R_last = plan.radix[plan.count - 1];
        const unsigned Ns_last = nh / R_last;
        const bool fused = nh >= 4096 && plan.count >= 2 && Ns_last >= 2u * (2048u / R_last);
        float2 *corr_;
        // lags stored by the last pass: complex point m holds lags 2m, 2m+1 (point 0, i.e. lag 0, always); the call's
        // final window is stored whole for tsdrgpu_autocorr_last_corr
        FftKeep keep;
        keep.on = 1;
        keep.full_b = (w0 + cnt == nwindows) ? cnt - 1 : -1;
        keep.lo0 = (unsigned)ac->frame_lo / 2;
        keep.hi0 = (unsigned)(ac->frame_lo + ac->frame_len + 1) / 2;
        keep.lo1 = (unsigned)ac->line_lo / 2;
        keep.hi1 = (unsigned)(ac->line_lo + ac->line_len + 1) / 2;
        if (ac4_supported(nh) && !ac->plan5) {
            // three trips (fft4step.h): columns -> row pairs (in place) -> columns
            launch_ac4(g, ac->st, src, in_is_iq, stride, cnt, nh, ac->d_a, ac->d_b, keep, retain_to ? retain_to + (size_t)w0 * ac->n : nullptr);
            corr_ = ac->d_b;
        } else if (fused) {
            // forward passes but the last ...
            float2 *z = run_fft_range(g, src, in_is_iq ? 4 : 3, stride, ac->d_a, ac->d_b, nh, cnt, plan.radix, plan.count, 0,
                                      plan.count - 1, 1, 0, 0, false, 1.0f, ac->st);
            // ... last forward pass + split + first inverse pass in one kernel ...
            float2 *mid = (z == ac->d_a) ? ac->d_b : ac->d_a;
            const dim3 grid(Ns_last / (2 * (2048 / R_last)) + 1, cnt);  // + workgroup 0 for column 0
            switch (R_last / 16) {
                case 1: TSDR_LAUNCH(g, PROF_AC_SPLIT, ac->st, (k_ac_mid<1>), grid, 256, z, mid, nh); break;
                case 2: TSDR_LAUNCH(g, PROF_AC_SPLIT, ac->st, (k_ac_mid<2>), grid, 256, z, mid, nh); break;
                case 4: TSDR_LAUNCH(g, PROF_AC_SPLIT, ac->st, (k_ac_mid<4>), grid, 256, z, mid, nh); break;
                case 8: TSDR_LAUNCH(g, PROF_AC_SPLIT, ac->st, (k_ac_mid<8>), grid, 256, z, mid, nh); break;
                default: TSDR_LAUNCH(g, PROF_AC_SPLIT, ac->st, (k_ac_mid<16>), grid, 256, z, mid, nh); break;
            }
            // ... the remaining inverse passes, radices in reverse order (the fused kernel did radix R_last, Ns = 1)
            int rev[32];
            for (int i = 0; i < plan.count; i++) rev[i] = plan.radix[plan.count - 1 - i];
            // the last pass stores only the two lag windows (complex point m holds lags 2m, 2m+1), plus the
            // whole correlation of the call's final window for tsdrgpu_autocorr_last_corr
            keep.on = plan.count >= 2 ? 1 : 0;  // with 2 passes the "last" one is pass 1 of rev[], still Ns > 1
            corr_ = run_fft_range(g, mid, 0, nh, ac->d_a, ac->d_b, nh, cnt, rev, plan.count, 1, plan.count, (unsigned)R_last, 1, 1,
                                  false, 1.0f, ac->st, keep);
        } else {
            float2 *zf = run_fft(g, src, in_is_iq ? 4 : 3, stride, ac->d_a, ac->d_b, nh, cnt, 0, false, 1.0f, ac->st);
            TSDR_LAUNCH(g, PROF_AC_SPLIT, ac->st, k_ac_split, dim3((nh / 2 + 1 + 255) / 256, cnt), 256, zf, nh);
            corr_ = run_fft(g, zf, 0, nh, ac->d_a, ac->d_b, nh, cnt, 1, false, 1.0f, ac->st);
        }
        corr = corr_;
        KERNEL_CHECK(g, "fft passes");
        TSDR_LAUNCH(g, PROF_ACCUMULATE, ac->st, k_accumulate, (L + 255) / 256, 256, (const float *)corr, ac->n, cnt, ac->frame_lo, ac->frame_len, ac->line_lo,
                                                          ac->line_len, ac->d_plots, (unsigned long long)(ac->calls + w0), mode);
        KERNEL_CHECK(g, "k_accumulate");
        last_count = cnt;
    }
    const int nwindows_last = last_count;
    ac->calls += (uint64_t)nwindows;
    ac->d_last = corr + (size_t)(nwindows_last - 1) * nh;
    ac->last_exact = 0;
    return TSDRGPU_OK;
}

static bool ac_retain_fused(const tsdrgpu_autocorr_t *ac, int in_is_iq)
{
    static const bool off = [] { const char *e = getenv("TSDRGPU_RETAIN_COPY"); return e && e[0] == '1'; }();  // A/B: the copy kernel
    return in_is_iq && !off && ac4_supported(ac->n / 2) && !ac->plan5;
}

// Replays the epoch's windows (every run since the last reset) in the reference's own arithmetic: the plots then hold
// what an exact run of the same calls would have left, and the rest of the epoch runs exact.
extern "C" int tsdrgpu_autocorr_promote_step(tsdrgpu_autocorr_t *ac, int max_windows, int *h_remaining);
extern "C" int tsdrgpu_autocorr_promote(tsdrgpu_autocorr_t *ac)
{
    if (!ac) return TSDRGPU_EINVAL;
    tsdrgpu_t *g = ac->g;
    if (!ac->certify) return tsdr_fail(g, TSDRGPU_ESTATE, "tsdrgpu_autocorr_promote", "certified mode is off (tsdrgpu_autocorr_set_certify)");
    int remaining = 0;
    return tsdrgpu_autocorr_promote_step(ac, 0x7fffffff, &remaining);
}

// The same replay a bounded number of windows at a time, for a host that must not block its queue for the whole of a long
// epoch (2048 retained windows of 2^22 samples are 0.28 s of transforms): the first call opens the replay (plots zeroed, the
// epoch marked exact), every call replays up to max_windows windows in order, *h_remaining tells how many are left.  Until
// it reaches 0 the plots are partial: tsdrgpu_autocorr_run returns TSDRGPU_ESTATE and the host skips (or defers) its windows
// — the reference's detector thread also correlates only the windows it has time for (frameratedetector.c:128-187).
extern "C" int tsdrgpu_autocorr_promote_step(tsdrgpu_autocorr_t *ac, int max_windows, int *h_remaining)
{
    if (!ac || max_windows < 1) return ac ? tsdr_fail(ac->g, TSDRGPU_EINVAL, "tsdrgpu_autocorr_promote_step", "bad argument") : TSDRGPU_EINVAL;
    tsdrgpu_t *g = ac->g;
    if (h_remaining) *h_remaining = 0;
    if (!ac->certify) return tsdr_fail(g, TSDRGPU_ESTATE, "tsdrgpu_autocorr_promote", "certified mode is off (tsdrgpu_autocorr_set_certify)");
    if (ac->exact || (ac->epoch_exact && ac->replay_rec < 0)) return TSDRGPU_OK;
    int rc;
    if (ac->replay_rec < 0) {  // open the replay
        if ((rc = ac_ensure_exact(ac))) return rc;
        HIP_TRY(g, hipMemsetAsync(ac->d_plots, 0, sizeof(double) * ((size_t)ac->frame_len + ac->line_len + 1), ac->st));
        ac->calls = 0;
        ac->epoch_exact = 1;
        ac->premise_broken = 0;
        ac->promotions++;
        ac->replay_rec = 0;
        ac->replay_win = 0;
    }
    int budget = max_windows;
    while (ac->replay_rec < ac->log_count && budget > 0) {
        const AcLogRec &r = ac->log[ac->replay_rec];
        const int left = r.nwindows - ac->replay_win;
        const int take = left < budget ? left : budget;
        const float *src = r.src + (size_t)ac->replay_win * (size_t)r.stride * AC_KIND_FLOATS(r.is_iq);
        if ((rc = ac_run_exact(ac, src, r.is_iq, r.stride, take, r.mode))) return rc;
        budget -= take;
        ac->replay_win += take;
        if (ac->replay_win == r.nwindows) { ac->replay_rec++; ac->replay_win = 0; }
    }
    if (ac->replay_rec >= ac->log_count) {
        ac->log_count = 0;
        ac->ring_count = 0;  // the ring's windows are read by the replay queued above; later runs of this epoch do not retain
        ac->replay_rec = -1;
        return TSDRGPU_OK;
    }
    if (h_remaining) {
        long long rem = (long long)ac->log[ac->replay_rec].nwindows - ac->replay_win;
        for (int i = ac->replay_rec + 1; i < ac->log_count; i++) rem += ac->log[i].nwindows;
        *h_remaining = rem > 0x7fffffff ? 0x7fffffff : (int)rem;
    }
    return TSDRGPU_OK;
}

// where the next windows of the epoch go in the ring: as many of `nwindows` as the current segment still holds (*take), or
// nullptr when there is no room — the ring is full, or the allocator has not got that far yet
static float *ac_ring_slot(tsdrgpu_autocorr_t *ac, int nwindows, int *pos_out, int *take)
{
    AcRing *rg = ac->ring;
    if (!rg) return nullptr;
    const int W = rg->seg_windows;
    const int pos = ac->ring_count;
    const int si = pos / W;
    if (si >= rg->nseg_max || si >= rg->ready.load(std::memory_order_acquire)) return nullptr;
    rg->ask(si + 3);  // the next two segments are made while this one fills
    *pos_out = pos;
    *take = nwindows < W - pos % W ? nwindows : W - pos % W;
    return rg->seg[si] + (size_t)(pos % W) * ac->n;
}

extern "C" int tsdrgpu_autocorr_run(tsdrgpu_autocorr_t *ac, const float *d_in, int in_is_iq, int64_t stride, int nwindows, int mode)
{
    if (!ac || !d_in || nwindows < 0 || stride < 0) return ac ? tsdr_fail(ac->g, TSDRGPU_EINVAL, "tsdrgpu_autocorr_run", "bad argument") : TSDRGPU_EINVAL;
    if (nwindows == 0) return TSDRGPU_OK;
    tsdrgpu_t *g = ac->g;
    if (nwindows > 65535) return tsdr_fail(g, TSDRGPU_EINVAL, "tsdrgpu_autocorr_run", "too many windows in one call");
    // a flag at the interface, a source KIND inside (0 magnitudes, 1 IQ; 2 / 3 are internal to the exact replay): any truthy
    // value means IQ on every path, the fast and the exact one alike
    in_is_iq = in_is_iq ? 1 : 0;
    if (ac->st != g->stream) {
        // side stream: everything already queued on the main stream (e.g. the producer of d_in) comes first
        HIP_TRY(g, hipEventRecord(g->fork, g->stream));
        HIP_TRY(g, hipStreamWaitEvent(ac->st, g->fork, 0));
    }
    if (ac->replay_rec >= 0) return tsdr_fail(g, TSDRGPU_ESTATE, "tsdrgpu_autocorr_run", "an incremental promotion is in progress (tsdrgpu_autocorr_promote_step)");
    if (ac->exact || ac->epoch_exact) return ac_run_exact(ac, d_in, in_is_iq, stride, nwindows, mode);
    if (!ac->certify) return ac_run_fast(ac, d_in, in_is_iq, stride, nwindows, mode);
    int rc;
    if (ac->certify == 2) {
        // the caller keeps the windows: remember where they are
        if (ac->log_count >= ac->log_cap) {  // an epoch of more calls than the log holds continues in the exact form
            if ((rc = tsdrgpu_autocorr_promote(ac))) return rc;
            return ac_run_exact(ac, d_in, in_is_iq, stride, nwindows, mode);
        }
        const AcLogRec r = {d_in, in_is_iq, (long long)stride, nwindows, mode};
        ac->log[ac->log_count++] = r;
        return ac_run_fast(ac, d_in, in_is_iq, stride, nwindows, mode);
    }
    // the library keeps them: the first n samples of every window, demodulated, into the ring; the float32 transform
    // then reads the ring.  An epoch that outgrows the ring is replayed exactly once and continues in the exact form.
    // (a call's windows go into the ring segment by segment: one piece and one log record per segment touched)
    for (int done = 0; done < nwindows;) {
        const float *src = d_in + (size_t)done * (size_t)stride * (in_is_iq ? 2 : 1);
        int pos = 0, take = 0;
        float *slot = ac->log_count >= ac->log_cap ? nullptr : ac_ring_slot(ac, nwindows - done, &pos, &take);
        if (!slot) {  // no room (or not allocated yet): what was run so far is replayed, the rest of the epoch runs exact
            if ((rc = tsdrgpu_autocorr_promote(ac))) return rc;
            return ac_run_exact(ac, src, in_is_iq, stride, nwindows - done, mode);
        }
        // From interleaved IQ on the three-trip plan, trip 1 itself fills the ring (it holds every sample in registers:
        // k_ac_cols_retain) — 8N read + 4N + 4N written per window where a copy kernel in front (k_fftx_retain) read the IQ
        // twice and the ring once more: 8N + 4N, then 4N + 4N.  What it leaves is am_demod's sum of squares, re*re + im*im in the
        // reference's own roundings, NOT the root: the correctly rounded root costs ten instructions a sample and is only ever
        // needed by a replay, whose first trip takes it on its loads (AC_KIND_SUMSQ).  Magnitude input, and sizes outside the
        // plan, keep the copy.
        const bool fused = ac_retain_fused(ac, in_is_iq);
        if (!fused && (rc = fftx_retain(g, ac->st, src, in_is_iq, (long long)stride, take, ac->n, slot))) return rc;
        const AcLogRec r = {slot, fused ? AC_KIND_SUMSQ : 0, (long long)ac->n, take, mode};
        ac->log[ac->log_count++] = r;
        ac->ring_count = pos + take;
        if (fused) rc = ac_run_fast(ac, src, 1, (long long)stride, take, mode, slot);
        else rc = ac_run_fast(ac, slot, 0, (long long)ac->n, take, mode);
        if (rc) return rc;
        done += take;
    }
    return TSDRGPU_OK;
}

extern "C" int tsdrgpu_autocorr_plots(tsdrgpu_autocorr_t *ac, double *h_frame, double *h_line, uint64_t *h_calls)
{
    if (!ac) return TSDRGPU_EINVAL;
    tsdrgpu_t *g = ac->g;
    if (h_frame) HIP_TRY(g, hipMemcpyAsync(h_frame, ac->d_plots, sizeof(double) * ac->frame_len, hipMemcpyDeviceToHost, ac->st));
    if (h_line) HIP_TRY(g, hipMemcpyAsync(h_line, ac->d_plots + ac->frame_len, sizeof(double) * ac->line_len, hipMemcpyDeviceToHost, ac->st));
    HIP_TRY(g, hipStreamSynchronize(ac->st));
    if (h_calls) *h_calls = ac->calls;
    return TSDRGPU_OK;
}

extern "C" int tsdrgpu_autocorr_plots_async(tsdrgpu_autocorr_t *ac, double *h_frame, double *h_line, uint64_t *h_calls)
{
    if (!ac) return TSDRGPU_EINVAL;
    tsdrgpu_t *g = ac->g;
    if (h_frame) HIP_TRY(g, hipMemcpyAsync(h_frame, ac->d_plots, sizeof(double) * ac->frame_len, hipMemcpyDeviceToHost, ac->st));
    if (h_line) HIP_TRY(g, hipMemcpyAsync(h_line, ac->d_plots + ac->frame_len, sizeof(double) * ac->line_len, hipMemcpyDeviceToHost, ac->st));
    if (h_calls) *h_calls = ac->calls;  // host-side count: known without the device
    return TSDRGPU_OK;
}

// A device-side copy of the plots as they are at this point of the object's lane (a kernel, not a DMA: the lane never
// waits for a copy engine).  Whoever wants them on the host copies the snapshot on a lane of its own once an event
// recorded behind this call has fired.
extern "C" int tsdrgpu_autocorr_plots_snapshot(tsdrgpu_autocorr_t *ac, const double **d_snapshot, uint64_t *h_calls)
{
    if (!ac || !d_snapshot) return TSDRGPU_EINVAL;
    tsdrgpu_t *g = ac->g;
    const size_t L = (size_t)ac->frame_len + ac->line_len;  // (the lag-0 entry behind them stays on the device)
    if (!ac->d_snapshot && hipMalloc(&ac->d_snapshot, sizeof(double) * L) != hipSuccess)
        return tsdr_fail(g, TSDRGPU_ENOMEM, "tsdrgpu_autocorr_plots_snapshot", "snapshot");
    HIP_TRY(g, hipMemcpyAsync(ac->d_snapshot, ac->d_plots, sizeof(double) * L, hipMemcpyDeviceToDevice, ac->st));
    *d_snapshot = ac->d_snapshot;
    if (h_calls) *h_calls = ac->calls;
    return TSDRGPU_OK;
}

extern "C" int tsdrgpu_autocorr_device_plots(tsdrgpu_autocorr_t *ac, double **d_plots, int64_t *count)
{
    if (!ac) return TSDRGPU_EINVAL;
    if (d_plots) *d_plots = ac->d_plots;
    if (count) *count = (int64_t)ac->frame_len + ac->line_len;
    return TSDRGPU_OK;
}

// What a caller that runs its own collective has to sum over the ranks: the lags AND the accumulated lag-0 value behind
// them (the scale R0 of the argmax certificate) — finalize_sums divides all of it by the global window count, so an R0
// left rank-local would make the certificate's margin too small by a factor `world`.
extern "C" int tsdrgpu_autocorr_device_sums(tsdrgpu_autocorr_t *ac, double **d_sums, int64_t *count)
{
    if (!ac) return TSDRGPU_EINVAL;
    if (d_sums) *d_sums = ac->d_plots;
    if (count) *count = (int64_t)ac->frame_len + ac->line_len + 1;
    return TSDRGPU_OK;
}

extern "C" int tsdrgpu_autocorr_finalize_sums(tsdrgpu_autocorr_t *ac, uint64_t total_windows)
{
    if (!ac || total_windows == 0) return TSDRGPU_EINVAL;
    tsdrgpu_t *g = ac->g;
    const int L = ac->frame_len + ac->line_len + 1;  // the lag-0 entry scales with the plots
    TSDR_LAUNCH(g, PROF_ACCUMULATE, ac->st, k_scale_plots, (L + 255) / 256, 256, ac->d_plots, L, (double)total_windows);
    KERNEL_CHECK(g, "k_scale_plots");
    ac->calls = total_windows;
    return TSDRGPU_OK;
}

// ---------------------------------------------------------------------------
// The certificate's premise, checked while the detector runs.  The certificate (best - runner_up > KAPPA * R0) proves the
// float32 argmax to be the reference's IF every float32 plot value lies within (KAPPA / 2) * R0 of the reference's.  That
// holds with a wide margin on everything measured (DESIGN.md section 2 has the error model), but it is a statement about
// rounding errors, not a theorem — so it is also CHECKED: on the first plot update after tsdrgpu_autocorr_set_certify and
// on every check_every-th one after that (TSDRGPU_AC_CHECK_EVERY, default 16; 0 = never) the newest retained window goes through
// the reference's arithmetic as well (one exact transform, ~0.13 ms at 2^22) and its lags are compared on the device
// with the float32 transform's: max |fast - exact| <= (KAPPA / 2) * (that window's exact lag-0 value), or the certificate
// of this update fails — the caller promotes the epoch like for any other uncertified plot.  A plot is a mean over
// windows of per-window values and R0 the same mean of the per-window lag-0 values, so the per-window bound carries over.
// ---------------------------------------------------------------------------
// (PREMISE_BLOCKS workgroups stride over the lags and each ends in ONE atomic: one 64-bit atomicMax per wave — 10 500 of them on a
// single address for the 671 000 lags of a 100 MS/s detector — took 122 us, as long as the exact transform the check runs.)
#define PREMISE_BLOCKS 256
__global__ __launch_bounds__(256) void k_premise_diff(const float *__restrict__ fast, const float2 *__restrict__ exact, int frame_lo, int frame_len,
                                                      int line_lo, int line_len, unsigned long long *__restrict__ check)
{
    double d = 0.0;
    bool nan = false;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i <= frame_len + line_len; i += gridDim.x * blockDim.x) {
        const int lag = (i < frame_len) ? (frame_lo + i) : (i < frame_len + line_len ? line_lo + (i - frame_len) : 0);
        const float2 v = exact[lag];
        const double re = v.x, im = v.y;
        const double want = sqrt(re * re + im * im);       // what k_fftx_accumulate folds into the plots
        const double got = fabs((double)fast[lag]);        // what k_accumulate folds
        if (i == frame_len + line_len) {
            check[1] = (unsigned long long)__double_as_longlong(want);  // lag 0: the scale of the bound, not one of the plots' lags
        } else {
            const double e = fabs(got - want);
            if (!(e == e)) nan = true;
            else d = e > d ? e : d;
        }
    }
    if (nan) d = __longlong_as_double(0x7ff8000000000000LL);  // NaN: larger than everything below, fails the bound
    // non-negative doubles order like their bit patterns (NaN above infinity)
    unsigned long long bits = (unsigned long long)__double_as_longlong(d);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long ob = __shfl_down(bits, o, 64);
        bits = ob > bits ? ob : bits;
    }
    __shared__ unsigned long long wb[4];
    if ((threadIdx.x & 63) == 0) wb[threadIdx.x >> 6] = bits;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) bits = wb[w] > bits ? wb[w] : bits;
        if (bits) atomicMax(check, bits);
    }
}

// queues the check behind the last run; 0 = not due / not possible, 1 = queued (d_check is valid behind it), < 0 error
static int ac_premise_check(tsdrgpu_autocorr_t *ac)
{
    tsdrgpu_t *g = ac->g;
    if (!ac->certify || ac->exact || ac->epoch_exact || ac->premise_broken || ac->check_every <= 0 || ac->log_count <= 0 || !ac->d_last || ac->last_exact) return 0;
    if (ac->since_check >= 0 && ac->since_check + 1 < ac->check_every) { ac->since_check++; return 0; }
    int rc = ac_ensure_exact(ac);
    if (rc) return rc;
    if (!ac->d_check && hipMalloc(&ac->d_check, 2 * sizeof(unsigned long long)) != hipSuccess) return tsdr_fail(g, TSDRGPU_ENOMEM, "tsdrgpu_autocorr", "premise check");
    const AcLogRec &r = ac->log[ac->log_count - 1];  // its final window is the one d_last holds whole
    const float *src = r.src + (size_t)(r.nwindows - 1) * (size_t)r.stride * AC_KIND_FLOATS(r.is_iq);
    if ((rc = fftx_correlate(g, ac->st, src, r.is_iq, r.stride, 1, ac->n, ac->d_tw, ac->d_xz, ac->d_xmag))) return rc;
    if (hipMemsetAsync(ac->d_check, 0, 2 * sizeof(unsigned long long), ac->st) != hipSuccess) return tsdr_fail(g, TSDRGPU_EHIP, "tsdrgpu_autocorr", "premise check");
    const int L = ac->frame_len + ac->line_len + 1;
    TSDR_LAUNCH(g, PROF_ARGMAX, ac->st, k_premise_diff, ((L + 255) / 256 < PREMISE_BLOCKS ? (L + 255) / 256 : PREMISE_BLOCKS), 256, (const float *)ac->d_last, (const float2 *)ac->d_xz, ac->frame_lo, ac->frame_len,
                ac->line_lo, ac->line_len, ac->d_check);
    if (hipGetLastError() != hipSuccess) return tsdr_fail(g, TSDRGPU_EHIP, "tsdrgpu_autocorr", "k_premise_diff");
    ac->since_check = 0;
    ac->premise_checks++;
    return 1;
}

// queue the two-stage argmax of the current plots; its last kernel writes the result and the certificate to pinned memory
extern "C" int tsdrgpu_autocorr_argmax_async(tsdrgpu_autocorr_t *ac)
{
    if (!ac) return TSDRGPU_EINVAL;
    tsdrgpu_t *g = ac->g;
    if (ac->arg_pending) return tsdr_fail(g, TSDRGPU_ESTATE, "tsdrgpu_autocorr_argmax_async", "the previous result was not collected");
    if (ac->replay_rec >= 0) return tsdr_fail(g, TSDRGPU_ESTATE, "tsdrgpu_autocorr_argmax", "an incremental promotion is in progress: the plots are partial");
    const int checked = ac_premise_check(ac);
    if (checked < 0) return checked;
    TSDR_LAUNCH(g, PROF_ARGMAX, ac->st, k_argmax_partial, dim3(ARGMAX_BLOCKS, 2), 256, ac->d_plots, ac->frame_len, ac->line_len, ac->d_pval, ac->d_psec,
                ac->d_pidx);
    TSDR_LAUNCH(g, PROF_ARGMAX, ac->st, k_argmax_final, 2, 64, ac->d_pval, ac->d_psec, ac->d_pidx, ac->d_plots, ac->frame_len, ac->line_len,
                (double)TSDRGPU_AC_CERT_KAPPA, (ac->exact || ac->epoch_exact) ? 1 : (ac->premise_broken ? -1 : 0), ac->d_arg, ac->h_arg,
                (const unsigned long long *)(checked ? ac->d_check : nullptr));
    KERNEL_CHECK(g, "k_argmax");
    HIP_TRY(g, hipEventRecord(ac->ev_arg, ac->st));
    ac->arg_pending = 1;
    return TSDRGPU_OK;
}

extern "C" int tsdrgpu_autocorr_argmax_result(tsdrgpu_autocorr_t *ac, int32_t *frame_idx, int32_t *line_idx)
{
    if (!ac) return TSDRGPU_EINVAL;
    tsdrgpu_t *g = ac->g;
    if (!ac->arg_pending) return tsdr_fail(g, TSDRGPU_ESTATE, "tsdrgpu_autocorr_argmax_result", "no argmax was queued");
    HIP_TRY(g, hipEventSynchronize(ac->ev_arg));
    ac->arg_pending = 0;
    ac->res = *ac->h_arg;
    if (ac->res.premise_checked && !ac->res.premise_ok) {
        ac->premise_failures++;
        ac->premise_broken = 1;  // sticks until the epoch is promoted or reset
        ac->recheck_next = 1;
    }
    if (frame_idx) *frame_idx = ac->res.idx[0];
    if (line_idx) *line_idx = ac->res.idx[1];
    return TSDRGPU_OK;
}

extern "C" int tsdrgpu_autocorr_argmax(tsdrgpu_autocorr_t *ac, int32_t *frame_idx, int32_t *line_idx)
{
    const int rc = tsdrgpu_autocorr_argmax_async(ac);
    if (rc) return rc;
    return tsdrgpu_autocorr_argmax_result(ac, frame_idx, line_idx);
}

extern "C" int tsdrgpu_autocorr_certificate(tsdrgpu_autocorr_t *ac, tsdrgpu_ac_certificate_t *out)
{
    if (!ac || !out) return TSDRGPU_EINVAL;
    memset(out, 0, sizeof(*out));
    out->frame_certified = ac->res.certified[0];
    out->line_certified = ac->res.certified[1];
    out->frame_best = ac->res.best[0];
    out->frame_runner_up = ac->res.second[0];
    out->line_best = ac->res.best[1];
    out->line_runner_up = ac->res.second[1];
    out->r0 = ac->res.r0;
    out->margin = ac->res.margin;
    out->exact_epoch = (ac->exact || ac->epoch_exact) ? 1 : 0;
    out->promotions = ac->promotions;
    out->premise_checked = ac->res.premise_checked;
    out->premise_ok = ac->res.premise_ok;
    out->premise_err = ac->res.premise_err;
    out->premise_r0 = ac->res.premise_r0;
    out->premise_checks = ac->premise_checks;
    out->premise_failures = ac->premise_failures;
    return TSDRGPU_OK;
}

// Certified argmax in one call: the argmax, and — when the certificate fails — the promotion of the epoch and the argmax
// of the exact plots.  Synchronises.
extern "C" int tsdrgpu_autocorr_argmax_certified(tsdrgpu_autocorr_t *ac, int32_t *frame_idx, int32_t *line_idx, int *h_promoted)
{
    if (!ac) return TSDRGPU_EINVAL;
    int32_t fi = -1, li = -1;
    int rc = tsdrgpu_autocorr_argmax(ac, &fi, &li);
    if (rc) return rc;
    int promoted = 0;
    if (ac->certify && !(ac->res.certified[0] && ac->res.certified[1])) {
        if ((rc = tsdrgpu_autocorr_promote(ac))) return rc;
        if ((rc = tsdrgpu_autocorr_argmax(ac, &fi, &li))) return rc;
        promoted = 1;
    }
    if (frame_idx) *frame_idx = fi;
    if (line_idx) *line_idx = li;
    if (h_promoted) *h_promoted = promoted;
    return TSDRGPU_OK;
}

extern "C" int tsdrgpu_autocorr_last_corr(tsdrgpu_autocorr_t *ac, const float **d_corr, uint32_t *n)
{
    if (!ac || !ac->d_last) return TSDRGPU_ESTATE;
    tsdrgpu_t *g = ac->g;
    if (!ac->last_exact && ac->certify && ac->log_count > 0) {
        // certified mode, epoch still in the float32 form: the last window once more in the reference's arithmetic, so that
        // what a host dumps (dump_autocorrect, frameratedetector.c:64-85) is the reference's bits
        const AcLogRec &r = ac->log[ac->log_count - 1];
        int rc = ac_ensure_exact(ac);
        if (rc) return rc;
        const float *src = r.src + (size_t)(r.nwindows - 1) * (size_t)r.stride * AC_KIND_FLOATS(r.is_iq);
        if ((rc = fftx_correlate(g, ac->st, src, r.is_iq, r.stride, 1, ac->n, ac->d_tw, ac->d_xz, ac->d_xmag))) return rc;
        HIP_TRY(g, hipStreamSynchronize(ac->st));
        if (d_corr) *d_corr = (const float *)ac->d_xz;
        if (n) *n = ac->n;
        return TSDRGPU_OK;
    }
    if (ac->last_exact) {  // already n complex values
        HIP_TRY(g, hipStreamSynchronize(ac->st));
        if (d_corr) *d_corr = (const float *)ac->d_last;
        if (n) *n = ac->n;
        return TSDRGPU_OK;
    }
    if (!ac->d_expand && hipMalloc(&ac->d_expand, sizeof(float2) * (size_t)ac->n) != hipSuccess)
        return tsdr_fail(g, TSDRGPU_ENOMEM, "tsdrgpu_autocorr_last_corr", "buffer");
    const uint32_t nh = ac->n / 2;
    TSDR_LAUNCH(g, PROF_SUPERB_MISC, ac->st, k_ac_expand, (nh + 255) / 256, 256, ac->d_last, ac->d_expand, nh);
    KERNEL_CHECK(g, "k_ac_expand");
    HIP_TRY(g, hipStreamSynchronize(ac->st));
    if (d_corr) *d_corr = (const float *)ac->d_expand;
    if (n) *n = ac->n;
    return TSDRGPU_OK;
}

extern "C" int tsdrgpu_autocorr_set_exact(tsdrgpu_autocorr_t *ac, int on)
{
    if (!ac) return TSDRGPU_EINVAL;
    tsdrgpu_t *g = ac->g;
    HIP_TRY(g, hipStreamSynchronize(ac->st));
    if (on) {
        const int rc = ac_ensure_exact(ac);
        if (rc) return rc;
    }
    ac->exact = on ? 1 : 0;
    ac->d_last = nullptr;  // the last correlation is kept in the mode's own layout
    return TSDRGPU_OK;
}

extern "C" int tsdrgpu_autocorr_set_certify(tsdrgpu_autocorr_t *ac, int mode, size_t retain_bytes)
{
    if (!ac || mode < 0 || mode > 2) return ac ? tsdr_fail(ac->g, TSDRGPU_EINVAL, "tsdrgpu_autocorr_set_certify", "mode must be 0, 1 or 2") : TSDRGPU_EINVAL;
    tsdrgpu_t *g = ac->g;
    HIP_TRY(g, hipStreamSynchronize(ac->st));
    if (ac->log_count || ac->epoch_exact) {  // mid-epoch: what was run so far can no longer be replayed consistently
        const int rc = tsdrgpu_autocorr_reset(ac);
        if (rc) return rc;
    }
    if (ac->ring) { ac->ring->stop(); delete ac->ring; ac->ring = nullptr; }
    ac->ring_cap = ac->ring_count = 0;
    free(ac->log);
    ac->log = nullptr;
    ac->log_cap = ac->log_count = 0;
    ac->certify = 0;
    if (!mode) return TSDRGPU_OK;
    int rc = ac_ensure_exact(ac);  // the table is built on the host (tens of ms at 2^22): now, not at the first promotion
    if (rc) return rc;
    {
        const char *e = getenv("TSDRGPU_AC_CHECK_EVERY");
        ac->check_every = e ? atoi(e) : 16;
        ac->since_check = -1;  // the first plot update of the object is checked
    }
    int cap = 1024;
    if (mode == 1) {
        if (retain_bytes == 0) {
            // default: a quarter of what is free on the device right now, at most 32 GiB (2048 windows of 2^22 samples =
            // 116 s of real-time signal at 100 MS/s), at least 256 MiB — HBM is not the scarce resource on a 288 GB part,
            // and an epoch that outgrows the ring costs one exact replay and runs in the (4x slower) exact form from then on
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = (size_t)4 << 30;
            retain_bytes = free_b / 4;
            if (retain_bytes > ((size_t)32 << 30)) retain_bytes = (size_t)32 << 30;
            if (retain_bytes < ((size_t)256 << 20)) retain_bytes = (size_t)256 << 20;
        }
        const size_t win_bytes = sizeof(float) * (size_t)ac->n;
        size_t w = retain_bytes / win_bytes;
        if (w < 1) w = 1;
        if (w > 65535) w = 65535;
        // segments of about 256 MiB (TSDRGPU_AC_SEGMENT_MB), at least 32 windows (a call's windows lie in one segment),
        // never more than the whole ring: 512 MiB at 100 MS/s, 1 GiB at 200 MS/s
        const char *sm = getenv("TSDRGPU_AC_SEGMENT_MB");
        const size_t seg_target = (sm && atol(sm) > 0) ? (size_t)atol(sm) << 20 : (size_t)256 << 20;
        size_t sw = seg_target / win_bytes;
        if (sw < 32) sw = 32;
        if (sw > w) sw = w;
        AcRing *rg = new (std::nothrow) AcRing();
        if (!rg) return tsdr_fail(g, TSDRGPU_ENOMEM, "tsdrgpu_autocorr_set_certify", "retention ring");
        rg->g = g;
        rg->seg_windows = (int)sw;
        rg->seg_bytes = sw * win_bytes;
        rg->nseg_max = (int)(w / sw);
        if (rg->nseg_max < 1) rg->nseg_max = 1;
        rg->seg.assign((size_t)rg->nseg_max, nullptr);
        if (!rg->alloc_one(ac->st)) {  // segment 0 is there when this call returns
            delete rg;
            return tsdr_fail(g, TSDRGPU_ENOMEM, "tsdrgpu_autocorr_set_certify", "retention ring");
        }
        // two segments beyond the one in use are kept ready (ac_ring_slot): a real-time stream fills a segment in seconds, the
        // thread makes one in 40-150 ms; a host that knows it will need more says so (tsdrgpu_autocorr_retention_reserve).
        // (Asking for the whole ring at once made a one-second session allocate — and free — 32 GiB it never used.)
        rg->want = rg->nseg_max > 1 ? 2 : 1;
        if (rg->nseg_max > 1) rg->th = std::thread([rg] { rg->run(); });
        ac->ring = rg;
        ac->ring_cap = rg->nseg_max * rg->seg_windows;
        if (cap < ac->ring_cap) cap = ac->ring_cap;
    }
    ac->log = (AcLogRec *)malloc(sizeof(AcLogRec) * (size_t)cap);
    if (!ac->log) {
        if (ac->ring) { ac->ring->stop(); delete ac->ring; ac->ring = nullptr; }
        ac->ring_cap = 0;
        return tsdr_fail(g, TSDRGPU_ENOMEM, "tsdrgpu_autocorr_set_certify", "log");
    }
    ac->log_cap = cap;
    ac->certify = mode;
    return TSDRGPU_OK;
}

// asks the allocator for room for `windows` windows of the ring now; waits up to wait_ms for it (0: do not wait)
extern "C" int tsdrgpu_autocorr_retention_reserve(tsdrgpu_autocorr_t *ac, int windows, int wait_ms)
{
    if (!ac || windows < 0 || wait_ms < 0) return TSDRGPU_EINVAL;
    AcRing *rg = ac->ring;
    if (ac->certify != 1 || !rg) return tsdr_fail(ac->g, TSDRGPU_ESTATE, "tsdrgpu_autocorr_retention_reserve", "no retention ring (tsdrgpu_autocorr_set_certify mode 1)");
    int nseg = (windows + rg->seg_windows - 1) / rg->seg_windows;
    if (nseg > rg->nseg_max) nseg = rg->nseg_max;
    rg->ask(nseg);
    for (int waited = 0; waited < wait_ms && rg->ready.load() < nseg; waited += 2) {
        { std::lock_guard<std::mutex> lk(rg->m); if (rg->failed) break; }
        std::this_thread::sleep_for(std::chrono::milliseconds(2));
    }
    return TSDRGPU_OK;
}

extern "C" int tsdrgpu_autocorr_retention(tsdrgpu_autocorr_t *ac, int *ring_windows, int *ring_ready, int *retained_windows, int *epoch_is_exact)
{
    if (!ac) return TSDRGPU_EINVAL;
    const int on = ac->certify == 1 && ac->ring;
    if (ring_windows) *ring_windows = on ? ac->ring_cap : 0;
    if (ring_ready) *ring_ready = on ? ac->ring->ready.load() * ac->ring->seg_windows : 0;
    if (retained_windows) *retained_windows = on ? ac->ring_count : 0;
    if (epoch_is_exact) *epoch_is_exact = (ac->exact || ac->epoch_exact) ? 1 : 0;
    return TSDRGPU_OK;
}

extern "C" int tsdrgpu_autocorr_set_async(tsdrgpu_autocorr_t *ac, int on)
{
    if (!ac) return TSDRGPU_EINVAL;
    tsdrgpu_t *g = ac->g;
    HIP_TRY(g, hipStreamSynchronize(ac->st));
    ac->st = on ? g->bg : g->stream;
    return TSDRGPU_OK;
}

extern "C" int tsdrgpu_autocorr_lane(tsdrgpu_autocorr_t *ac) { return (ac && ac->st == ac->g->bg) ? TSDRGPU_LANE_BACKGROUND : TSDRGPU_LANE_COMPUTE; }

extern "C" int tsdrgpu_autocorr_set_plan(tsdrgpu_autocorr_t *ac, int trips)
{
    if (!ac || (trips != 3 && trips != 5)) return ac ? tsdr_fail(ac->g, TSDRGPU_EINVAL, "tsdrgpu_autocorr_set_plan", "trips must be 3 or 5") : TSDRGPU_EINVAL;
    ac->plan5 = trips == 5 ? 1 : 0;
    return TSDRGPU_OK;
}

// ---------------------------------------------------------------------------
// a13/a14  super-bandwidth stitch — superbandwidth.c:67-152, fft.c:69-93
// ---------------------------------------------------------------------------
// complex_to_abs_diff (superbandwidth.c:67-81): first difference of magnitudes;
// element 0 is seeded with |z0|^2 (kept literally).
__global__ __launch_bounds__(256) void k_abs_diff(const float2 *__restrict__ z, float2 *__restrict__ out, unsigned n)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = make_float2(sb_absdiff(z, i), 0.f);
}

// superb_bestfit's peak search (superbandwidth.c:100-116): first maximum of |.| over n points of each of
// gridDim.y correlations (z + y*stride), in two stages like the plots' argmax: SB_ARG_BLOCKS workgroups per
// correlation, then one wave each; the lowest index wins ties.
#define SB_ARG_BLOCKS 256
__global__ __launch_bounds__(256) void k_argmax_abs_partial(const float2 *__restrict__ z, long long stride, unsigned n, float *__restrict__ pval,
                                                            int *__restrict__ pidx)
{
    const float2 *zb = z + (long long)blockIdx.y * stride;
    float best = -1.f;
    int at = 0x7fffffff;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float2 c = zb[i];
        const float v = sqrtf(c.x * c.x + c.y * c.y);
        if (v > best) { best = v; at = (int)i; }  // i ascending per thread
    }
    __shared__ float sb[4];
    __shared__ int si[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_down(best, o, 64);
        const int oi = __shfl_down(at, o, 64);
        if (ob > best || (ob == best && oi < at)) { best = ob; at = oi; }
    }
    if ((threadIdx.x & 63) == 0) { sb[threadIdx.x >> 6] = best; si[threadIdx.x >> 6] = at; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++)
            if (sb[w] > best || (sb[w] == best && si[w] < at)) { best = sb[w]; at = si[w]; }
        pval[blockIdx.y * SB_ARG_BLOCKS + blockIdx.x] = best;
        pidx[blockIdx.y * SB_ARG_BLOCKS + blockIdx.x] = at;
    }
}

__global__ __launch_bounds__(64) void k_argmax_abs_final(const float *__restrict__ pval, const int *__restrict__ pidx, int *__restrict__ out_floats)
{
    float best = -1.f;
    int at = 0x7fffffff;
    for (int b = threadIdx.x; b < SB_ARG_BLOCKS; b += 64) {
        const float ob = pval[blockIdx.x * SB_ARG_BLOCKS + b];
        const int oi = pidx[blockIdx.x * SB_ARG_BLOCKS + b];
        if (ob > best || (ob == best && oi < at)) { best = ob; at = oi; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_down(best, o, 64);
        const int oi = __shfl_down(at, o, 64);
        if (ob > best || (ob == best && oi < at)) { best = ob; at = oi; }
    }
    if (threadIdx.x == 0) out_floats[blockIdx.x] = 2 * at;  // the reference returns the offset in floats
}

// fft_crosscorrelation's product (fft.c:80-89) for a batch: b[y][i] <- a[i] (x) b[y][i], a shared by the batch
__global__ __launch_bounds__(256) void k_mul_conj_batch(const float2 *__restrict__ a, float2 *__restrict__ b, unsigned n)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float2 *bb = b + (long long)blockIdx.y * n;
    const float2 x = a[i], y = bb[i];
    bb[i] = make_float2(x.x * y.x + x.y * y.y, x.x * y.y - x.y * y.x);
}

// circular left rotation by *off floats (superbandwidth.c:135-137)
__global__ __launch_bounds__(256) void k_rotate(const float *__restrict__ in, float *__restrict__ out, unsigned nfloats,
                                                const int *__restrict__ off)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nfloats) return;
    unsigned s = i + (unsigned)*off;
    if (s >= nfloats) s -= nfloats;
    out[i] = in[s];
}

static uint32_t pow2_floor(uint32_t v)  // fft_getrealsize, fft.c:5-11
{
    uint32_t m = 0;
    while ((v /= 2) != 0) m++;
    return 1u << m;
}

// Scratch of the stitch, kept in the context between calls (one stitch per 4-hop cycle, always the same size).
static int superb_scratch(tsdrgpu_t *g, size_t bytes, char **out)
{
    if (g->superb_ws_bytes < bytes) {
        HIP_TRY(g, hipStreamSynchronize(g->stream));
        if (g->superb_ws) (void)hipFree(g->superb_ws);
        g->superb_ws = nullptr;
        g->superb_ws_bytes = 0;
        if (hipMalloc(&g->superb_ws, bytes) != hipSuccess) return tsdr_fail(g, TSDRGPU_ENOMEM, "tsdrgpu_superb_stitch", "work buffers");
        g->superb_ws_bytes = bytes;
    }
    *out = (char *)g->superb_ws;
    return TSDRGPU_OK;
}

// ---------------------------------------------------------------------------
// The stitch on the three-trip plan (fft4step.h, "The super-bandwidth stitch on the same three trips"): four hops (the
// reference's SUPER_HOPS_TO_MAKE, superbandwidth.c:22) whose length and correlated length are N1 * 4096 points,
// N1 = 16 .. 2048.  Seven launches: columns / rows / columns+argmax / peaks for the alignment, columns / rows / columns
// for the transforms; 3 trips over the 4 bn and 3 over the 4 M points where the pass-per-radix plan below makes 7 x 3
// and 4 x 3 + 4.
// ---------------------------------------------------------------------------
static bool sb3_size_ok(uint32_t n) { return n >= 4096u * 16u && n <= 4096u * 2048u && (n & (n - 1u)) == 0u; }

// the three peaks from the partials of k_sb_cols_argmax: slots (array 0 |re|, array 0 |im|, array 1 |re|, array 1 |im|) of
// T partials each; correlation i (hop i + 1 against hop 0) is slot {2, 0, 3}[i] (k_sb_rows<XCORR>: F(P) = c_1 + i c_3, F(Q) = c_2)
// out_floats[0..3]: the four hops' offsets (hop 0: 0), on the device for the rotation and — h_out, pinned — for the host, so
// that neither a memset in front nor a copy behind is queued
__global__ __launch_bounds__(64) void k_sb_argmax_final(const float *__restrict__ pval, const int *__restrict__ pidx, int T, int *__restrict__ out_floats,
                                                        int *__restrict__ h_out)
{
    const int slot = blockIdx.x == 0 ? 2 : (blockIdx.x == 1 ? 0 : 3);
    float best = -1.f;
    int at = 0x7fffffff;
    for (int b = threadIdx.x; b < T; b += 64) {
        const float ob = pval[slot * T + b];
        const int oi = pidx[slot * T + b];
        if (ob > best || (ob == best && oi < at)) { best = ob; at = oi; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_down(best, o, 64);
        const int oi = __shfl_down(at, o, 64);
        if (ob > best || (ob == best && oi < at)) { best = ob; at = oi; }
    }
    // (nothing compared greater than the start value: all NaN — the reference's scan then keeps index 0, superbandwidth.c:104-113)
    if (threadIdx.x == 0) {
        const int off = (at == 0x7fffffff) ? 0 : 2 * at;  // the reference returns the offset in floats
        out_floats[blockIdx.x + 1] = off;
        h_out[blockIdx.x + 1] = off;
        if (blockIdx.x == 0) { out_floats[0] = 0; h_out[0] = 0; }
    }
}

template <int LOGN1>
static void launch_sb3_align(tsdrgpu_t *g, hipStream_t st, const SbHops &hops, uint32_t bn, float2 *W, float2 *V, float *pval, int *pidx, int *d_off, int *h_off)
{
    typedef ColGeom<LOGN1> G;
    const unsigned T = AC4_ROW / G::C;
    TSDR_LAUNCH(g, PROF_AC_COLS, st, (k_sb_cols<LOGN1, 5>), dim3(T, 4), G::NT, hops, W, bn, (const int *)nullptr);
    TSDR_LAUNCH(g, PROF_AC_ROWS, st, (k_sb_rows<SB_ROWS_XCORR>), dim3(1u << LOGN1), 512, (const float2 *)W, V, bn, 1.0f / (float)bn);
    TSDR_LAUNCH(g, PROF_AC_COLS, st, (k_sb_cols_argmax<LOGN1>), dim3(T, 2), G::NT, (const float2 *)V, bn, pval, pidx);
    TSDR_LAUNCH(g, PROF_ARGMAX, st, k_sb_argmax_final, 3, 64, (const float *)pval, (const int *)pidx, (int)T, d_off, h_off);
}

template <int LOGN1>
static void launch_sb3_stitch(tsdrgpu_t *g, hipStream_t st, const SbHops &hops, uint32_t per, float2 *W, float2 *V, float2 *out, const int *d_off)
{
    typedef ColGeom<LOGN1> G;
    const unsigned T = AC4_ROW / G::C;
    TSDR_LAUNCH(g, PROF_AC_COLS, st, (k_sb_cols<LOGN1, 6>), dim3(T, 4), G::NT, hops, W, per, d_off);
    TSDR_LAUNCH(g, PROF_AC_ROWS, st, (k_sb_rows<SB_ROWS_STITCH>), dim3(1u << LOGN1), 512, (const float2 *)W, V, per, 1.0f / (float)per);
    // the last trip of ONE transform of 4 * per points with rows of 4 * 4096: the autocorrelation's own kernel
    TSDR_LAUNCH(g, PROF_AC_COLS, st, (k_ac_cols<LOGN1, 0, true>), dim3(4u * T, 1), G::NT, (const void *)V, 4ll * per, out, 4u * per, KEEP_ALL);
}

#define SB3_DISPATCH(fn_, n_, ...)                                  \
    do {                                                            \
        int l_ = 0;                                                 \
        while ((4096u << l_) < (n_)) l_++;                          \
        switch (l_) {                                               \
            case 4: fn_<4>(__VA_ARGS__); break;                     \
            case 5: fn_<5>(__VA_ARGS__); break;                     \
            case 6: fn_<6>(__VA_ARGS__); break;                     \
            case 7: fn_<7>(__VA_ARGS__); break;                     \
            case 8: fn_<8>(__VA_ARGS__); break;                     \
            case 9: fn_<9>(__VA_ARGS__); break;                     \
            case 10: fn_<10>(__VA_ARGS__); break;                   \
            default: fn_<11>(__VA_ARGS__); break;                   \
        }                                                           \
    } while (0)

static int superb_stitch3(tsdrgpu_t *g, float *const *d_hops, uint32_t per, uint32_t bn, float *d_out, int32_t *h_offsets)
{
    const uint32_t wn = per > bn ? per : bn;
    const unsigned tmax = AC4_ROW / 8u;  // at most 512 column tiles per array
    const size_t bytes = sizeof(float2) * (size_t)wn * 8 + sizeof(int) * 4 + (sizeof(float) + sizeof(int)) * 4 * (size_t)tmax;
    char *ws = nullptr;
    int rc = superb_scratch(g, bytes + 64, &ws);
    if (rc) return rc;
    float2 *W = (float2 *)ws, *V = W + 4 * (size_t)wn;
    int *d_off = (int *)(V + 4 * (size_t)wn);
    float *pval = (float *)(d_off + 4);
    int *pidx = (int *)(pval + 4 * (size_t)tmax);
    hipStream_t st = g->stream;
    if (!g->superb_h_off && hipHostMalloc((void **)&g->superb_h_off, sizeof(int) * 4, hipHostMallocDefault) != hipSuccess) {
        g->superb_h_off = nullptr;
        return tsdr_fail(g, TSDRGPU_ENOMEM, "tsdrgpu_superb_stitch", "pinned offsets");
    }
    SbHops hops;
    for (int i = 0; i < 4; i++) hops.p[i] = d_hops[i];
    SB3_DISPATCH(launch_sb3_align, bn, g, st, hops, bn, W, V, pval, pidx, d_off, g->superb_h_off);
    KERNEL_CHECK(g, "hop alignment");
    SB3_DISPATCH(launch_sb3_stitch, per, g, st, hops, per, W, V, (float2 *)d_out, (const int *)d_off);
    KERNEL_CHECK(g, "stitch transform");
    HIP_TRY(g, hipStreamSynchronize(st));
    if (h_offsets)
        for (int i = 0; i < 4; i++) h_offsets[i] = g->superb_h_off[i];
    return TSDRGPU_OK;
}

extern "C" int tsdrgpu_superb_set_plan(tsdrgpu_t *g, int trips)
{
    if (!g || (trips != 3 && trips != 0)) return g ? tsdr_fail(g, TSDRGPU_EINVAL, "tsdrgpu_superb_set_plan", "trips must be 3 or 0") : TSDRGPU_EINVAL;
    g->superb_passes = trips == 0 ? 1 : 0;
    return TSDRGPU_OK;
}

extern "C" int tsdrgpu_superb_stitch(tsdrgpu_t *g, float *const *d_hops, int nhops, int gathered, int samples_in_frame,
                                     float *d_out, int32_t *h_offsets, uint32_t *h_total)
{
    if (!g || !d_hops || nhops < 1 || nhops > 64 || gathered < 2 || samples_in_frame < 1 || !d_out)
        return g ? tsdr_fail(g, TSDRGPU_EINVAL, "tsdrgpu_superb_stitch", "bad argument") : TSDRGPU_EINVAL;
    const uint32_t per = pow2_floor((uint32_t)gathered);  // superbandwidth.c:124
    const uint32_t total = (uint32_t)nhops * per;
    const uint32_t nfl = per * 2;
    // superb_bestfit sizes (superbandwidth.c:84-86)
    int bsize = ((int)nfl / samples_in_frame) * samples_in_frame;
    if (bsize < 2) return tsdr_fail(g, TSDRGPU_EINVAL, "tsdrgpu_superb_stitch", "hop shorter than one frame");
    const uint32_t bfl = pow2_floor((uint32_t)bsize);
    const uint32_t bn = bfl / 2;  // complex samples cross-correlated
    const uint32_t nfft = pow2_floor(total);  // fft_perform truncates, fft.c:101-105
    const int nb = nhops - 1;              // correlations against hop 0
    if (nhops == 4 && !g->superb_passes && sb3_size_ok(per) && sb3_size_ok(bn)) {
        const int rc3 = superb_stitch3(g, d_hops, per, bn, d_out, h_offsets);
        if (rc3) return rc3;
        if (h_total) *h_total = total;
        return TSDRGPU_OK;
    }

    // [A bn][FA 2 bn][B nb bn][FB 2 nb bn][R total][S 2 total][BIG 2 nfft] float2, then offsets and argmax partials
    const size_t f2 = (size_t)bn * 3 + (size_t)bn * nb * 3 + (size_t)total * 3 + (size_t)nfft * 2;
    const size_t bytes = f2 * sizeof(float2) + sizeof(int) * (size_t)nhops + (sizeof(float) + sizeof(int)) * (size_t)SB_ARG_BLOCKS * (nb > 0 ? nb : 1);
    char *ws = nullptr;
    int rc = superb_scratch(g, bytes + 64, &ws);
    if (rc) return rc;
    float2 *A = (float2 *)ws, *FA = A + bn, *B = FA + 2 * (size_t)bn, *FB = B + (size_t)bn * nb, *R = FB + 2 * (size_t)bn * nb;
    float2 *S = R + total, *BIG = S + 2 * (size_t)total;
    int *d_off = (int *)(BIG + 2 * (size_t)nfft);
    float *pval = (float *)(d_off + nhops);
    int *pidx = (int *)(pval + (size_t)SB_ARG_BLOCKS * (nb > 0 ? nb : 1));
    hipStream_t st = g->stream;
    HIP_TRY(g, hipMemsetAsync(d_off, 0, sizeof(int) * nhops, st));

    // A transform's first pass reads its input through a mode of its own, so complex_to_abs_diff (superbandwidth.c:67-81)
    // and the rotation (:135-137) are loads of the first pass instead of kernels with a round trip over HBM each; the hop
    // buffers are separate allocations, so that pass runs hop by hop and the remaining passes batched.
    const PassPlan pb = plan_passes(bn), pp_ = plan_passes(per), pf = plan_passes(nfft);
    if (nb > 0) {
        // ONE forward transform of hop 0's abs-diff signal and one batched transform of the others', the products, one
        // batched inverse transform, the peaks
        float2 *fa = run_fft_range(g, d_hops[0], 5, 0, FA, FA + bn, bn, 1, pb.radix, pb.count, 0, pb.count, 1, 0, 0, false, 1.0f / (float)bn, st);
        float2 *fb;
        if (pb.count == 1) {
            for (int i = 1; i < nhops; i++)
                (void)run_fft_range(g, d_hops[i], 5, 0, FB + (size_t)(i - 1) * bn, nullptr, bn, 1, pb.radix, 1, 0, 1, 1, 0, 0, false, 1.0f / (float)bn, st);
            fb = FB;
        } else {
            for (int i = 1; i < nhops; i++)
                (void)run_fft_range(g, d_hops[i], 5, 0, FB + (size_t)(i - 1) * bn, nullptr, bn, 1, pb.radix, pb.count, 0, 1, 1, 0, 0, false, 1.0f, st);
            fb = run_fft_range(g, FB, 0, bn, FB, FB + (size_t)bn * nb, bn, nb, pb.radix, pb.count, 1, pb.count, (unsigned)pb.radix[0], 0, 0, false,
                               1.0f / (float)bn, st);
        }
        KERNEL_CHECK(g, "abs-diff transforms");
        TSDR_LAUNCH(g, PROF_SUPERB_MISC, st, k_mul_conj_batch, dim3((bn + 255) / 256, nb), 256, fa, fb, bn);
        KERNEL_CHECK(g, "k_mul_conj_batch");
        float2 *other = (fb == FB) ? FB + (size_t)bn * nb : FB;
        float2 *xc = run_fft(g, fb, 0, bn, fb, other, bn, nb, 1, false, 1.0f);
        TSDR_LAUNCH(g, PROF_ARGMAX, st, k_argmax_abs_partial, dim3(SB_ARG_BLOCKS, nb), 256, xc, (long long)bn, bn, pval, pidx);
        TSDR_LAUNCH(g, PROF_ARGMAX, st, k_argmax_abs_final, nb, 64, pval, pidx, d_off + 1);
        KERNEL_CHECK(g, "k_argmax_abs");
    }
    // every hop rotated by its offset (hop 0: none) and transformed: the result IS the concatenation of the spectra in hop
    // order, no fftshift (superbandwidth.c:135-144)
    float2 *sp;
    if (pp_.count == 1) {
        for (int i = 0; i < nhops; i++)
            (void)run_fft_range(g, d_hops[i], 6, 0, S + (size_t)i * per, nullptr, per, 1, pp_.radix, 1, 0, 1, 1, 0, 0, false, 1.0f / (float)per, st,
                                KEEP_ALL, d_off + i);
        sp = S;
    } else {
        for (int i = 0; i < nhops; i++)
            (void)run_fft_range(g, d_hops[i], 6, 0, S + (size_t)i * per, nullptr, per, 1, pp_.radix, pp_.count, 0, 1, 1, 0, 0, false, 1.0f, st,
                                KEEP_ALL, d_off + i);
        sp = run_fft_range(g, S, 0, per, S, S + total, per, nhops, pp_.radix, pp_.count, 1, pp_.count, (unsigned)pp_.radix[0], 0, 0, false,
                           1.0f / (float)per, st);
    }
    KERNEL_CHECK(g, "hop transforms");
    // the reference leaves every hop buffer holding its spectrum, and what lies beyond the largest power of two <= total
    // keeps the spectra: copies that nothing below reads, so they go beside the stitch transform on the side stream
    HIP_TRY(g, hipEventRecord(g->fork, st));
    HIP_TRY(g, hipStreamWaitEvent(g->stream2, g->fork, 0));
    for (int i = 0; i < nhops; i++)
        HIP_TRY(g, hipMemcpyAsync(d_hops[i], sp + (size_t)i * per, sizeof(float2) * per, hipMemcpyDeviceToDevice, g->stream2));
    if (total > nfft)
        HIP_TRY(g, hipMemcpyAsync(d_out + 2 * (size_t)nfft, sp + nfft, sizeof(float2) * (size_t)(total - nfft), hipMemcpyDeviceToDevice, g->stream2));
    // the passes ping-pong between two buffers: the caller's d_out takes the place of the one the last pass writes
    {
        float2 *out2 = (float2 *)d_out;
        float2 *a = (pf.count & 1) ? out2 : BIG, *b = (pf.count & 1) ? BIG : out2;
        float2 *res = run_fft_range(g, sp, 0, nfft, a, b, nfft, 1, pf.radix, pf.count, 0, pf.count, 1, 1, 1, false, 1.0f, st);
        KERNEL_CHECK(g, "stitch transform");
        if (res != out2) HIP_TRY(g, hipMemcpyAsync(d_out, res, sizeof(float2) * (size_t)nfft, hipMemcpyDeviceToDevice, st));
    }
    if (h_offsets) HIP_TRY(g, hipMemcpyAsync(h_offsets, d_off, sizeof(int) * nhops, hipMemcpyDeviceToHost, st));
    HIP_TRY(g, hipStreamSynchronize(st));
    HIP_TRY(g, hipStreamSynchronize(g->stream2));
    if (h_total) *h_total = total;
    return TSDRGPU_OK;
}

// ---------------------------------------------------------------------------
// the stitch with ONE HOP PER GPU.  superb_ondataready (superbandwidth.c:121-152) aligns hops 1..3
// to hop 0 (abs-diff, cross-correlation, peak), rotates them, transforms every hop and inverse-transforms the
// concatenated spectra.  All of it but two things is per hop: the reference spectrum of hop 0 (needed by everybody:
// one broadcast of bn complex values) and the final transform (needs everybody's spectrum: one all-gather of
// nhops * per complex values).  The phases below run the kernels of the single-GPU call's pass-per-radix plan
// (tsdrgpu_superb_set_plan(g, 0); also what it takes below 2^16 points per hop) on this rank's hop, so offsets and the
// stitched signal are bit-identical to that plan's; against the three-trip plan the offsets are identical and the signal
// agrees to rounding.  The exchanges are the caller's (tsdrgpu_comm_broadcast_f32
// / _allgather_f32 over RCCL), which is how the two-process tests run them through gloo on one device.
// ---------------------------------------------------------------------------
struct tsdrgpu_superb_shard {
    tsdrgpu_t *g;
    int nhops, my_hop, gathered, samples_in_frame;
    uint32_t per, total, nfl, bn, nfft;
    float2 *ws;      // [A bn][FA 2 bn][B bn][FB 2 bn][R per][S 2 per][ALL total (+ work total)][BIG 2 nfft]
    float2 *A, *FA, *B, *FB, *R, *S, *ALL, *BIG;
    float2 *fa;      // where the reference spectrum sits (FA or FA + bn)
    int *d_off;
    float *pval;
    int *pidx;
    int phase;       // 0 created, 1 reference queued, 2 spectrum queued
};

extern "C" int tsdrgpu_superb_shard_create(tsdrgpu_t *g, tsdrgpu_superb_shard_t **out, int nhops, int my_hop, int gathered, int samples_in_frame)
{
    if (!g || !out || nhops < 1 || nhops > 64 || my_hop < 0 || my_hop >= nhops || gathered < 2 || samples_in_frame < 1)
        return g ? tsdr_fail(g, TSDRGPU_EINVAL, "tsdrgpu_superb_shard_create", "bad argument") : TSDRGPU_EINVAL;
    tsdrgpu_superb_shard_t *sh = (tsdrgpu_superb_shard_t *)calloc(1, sizeof(*sh));
    if (!sh) return TSDRGPU_ENOMEM;
    sh->g = g;
    sh->nhops = nhops;
    sh->my_hop = my_hop;
    sh->gathered = gathered;
    sh->samples_in_frame = samples_in_frame;
    sh->per = pow2_floor((uint32_t)gathered);  // superbandwidth.c:124
    sh->total = (uint32_t)nhops * sh->per;
    sh->nfl = sh->per * 2;
    const int bsize = ((int)sh->nfl / samples_in_frame) * samples_in_frame;  // superb_bestfit sizes, superbandwidth.c:84-86
    if (bsize < 2) {
        free(sh);
        return tsdr_fail(g, TSDRGPU_EINVAL, "tsdrgpu_superb_shard_create", "hop shorter than one frame");
    }
    sh->bn = pow2_floor((uint32_t)bsize) / 2;
    sh->nfft = pow2_floor(sh->total);
    const size_t bn = sh->bn, per = sh->per, total = sh->total, nfft = sh->nfft;
    const size_t f2 = bn * 6 + per * 3 + total + 2 * nfft;
    const size_t bytes = f2 * sizeof(float2) + sizeof(int) * 4 + (sizeof(float) + sizeof(int)) * (size_t)SB_ARG_BLOCKS + 64;
    if (hipMalloc((void **)&sh->ws, bytes) != hipSuccess) {
        free(sh);
        return tsdr_fail(g, TSDRGPU_ENOMEM, "tsdrgpu_superb_shard_create", "work buffers");
    }
    sh->A = sh->ws;
    sh->FA = sh->A + bn;
    sh->B = sh->FA + 2 * bn;
    sh->FB = sh->B + bn;
    sh->R = sh->FB + 2 * bn;
    sh->S = sh->R + per;
    sh->ALL = sh->S + 2 * per;
    sh->BIG = sh->ALL + total;
    sh->d_off = (int *)(sh->BIG + 2 * nfft);
    sh->pval = (float *)(sh->d_off + 4);
    sh->pidx = (int *)(sh->pval + SB_ARG_BLOCKS);
    *out = sh;
    return TSDRGPU_OK;
}

extern "C" void tsdrgpu_superb_shard_destroy(tsdrgpu_superb_shard_t *sh)
{
    if (!sh) return;
    (void)hipStreamSynchronize(sh->g->stream);
    (void)hipFree(sh->ws);
    free(sh);
}

// phase 1.  The rank of hop 0 transforms its abs-diff signal; every rank gets the buffer the caller broadcasts it into.
extern "C" int tsdrgpu_superb_shard_reference(tsdrgpu_superb_shard_t *sh, const float *d_my_hop, float **d_ref, int64_t *n_floats)
{
    if (!sh || !d_my_hop) return TSDRGPU_EINVAL;
    tsdrgpu_t *g = sh->g;
    hipStream_t st = g->stream;
    // the reference spectrum always ends up in the first half of FA (hop 0's rank copies it there if the transform's
    // ping-pong ended in the second), so that every rank exchanges the same buffer
    if (sh->my_hop == 0) {
        TSDR_LAUNCH(g, PROF_SUPERB_MISC, st, k_abs_diff, (sh->bn + 255) / 256, 256, (const float2 *)d_my_hop, sh->A, sh->bn);
        KERNEL_CHECK(g, "k_abs_diff");
        float2 *fa = run_fft(g, sh->A, 0, sh->bn, sh->FA, sh->FA + sh->bn, sh->bn, 1, 0, false, 1.0f / (float)sh->bn);
        if (fa != sh->FA) HIP_TRY(g, hipMemcpyAsync(sh->FA, fa, sizeof(float2) * sh->bn, hipMemcpyDeviceToDevice, st));
    }
    sh->fa = sh->FA;
    sh->phase = 1;
    if (d_ref) *d_ref = (float *)sh->FA;
    if (n_floats) *n_floats = 2 * (int64_t)sh->bn;
    return TSDRGPU_OK;
}

// phase 2 (after the broadcast).  This rank's offset against hop 0, its rotation, its spectrum — written into slot
// my_hop of the gather buffer, which the caller all-gathers in place.
extern "C" int tsdrgpu_superb_shard_spectrum(tsdrgpu_superb_shard_t *sh, float *d_my_hop, float **d_spectra, int64_t *n_floats_per_hop,
                                             int32_t *h_my_offset)
{
    if (!sh || !d_my_hop) return TSDRGPU_EINVAL;
    tsdrgpu_t *g = sh->g;
    if (sh->phase != 1) return tsdr_fail(g, TSDRGPU_ESTATE, "tsdrgpu_superb_shard_spectrum", "call tsdrgpu_superb_shard_reference first");
    hipStream_t st = g->stream;
    const uint32_t bn = sh->bn, per = sh->per;
    HIP_TRY(g, hipMemsetAsync(sh->d_off, 0, sizeof(int) * 4, st));
    if (sh->my_hop != 0) {
        TSDR_LAUNCH(g, PROF_SUPERB_MISC, st, k_abs_diff, (bn + 255) / 256, 256, (const float2 *)d_my_hop, sh->B, bn);
        float2 *fb = run_fft(g, sh->B, 0, bn, sh->FB, sh->FB + bn, bn, 1, 0, false, 1.0f / (float)bn);
        TSDR_LAUNCH(g, PROF_SUPERB_MISC, st, k_mul_conj_batch, dim3((bn + 255) / 256, 1), 256, sh->fa, fb, bn);
        float2 *other = (fb == sh->FB) ? sh->FB + bn : sh->FB;
        float2 *xc = run_fft(g, fb, 0, bn, fb, other, bn, 1, 1, false, 1.0f);
        TSDR_LAUNCH(g, PROF_ARGMAX, st, k_argmax_abs_partial, dim3(SB_ARG_BLOCKS, 1), 256, xc, (long long)bn, bn, sh->pval, sh->pidx);
        TSDR_LAUNCH(g, PROF_ARGMAX, st, k_argmax_abs_final, 1, 64, sh->pval, sh->pidx, sh->d_off);
        KERNEL_CHECK(g, "hop alignment");
    }
    TSDR_LAUNCH(g, PROF_SUPERB_MISC, st, k_rotate, (sh->nfl + 255) / 256, 256, d_my_hop, (float *)sh->R, sh->nfl, sh->d_off);
    float2 *sp = run_fft(g, sh->R, 0, per, sh->S, sh->S + per, per, 1, 0, false, 1.0f / (float)per);
    KERNEL_CHECK(g, "hop transform");
    HIP_TRY(g, hipMemcpyAsync(sh->ALL + (size_t)sh->my_hop * per, sp, sizeof(float2) * per, hipMemcpyDeviceToDevice, st));
    HIP_TRY(g, hipMemcpyAsync(d_my_hop, sp, sizeof(float2) * per, hipMemcpyDeviceToDevice, st));  // the reference leaves the spectrum in the hop buffer
    if (h_my_offset) {
        HIP_TRY(g, hipMemcpyAsync(h_my_offset, sh->d_off, sizeof(int), hipMemcpyDeviceToHost, st));
        HIP_TRY(g, hipStreamSynchronize(st));
    }
    sh->phase = 2;
    if (d_spectra) *d_spectra = (float *)sh->ALL;
    if (n_floats_per_hop) *n_floats_per_hop = 2 * (int64_t)per;
    return TSDRGPU_OK;
}

// phase 3 (after the all-gather): the inverse transform of the concatenated spectra -> d_out (nhops * 2 * per floats)
extern "C" int tsdrgpu_superb_shard_finish(tsdrgpu_superb_shard_t *sh, float *d_out, uint32_t *h_total)
{
    if (!sh || !d_out) return TSDRGPU_EINVAL;
    tsdrgpu_t *g = sh->g;
    if (sh->phase != 2) return tsdr_fail(g, TSDRGPU_ESTATE, "tsdrgpu_superb_shard_finish", "call tsdrgpu_superb_shard_spectrum first");
    hipStream_t st = g->stream;
    const uint32_t total = sh->total, nfft = sh->nfft;
    if (total > nfft)  // what lies beyond the largest power of two keeps the spectra (fft_perform truncates, fft.c:101-105)
        HIP_TRY(g, hipMemcpyAsync(d_out + 2 * (size_t)nfft, sh->ALL + nfft, sizeof(float2) * (size_t)(total - nfft), hipMemcpyDeviceToDevice, st));
    float2 *res = run_fft(g, sh->ALL, 0, nfft, sh->BIG, sh->BIG + nfft, nfft, 1, 1, false, 1.0f);
    KERNEL_CHECK(g, "stitch transform");
    HIP_TRY(g, hipMemcpyAsync(d_out, res, sizeof(float2) * (size_t)nfft, hipMemcpyDeviceToDevice, st));
    HIP_TRY(g, hipStreamSynchronize(st));
    sh->phase = 0;
    if (h_total) *h_total = total;
    return TSDRGPU_OK;
}
