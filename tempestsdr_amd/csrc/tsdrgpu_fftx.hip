// tsdrgpu_fftx.hip — the reference's FFT, bit for bit (fft_perform, TempestSDR/src/fft.c:96-176), and the
// autocorrelation built on it (fft.c:49-64, frameratedetector.c:34-62) as an optional EXACT mode of
// tsdrgpu_autocorr_* (tsdrgpu_autocorr_set_exact).
//
// The default path (tsdrgpu_fft.hip) is a different algorithm — Stockham radix-128 passes on the real
// window packed as N/2 complex points — and agrees with the reference to ~1e-6 of the plot maximum.
// Where a plot holds exact mathematical ties (a circular autocorrelation is symmetric, R[j] == R[N-j],
// and the 8 MS/s frame-lag window holds both) the reference's winner is its own rounding noise, so only
// the same arithmetic can give the same answer.  That arithmetic is:
//   * in-place radix-2 decimation in time on N complex points (the real window with zero imaginary
//     parts), bit-reversal first;
//   * per stage s the twiddle of butterfly q is NOT exp(-2 pi i q / 2^(s+1)) but the q-fold f64 product
//     u <- u * w_s started from 1, and w_s itself comes from w_0 = -1 by the half-angle formulas with
//     sqrt — reproduced here on the host, once per transform size, into a table of N-1 doubles pairs;
//   * butterflies in f64 on f32 data, every stage's results rounded to f32 (fft.c stores floats);
//   * forward transform divided by (float)N in f32; magnitude sqrtf(re*re + im*im) in f32; inverse
//     transform with the conjugate twiddles (exactly the negated imaginary parts), unscaled;
//   * running mean of sqrt(re^2 + im^2) in f64 over the lag windows.
// Compiled with -ffp-contract=off; f32 division and sqrt correctly rounded.  Three trips over memory per
// transform (7 + 7 + 8 stages at N = 2^22) through 4096-point LDS tiles whose values are kept as f32
// between stages, which is precisely the reference's storage rounding.  About 7x the default path's time.
#include "tsdrgpu_internal.h"
#include <math.h>
#include <stdlib.h>

struct FftxTrip {
    int s0;  // first stage of the trip
    int L;   // stages in the trip: tile rows = 2^L elements 2^s0 apart
};

__device__ __forceinline__ unsigned fftx_rev(unsigned v, int bits) { return bits ? (__brev(v) >> (32 - bits)) : 0u; }

// One trip: stages s0 .. s0+L-1 on tiles of R = 2^L rows x C columns (R*C <= 4096).
//  s0 == 0: the rows of column r are the 2^L elements of bit-reversed block B = rev(r), i.e. the source
//           elements rev_L(t)*S + r, S = n >> L (coalesced in r); results go to z[B*R + t].
//  s0 >  0: in place on z, element (row, c) of tile (group, c0) is z[group*2^(s0+L) + row*2^s0 + c0 + c].
// src_mode (s0 == 0 only): 0 real floats, 1 interleaved IQ demodulated on the fly (TSDRLibrary.c:244-262), 3 real floats
//           holding re*re + im*im as am_demod forms it (two rounded products, their rounded sum): the root is taken here — what
//           trip 1 of the float32 transform leaves in the retention ring (k_ac_cols_retain, fft4step.h),
//           2 complex.
// epilogue (last trip of a forward transform): 1 = divide by nf, magnitude -> mag[] (real), fft.c:167-175,34-45;
//           2 = divide by nf only.
__global__ __launch_bounds__(256) void k_fftx_trip(const float *__restrict__ src, int src_mode, long long src_stride,
                                                   float2 *__restrict__ z, float *__restrict__ mag, unsigned n, int m, int s0, int L,
                                                   int C, const double2 *__restrict__ tw, int inverse, int epilogue, float nf)
{
    __shared__ float2 t[4096 + 256];  // rows padded by one element: the block-wise write-back walks down a column
    const unsigned R = 1u << L;
    const unsigned Cp = (unsigned)C + 1u;
    const unsigned tid = threadIdx.x;
    const unsigned tile = blockIdx.x;
    float2 *zb = z + (long long)blockIdx.y * n;
    const unsigned total = R * (unsigned)C;  // elements of the tile
    unsigned colbase = 0;                    // global column of tile column 0, for the twiddle index
    unsigned S = 0, r0 = 0, group = 0;
    if (s0 == 0) {
        S = n >> L;
        r0 = tile * (unsigned)C;
        const float *sb = src + (long long)blockIdx.y * src_stride * ((src_mode == 0 || src_mode == 3) ? 1 : 2);
        for (unsigned e = tid; e < total; e += 256) {
            const unsigned row = e / (unsigned)C, c = e % (unsigned)C;
            const unsigned long long at = (unsigned long long)fftx_rev(row, L) * S + r0 + c;
            if (src_mode == 2) {  // complex input (fft_perform on a caller's buffer)
                t[row * Cp + c] = ((const float2 *)sb)[at];
                continue;
            }
            float v;
            if (src_mode == 1) {
                const float2 iq = ((const float2 *)sb)[at];
                v = sqrtf(iq.x * iq.x + iq.y * iq.y);
            } else if (src_mode == 3) {
                v = sqrtf(sb[at]);
            } else {
                v = sb[at];
            }
            t[row * Cp + c] = make_float2(v, 0.f);  // real_to_complex, fft.c:14-22
        }
    } else {
        const unsigned per_group = (1u << s0) / (unsigned)C;
        group = tile / per_group;
        colbase = (tile % per_group) * (unsigned)C;
        const unsigned long long gbase = (unsigned long long)group << (s0 + L);
        for (unsigned e = tid; e < total; e += 256) {
            const unsigned row = e / (unsigned)C, c = e % (unsigned)C;
            t[row * Cp + c] = zb[gbase + ((unsigned long long)row << s0) + colbase + c];
        }
    }
    __syncthreads();
    const unsigned nbf = total >> 1;  // butterflies per stage in the tile
    for (int ls = 0; ls < L; ls++) {
        const int s = s0 + ls;
        const unsigned halfrows = 1u << ls;
        const double2 *ts = tw + ((1ull << s) - 1ull);  // this stage's u[q], q < 2^s
        for (unsigned b = tid; b < nbf; b += 256) {
            const unsigned br = b / (unsigned)C, c = b % (unsigned)C;
            const unsigned ra = ((br >> ls) << (ls + 1)) | (br & (halfrows - 1u));
            const unsigned rb = ra + halfrows;
            const unsigned long long q = ((unsigned long long)(ra & (halfrows - 1u)) << s0) + (s0 ? colbase + c : 0u);
            double2 u = ts[q];
            if (inverse) u.y = -u.y;  // the inverse recurrence yields exactly the conjugates
            const float2 za = t[ra * Cp + c], zq = t[rb * Cp + c];
            const double tr = u.x * (double)zq.x - u.y * (double)zq.y;  // fft.c:153-154
            const double ti = u.x * (double)zq.y + u.y * (double)zq.x;
            t[rb * Cp + c] = make_float2((float)((double)za.x - tr), (float)((double)za.y - ti));
            t[ra * Cp + c] = make_float2((float)((double)za.x + tr), (float)((double)za.y + ti));
        }
        __syncthreads();
    }
    if (s0 == 0) {
        // column c is block B = rev_{m-L}(r0 + c): contiguous R results per block
        for (unsigned e = tid; e < total; e += 256) {
            const unsigned c = e / R, row = e % R;
            const unsigned long long B = fftx_rev(r0 + c, m - L);
            float2 v = t[row * Cp + c];
            const unsigned long long at = B * R + row;
            if (epilogue) {
                v.x = v.x / nf;
                v.y = v.y / nf;
            }
            if (epilogue == 1) mag[(long long)blockIdx.y * n + at] = sqrtf(v.x * v.x + v.y * v.y);
            else zb[at] = v;
        }
    } else {
        const unsigned long long gbase = (unsigned long long)group << (s0 + L);
        for (unsigned e = tid; e < total; e += 256) {
            const unsigned row = e / (unsigned)C, c = e % (unsigned)C;
            float2 v = t[row * Cp + c];
            const unsigned long long at = gbase + ((unsigned long long)row << s0) + colbase + c;
            if (epilogue) {
                v.x = v.x / nf;
                v.y = v.y / nf;
            }
            if (epilogue == 1) mag[(long long)blockIdx.y * n + at] = sqrtf(v.x * v.x + v.y * v.y);
            else zb[at] = v;
        }
    }
}

// ---------------------------------------------------------------------------
// The same trips, specialised (round 3).  k_fftx_trip above is generic — runtime tile shape, one radix-2 stage per LDS
// round trip, a twiddle load per butterfly — and ran at 0.06 of the HBM roofline.  k_fftx_fast keeps the arithmetic
// bit for bit (the same f64 butterfly on f32 values, every stage's results rounded to f32) and changes how it is fed:
//   * tile shape at compile time (L stages x 4096 >> L columns): index math is shifts and masks;
//   * up to three stages per LDS round trip: a thread holds the 8 (4, 2) points of one radix-8 (4, 2) group, rounds
//     them to f32 between the stages exactly as the LDS store did, and needs 7 (3, 1) twiddles for its 12 (4, 1)
//     butterflies;
//   * the windows of a batch that share a tile position run back to back on the same XCD (block -> (tile, window)
//     map below), so the late stages' twiddles — 64 MB of f64 pairs at N = 2^22, every entry used by one butterfly per
//     window — come from that XCD's L2 for all windows but the first;
//   * the last trip of the inverse transform stores only what the detector reads (the two lag windows and lag 0),
//     except for the window kept whole for tsdrgpu_autocorr_last_corr.
// ---------------------------------------------------------------------------
#ifndef FFTX_LOAD_UNROLL
#define FFTX_LOAD_UNROLL 16
#endif
#ifndef FFTX_TW_FIRST
#define FFTX_TW_FIRST 0
#endif
struct FftxKeep {
    int on;       // 0: store everything
    int full_w;   // window of the batch stored whole (-1: none)
    unsigned lo0, hi0, lo1, hi1;  // lag ranges kept (point 0 always)
};

__device__ __forceinline__ void fftx_bfly(float2 &a, float2 &b, double2 u)
{
    const double tr = u.x * (double)b.x - u.y * (double)b.y;  // fft.c:153-154
    const double ti = u.x * (double)b.y + u.y * (double)b.x;
    const float2 na = make_float2((float)((double)a.x + tr), (float)((double)a.y + ti));
    b = make_float2((float)((double)a.x - tr), (float)((double)a.y - ti));
    a = na;
}

// K consecutive stages, local stages a .. a+K-1 of the tile, on groups of 2^K rows {base + (j << a)} of column c.
// U groups side by side (their loads in flight together, their butterflies interleaved): the rows of group u
template <int K, bool FIRST, int U, bool INV>
__device__ __forceinline__ void fftx_groups(float2 *__restrict__ t, unsigned Cp, const unsigned (&base)[U], const unsigned (&c)[U], int a, int s0,
                                            unsigned colbase, const double2 *__restrict__ tw)
{
    constexpr int NP = 1 << K;
    // The chunk's twiddles (2^K - 1 per group), w[u][(1 << st) - 1 + jl] = stage st, butterfly jl.  Left to itself the scheduler
    // sinks every request next to its use (K waits per chunk); FFTX_TW_FIRST=1 pins all of them above the arithmetic (one wait
    // per chunk, +20 VGPRs).  Measured: 123.0 vs 123.7 us per 2^22 window — the trips are not waiting for their twiddles — so
    // the default is the scheduler's order.
    double2 w[U][NP - 1];
#pragma unroll
    for (int st = 0; st < K; st++) {
        const int s = s0 + a + st;
        const unsigned sbase = (1u << s) - 1u;  // this stage's u[q], q < 2^s, start here (the whole table has n - 1 < 2^32 entries)
#pragma unroll
        for (int u = 0; u < U; u++) {
            const unsigned low = base[u] & ((1u << a) - 1u);
#pragma unroll
            for (int jl = 0; jl < (1 << st); jl++) {
                const unsigned rowlow = low + ((unsigned)jl << a);
                const unsigned q = sbase + (FIRST ? rowlow : ((rowlow << s0) + colbase + c[u]));
                w[u][(1 << st) - 1 + jl] = tw[q];
            }
        }
    }
    float2 v[U][NP];
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
        for (int j = 0; j < NP; j++) v[u][j] = t[(base[u] + ((unsigned)j << a)) * Cp + c[u]];
#if FFTX_TW_FIRST
    __builtin_amdgcn_sched_barrier(0);  // keep the requests above the arithmetic (the scheduler would sink each next to its use)
#endif
#pragma unroll
    for (int st = 0; st < K; st++) {
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int jl = 0; jl < (1 << st); jl++) {
                double2 wq = w[u][(1 << st) - 1 + jl];
                if (INV) wq.y = -wq.y;  // the inverse recurrence yields exactly the conjugates
#pragma unroll
                for (int jh = 0; jh < (NP >> (st + 1)); jh++) {
                    const int j0 = (jh << (st + 1)) | jl;
                    fftx_bfly(v[u][j0], v[u][j0 | (1 << st)], wq);
                }
            }
    }
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
        for (int j = 0; j < NP; j++) t[(base[u] + ((unsigned)j << a)) * Cp + c[u]] = v[u][j];
}

// all groups of one chunk of K stages starting at local stage a: 4096 >> K groups over 256 threads, U at a time
template <int K, int L, bool FIRST, int U, bool INV>
__device__ __forceinline__ void fftx_chunk(float2 *__restrict__ t, int a, int s0, unsigned colbase, const double2 *__restrict__ tw)
{
    constexpr unsigned C = 4096u >> L, Cp = C + 1u;
    constexpr unsigned NG = 4096u >> K;
    static_assert(NG % (256u * U) == 0, "groups per pass");
#pragma unroll 1
    for (unsigned g0 = threadIdx.x; g0 < NG; g0 += 256u * U) {
        unsigned base[U], c[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const unsigned gid = g0 + 256u * (unsigned)u;
            c[u] = gid & (C - 1u);
            const unsigned gb = gid / C;
            base[u] = ((gb >> a) << (a + K)) | (gb & ((1u << a) - 1u));
        }
        fftx_groups<K, FIRST, U, INV>(t, Cp, base, c, a, s0, colbase, tw);
    }
    __syncthreads();
}

// the L stages of a tile as chunks: each chunk is one round trip of the tile through LDS
#ifndef FFTX_K4
#define FFTX_K4 1
#endif
template <int L, bool FIRST, int U, bool INV>
__device__ __forceinline__ void fftx_stages(float2 *__restrict__ t, int s0, unsigned colbase, const double2 *__restrict__ tw)
{
    if constexpr (FFTX_K4 && L >= 7) {
        // radix-16 groups (one per thread): two round trips instead of three, the index arithmetic once per 32 butterflies
        fftx_chunk<4, L, FIRST, 1, INV>(t, 0, s0, colbase, tw);
        fftx_chunk<L - 4, L, FIRST, 1, INV>(t, 4, s0, colbase, tw);
        return;
    }
    if (L >= 3) fftx_chunk<3, L, FIRST, U, INV>(t, 0, s0, colbase, tw);
    if (L >= 6) fftx_chunk<3, L, FIRST, U, INV>(t, 3, s0, colbase, tw);
    if (L == 5) fftx_chunk<2, L, FIRST, U, INV>(t, 3, s0, colbase, tw);
    if (L == 7) fftx_chunk<1, L, FIRST, 2 * U, INV>(t, 6, s0, colbase, tw);
    if (L == 8) fftx_chunk<2, L, FIRST, U, INV>(t, 6, s0, colbase, tw);
}

template <int L, bool FIRST, int U, bool INV>
__global__ __launch_bounds__(256) void k_fftx_fast(const float *__restrict__ src, int src_mode, long long src_stride, float2 *__restrict__ z,
                                                   float *__restrict__ mag, unsigned n, int m, int s0, int batch,
                                                   const double2 *__restrict__ tw, int epilogue, float inv_nf, FftxKeep keep)
{
    constexpr unsigned R = 1u << L, C = 4096u >> L, Cp = C + 1u;
    constexpr int LOGC = 12 - L;
    __shared__ float2 t[R * Cp];
    const unsigned tid = threadIdx.x;
    // block -> (tile, window): the `batch` windows of a tile follow each other on one XCD (blocks are dealt round-robin
    // to the 8 XCDs), tiles advance in octets; the tile count is a multiple of 8 (n >= 2^15)
    const unsigned per_oct = 8u * (unsigned)batch;
    const unsigned oct = blockIdx.x / per_oct, rem = blockIdx.x % per_oct;
    const unsigned w = rem >> 3, tile = oct * 8u + (rem & 7u);
    float2 *zb = z + (long long)w * n;
    unsigned colbase = 0, r0 = 0;
    unsigned long long gbase = 0;
    if (FIRST) {
        // the rows of column r are the 2^L elements of bit-reversed block B = rev(r): source elements rev_L(row)*S + r
        const unsigned S = n >> L;
        r0 = tile * C;
        const float *sb = src + (long long)w * src_stride * ((src_mode == 0 || src_mode == 3) ? 1 : 2);
        // all 16 gathers of a thread are requested before the first is used: the trip is bound by their latency (SQ counters:
        // 68 % of a wave's cycles waiting, the VALU 12 % busy)
#pragma unroll FFTX_LOAD_UNROLL
        for (unsigned e = tid; e < 4096u; e += 256u) {
            const unsigned row = e >> LOGC, c = e & (C - 1u);
            const unsigned long long at = (unsigned long long)fftx_rev(row, L) * S + r0 + c;
            float2 v;
            if (src_mode == 2) v = ((const float2 *)sb)[at];
            else if (src_mode == 1) {
                const float2 iq = ((const float2 *)sb)[at];
                v = make_float2(sqrtf(iq.x * iq.x + iq.y * iq.y), 0.f);  // am_demod, TSDRLibrary.c:244-262
            } else if (src_mode == 3) v = make_float2(sqrtf(sb[at]), 0.f);  // the root of am_demod's retained sum of squares
            else v = make_float2(sb[at], 0.f);  // real_to_complex, fft.c:14-22
            t[row * Cp + c] = v;
        }
    } else {
        const unsigned per_group = (1u << s0) >> LOGC;  // tiles across one group's 2^s0 columns
        const unsigned group = tile / per_group;
        colbase = (tile % per_group) << LOGC;
        gbase = ((unsigned long long)group << (s0 + L)) + colbase;
#pragma unroll FFTX_LOAD_UNROLL
        for (unsigned e = tid; e < 4096u; e += 256u) {
            const unsigned row = e >> LOGC, c = e & (C - 1u);
            t[row * Cp + c] = zb[gbase + ((unsigned long long)row << s0) + c];
        }
    }
    __syncthreads();
    fftx_stages<L, FIRST, U, INV>(t, s0, colbase, tw);
    const bool filter = keep.on && (int)w != keep.full_w;
    if (FIRST) {
        // column c is block B = rev_{m-L}(r0 + c): R contiguous results per block
#pragma unroll 4
        for (unsigned e = tid; e < 4096u; e += 256u) {
            const unsigned c = e >> L, row = e & (R - 1u);
            const unsigned long long B = fftx_rev(r0 + c, m - L);
            float2 v = t[row * Cp + c];
            const unsigned long long at = B * R + row;
            if (epilogue == 1 || epilogue == 2) {  // fft.c:167-175 divides by (float)n, a power of two: the same bits as
                v.x = v.x * inv_nf;                // this multiplication by its exact reciprocal (also where the quotient
                v.y = v.y * inv_nf;                // is subnormal: both round the same exact value)
            }
            if (epilogue == 1) mag[(long long)w * n + at] = sqrtf(v.x * v.x + v.y * v.y);
            else if (!filter || at == 0ull || (at >= keep.lo0 && at < keep.hi0) || (at >= keep.lo1 && at < keep.hi1)) zb[at] = v;
        }
    } else {
#pragma unroll 4
        for (unsigned e = tid; e < 4096u; e += 256u) {
            const unsigned row = e >> LOGC, c = e & (C - 1u);
            float2 v = t[row * Cp + c];
            const unsigned long long at = gbase + ((unsigned long long)row << s0) + c;
            if (epilogue == 1 || epilogue == 2) {
                v.x = v.x * inv_nf;
                v.y = v.y * inv_nf;
            }
            if (epilogue == 1) mag[(long long)w * n + at] = sqrtf(v.x * v.x + v.y * v.y);
            else if (!filter || at == 0ull || (at >= keep.lo0 && at < keep.hi0) || (at >= keep.lo1 && at < keep.hi1)) zb[at] = v;
        }
    }
}

// The autocorrelation's MIDDLE trip: the forward transform's last L stages, fft.c:167-175's division, the magnitude
// (fft.c:49-64) and the inverse transform's first L stages on one tile, without the round trip of the magnitudes over HBM
// and without two of the six kernel ramps.  It works because the two trips want the SAME elements: the forward's last trip
// (s0 = m - L) holds rows {colbase + c + row * 2^(m-L)}, and the inverse's first trip gathers, for its row r', the element
// rev_L(r') * 2^(m-L) + r0 + c — the rows of the tile in bit-reversed order.  So between the two halves the tile is permuted
// in LDS (new row r' = |old row rev_L(r')|), and the inverse's stages then run with the FIRST trip's twiddle indices and its
// store (R contiguous results per column).  The inverse therefore takes the forward's plan backwards (L_last first); the
// stages and the f32 roundings between them are the same sequence whatever the grouping, so the bits do not change.
// zin and zout are different buffers: a tile's results land in other tiles' inputs.
template <int L, int U>
__global__ __launch_bounds__(256, 4) void k_fftx_mid(const float2 *__restrict__ zin, float2 *__restrict__ zout, unsigned n, int m, int batch,
                                                  const double2 *__restrict__ tw, float inv_nf)
{
    constexpr unsigned R = 1u << L, C = 4096u >> L, Cp = C + 1u;
    constexpr int LOGC = 12 - L;
    __shared__ float2 t[R * Cp];
    const unsigned tid = threadIdx.x;
    const unsigned per_oct = 8u * (unsigned)batch;
    const unsigned oct = blockIdx.x / per_oct, rem = blockIdx.x % per_oct;
    const unsigned w = rem >> 3, tile = oct * 8u + (rem & 7u);
    const float2 *zi = zin + (long long)w * n;
    float2 *zo = zout + (long long)w * n;
    const int s0 = m - L;                 // the forward's last trip: one group, 2^s0 columns
    const unsigned colbase = tile << LOGC;
#pragma unroll FFTX_LOAD_UNROLL
    for (unsigned e = tid; e < 4096u; e += 256u) {
        const unsigned row = e >> LOGC, c = e & (C - 1u);
        t[row * Cp + c] = zi[(unsigned long long)colbase + ((unsigned long long)row << s0) + c];
    }
    __syncthreads();
    fftx_stages<L, false, U, false>(t, s0, colbase, tw);
    // (the chunk ends with a barrier)  division, magnitude, rows into bit-reversed order
    float mg[16];
#pragma unroll
    for (unsigned k = 0; k < 16u; k++) {
        const unsigned e = tid + 256u * k;
        const unsigned row = e >> LOGC, c = e & (C - 1u);
        float2 v = t[fftx_rev(row, L) * Cp + c];
        v.x = v.x * inv_nf;  // fft.c:167-175 (see k_fftx_fast)
        v.y = v.y * inv_nf;
        mg[k] = sqrtf(v.x * v.x + v.y * v.y);
    }
    __syncthreads();
#pragma unroll
    for (unsigned k = 0; k < 16u; k++) {
        const unsigned e = tid + 256u * k;
        const unsigned row = e >> LOGC, c = e & (C - 1u);
        t[row * Cp + c] = make_float2(mg[k], 0.f);  // real_to_complex, fft.c:14-22
    }
    __syncthreads();
    fftx_stages<L, true, U, true>(t, 0, 0u, tw);
    // the inverse's first-trip store: column c is block B = rev_{m-L}(r0 + c), R contiguous results
#pragma unroll 4
    for (unsigned e = tid; e < 4096u; e += 256u) {
        const unsigned c = e >> L, row = e & (R - 1u);
        const unsigned long long B = fftx_rev(colbase + c, m - L);
        zo[B * R + row] = t[row * Cp + c];
    }
}

// accummulate (frameratedetector.c:34-62) on the complex correlation, in window order.  Entry frame_len + line_len
// of `plots` accumulates lag 0 the same way (the scale of the argmax certificate, tsdrgpu_autocorr_certificate).
__global__ __launch_bounds__(256) void k_fftx_accumulate(const float2 *__restrict__ corr, unsigned n, int nwindows, int frame_lo,
                                                         int frame_len, int line_lo, int line_len, double *__restrict__ plots,
                                                         unsigned long long calls_before, int mode)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > frame_len + line_len) return;
    const int lag = (i < frame_len) ? (frame_lo + i) : (i < frame_len + line_len ? line_lo + (i - frame_len) : 0);
    double acc = plots[i];
    for (int w = 0; w < nwindows; w++) {
        const float2 v = corr[(long long)w * n + lag];
        const double re = v.x, im = v.y;
        const double now = sqrt(re * re + im * im);
        if (mode == 0) {
            const unsigned long long calls = calls_before + w + 1;
            acc = (acc * (double)(calls - 1) + now) / (double)calls;  // calls == 1: (0*0 + now)/1 == now
        } else {
            acc += now;
        }
    }
    plots[i] = acc;
}

// The first n samples of `cnt` capture windows as the reference's detector sees them — am_demod
// (TSDRLibrary.c:244-262) of interleaved IQ, or a copy of magnitudes — into a contiguous ring (n floats per
// window): what tsdrgpu_autocorr_set_certify(1) retains so that an epoch can be replayed exactly.
__global__ __launch_bounds__(256) void k_fftx_retain(const float *__restrict__ src, int is_iq, long long stride, unsigned n,
                                                     float *__restrict__ dst)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long w = blockIdx.y;
    float v;
    if (is_iq) {
        const float2 iq = ((const float2 *)src)[w * stride + i];
        v = sqrtf(iq.x * iq.x + iq.y * iq.y);
    } else {
        v = src[w * stride + i];
    }
    dst[w * (long long)n + i] = v;
}

int fftx_retain(tsdrgpu_t *g, hipStream_t st, const float *src, int is_iq, long long stride, int cnt, uint32_t n, float *dst)
{
    TSDR_LAUNCH(g, PROF_DEMOD, st, k_fftx_retain, dim3((n + 255) / 256, cnt), 256, src, is_iq, stride, n, dst);
    if (hipGetLastError() != hipSuccess) return tsdr_fail(g, TSDRGPU_EHIP, "exact FFT", "retain");
    return TSDRGPU_OK;
}

// u[q] of every stage, exactly as fft.c:132-165 computes them while it runs
int fftx_build_table(tsdrgpu_t *g, uint32_t n, double2 **d_tw)
{
    int m = 0;
    while ((1u << m) < n) m++;
    const size_t count = n > 1 ? (size_t)n - 1 : 1;
    double2 *h = (double2 *)malloc(sizeof(double2) * count);
    if (!h) return tsdr_fail(g, TSDRGPU_ENOMEM, "exact FFT", "twiddle table (host)");
    double wr = -1.0, wi = 0.0;
    size_t at = 0;
    for (int s = 0; s < m; s++) {
        const size_t half = (size_t)1 << s;
        double ur = 1.0, ui = 0.0;
        for (size_t q = 0; q < half; q++) {
            h[at + q].x = ur;
            h[at + q].y = ui;
            const double nr = ur * wr - ui * wi;
            ui = ur * wi + ui * wr;
            ur = nr;
        }
        at += half;
        wi = -sqrt((1.0 - wr) / 2.0);  // the forward direction; the inverse one is the exact negation
        wr = sqrt((1.0 + wr) / 2.0);
    }
    if (m == 0) { h[0].x = 1.0; h[0].y = 0.0; }
    int rc = TSDRGPU_OK;
    if (hipMalloc((void **)d_tw, sizeof(double2) * count) != hipSuccess) {
        rc = tsdr_fail(g, TSDRGPU_ENOMEM, "exact FFT", "twiddle table (device)");
    } else if (hipMemcpy(*d_tw, h, sizeof(double2) * count, hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(*d_tw);
        *d_tw = nullptr;
        rc = tsdr_fail(g, TSDRGPU_EHIP, "exact FFT", "twiddle table upload");
    }
    free(h);
    return rc;
}

static int fftx_plan(int m, FftxTrip *trips)
{
    int count = 0;
    if (m == 0) return 0;
    const int first = m < 7 ? m : 7;
    trips[count].s0 = 0;
    trips[count].L = first;
    count++;
    int rest = m - first, s = first;
    if (rest > 0) {
        const int parts = (rest + 7) / 8;
        for (int p = 0; p < parts; p++) {
            const int L = rest / parts + (p < rest % parts ? 1 : 0);
            trips[count].s0 = s;
            trips[count].L = L;
            count++;
            s += L;
        }
    }
    return count;
}

static const FftxKeep FFTX_KEEP_ALL = {0, -1, 0u, 0u, 0u, 0u};

template <bool FIRST, int U, bool INV>
static bool fftx_launch_fast(tsdrgpu_t *g, hipStream_t st, int L, unsigned blocks, const float *src, int src_mode, long long src_stride, float2 *z,
                             float *mag, uint32_t n, int m, int s0, int batch, const double2 *d_tw, int epi, const FftxKeep &keep)
{
    const float inv_nf = 1.0f / (float)n;  // exact: n is a power of two
    switch (L) {
        case 5: TSDR_LAUNCH(g, PROF_FFT_PASS, st, (k_fftx_fast<5, FIRST, U, INV>), blocks, 256, src, src_mode, src_stride, z, mag, n, m, s0, batch, d_tw, epi, inv_nf, keep); return true;
        case 6: TSDR_LAUNCH(g, PROF_FFT_PASS, st, (k_fftx_fast<6, FIRST, U, INV>), blocks, 256, src, src_mode, src_stride, z, mag, n, m, s0, batch, d_tw, epi, inv_nf, keep); return true;
        case 7: TSDR_LAUNCH(g, PROF_FFT_PASS, st, (k_fftx_fast<7, FIRST, U, INV>), blocks, 256, src, src_mode, src_stride, z, mag, n, m, s0, batch, d_tw, epi, inv_nf, keep); return true;
        case 8: TSDR_LAUNCH(g, PROF_FFT_PASS, st, (k_fftx_fast<8, FIRST, U, INV>), blocks, 256, src, src_mode, src_stride, z, mag, n, m, s0, batch, d_tw, epi, inv_nf, keep); return true;
        default: return false;
    }
}

template <int U>
static bool fftx_launch_fast_u(tsdrgpu_t *g, hipStream_t st, int L, unsigned blocks, const float *src, int src_mode, long long src_stride, float2 *z,
                               float *mag, uint32_t n, int m, int s0, int batch, const double2 *d_tw, int inverse, int epi, const FftxKeep &keep)
{
    if (s0 == 0)
        return inverse ? fftx_launch_fast<true, U, true>(g, st, L, blocks, src, src_mode, src_stride, z, mag, n, m, s0, batch, d_tw, epi, keep)
                       : fftx_launch_fast<true, U, false>(g, st, L, blocks, src, src_mode, src_stride, z, mag, n, m, s0, batch, d_tw, epi, keep);
    return inverse ? fftx_launch_fast<false, U, true>(g, st, L, blocks, src, src_mode, src_stride, z, mag, n, m, s0, batch, d_tw, epi, keep)
                   : fftx_launch_fast<false, U, false>(g, st, L, blocks, src, src_mode, src_stride, z, mag, n, m, s0, batch, d_tw, epi, keep);
}

// One transform (forward: from `src`; inverse: from the real array `mag`) of `batch` windows into z.  `keep` filters what
// the LAST trip stores (the inverse transform of the autocorrelation); intermediate trips store everything.
static int fftx_transform(tsdrgpu_t *g, hipStream_t st, const float *src, int src_mode, long long src_stride, float2 *z, float *mag,
                          uint32_t n, int m, int batch, const double2 *d_tw, int inverse, int epilogue, const FftxKeep &keep = FFTX_KEEP_ALL)
{
    static const int generic_only = getenv("TSDRGPU_FFTX_GENERIC") ? 1 : 0;  // A/B switch: the round-2 kernel for every trip
    FftxTrip trips[8];
    const int nt = fftx_plan(m, trips);
    for (int k = 0; k < nt; k++) {
        const int s0 = trips[k].s0, L = trips[k].L;
        const unsigned R = 1u << L;
        unsigned C = 4096u / R;
        const unsigned width = s0 == 0 ? (n >> L) : (1u << s0);  // columns available to a tile
        const bool last = k == nt - 1;
        const int epi = last ? epilogue : 0;
        // the specialised kernel wants full 4096-point tiles, a tile count that is a multiple of 8 and L in 5..8
        const bool fast_ok = !generic_only && C <= width && n >= (1u << 15) && L >= 5 && L <= 8;
        bool done = false;
        if (fast_ok) {
            const unsigned blocks = (n / 4096u) * (unsigned)batch;
            const FftxKeep &kp = last ? keep : FFTX_KEEP_ALL;
            static const int two = (getenv("TSDRGPU_FFTX_U") && getenv("TSDRGPU_FFTX_U")[0] == '2') ? 1 : 0;  // groups per thread and pass (A/B)
            done = two ? fftx_launch_fast_u<2>(g, st, L, blocks, src, src_mode, src_stride, z, mag, n, m, s0, batch, d_tw, inverse, epi, kp)
                       : fftx_launch_fast_u<1>(g, st, L, blocks, src, src_mode, src_stride, z, mag, n, m, s0, batch, d_tw, inverse, epi, kp);
        }
        if (!done) {
            if (C > width) C = width;
            const unsigned tiles = n / (R * C);
            TSDR_LAUNCH(g, PROF_FFT_PASS, st, k_fftx_trip, dim3(tiles, batch), 256, src, src_mode, src_stride, z, mag, n, m, s0, L, (int)C, d_tw,
                        inverse, epi, (float)n);
        }
    }
    if (hipGetLastError() != hipSuccess) return tsdr_fail(g, TSDRGPU_EHIP, "exact FFT", "launch");
    return TSDRGPU_OK;
}

template <int U>
static bool fftx_launch_mid(tsdrgpu_t *g, hipStream_t st, int L, unsigned blocks, const float2 *zin, float2 *zout, uint32_t n, int m, int batch,
                            const double2 *d_tw)
{
    const float inv_nf = 1.0f / (float)n;  // exact: n is a power of two
    switch (L) {
        case 5: TSDR_LAUNCH(g, PROF_FFT_PASS, st, (k_fftx_mid<5, U>), blocks, 256, zin, zout, n, m, batch, d_tw, inv_nf); return true;
        case 6: TSDR_LAUNCH(g, PROF_FFT_PASS, st, (k_fftx_mid<6, U>), blocks, 256, zin, zout, n, m, batch, d_tw, inv_nf); return true;
        case 7: TSDR_LAUNCH(g, PROF_FFT_PASS, st, (k_fftx_mid<7, U>), blocks, 256, zin, zout, n, m, batch, d_tw, inv_nf); return true;
        case 8: TSDR_LAUNCH(g, PROF_FFT_PASS, st, (k_fftx_mid<8, U>), blocks, 256, zin, zout, n, m, batch, d_tw, inv_nf); return true;
        default: return false;
    }
}

// The five-trip form of the autocorrelation (k_fftx_mid): forward trips but the last in `work`, the middle trip from `work`
// into z, the inverse's remaining trips — the forward's plan backwards — in place in z.  Returns false (nothing queued) when
// the plan does not qualify; the caller then runs the six-trip form.
static bool fftx_correlate_fused(tsdrgpu_t *g, hipStream_t st, const float *d_in, int in_is_iq, long long stride, int cnt, uint32_t n, int m,
                                 const double2 *d_tw, float2 *z, float2 *work, const FftxKeep &keep)
{
    static const int off = (getenv("TSDRGPU_FFTX_GENERIC") || (getenv("TSDRGPU_FFTX_FUSED") && getenv("TSDRGPU_FFTX_FUSED")[0] == '0')) ? 1 : 0;
    if (off || n < (1u << 15)) return false;
    FftxTrip trips[8];
    const int nt = fftx_plan(m, trips);
    if (nt < 2) return false;
    for (int k = 0; k < nt; k++) {
        const unsigned width = trips[k].s0 == 0 ? (n >> trips[k].L) : (1u << trips[k].s0);
        if (trips[k].L < 5 || trips[k].L > 8 || (4096u >> trips[k].L) > width) return false;
    }
    for (int j = nt - 2, s0 = trips[nt - 1].L; j >= 0; s0 += trips[j].L, j--)  // the same plan backwards: do its tiles fit too?
        if ((4096u >> trips[j].L) > (1u << s0)) return false;
    const unsigned blocks = (n / 4096u) * (unsigned)cnt;
    const int src_mode = in_is_iq;  // the input kind: 0 real, 1 interleaved IQ, 3 retained sums of squares
    for (int k = 0; k < nt - 1; k++)
        if (!fftx_launch_fast_u<1>(g, st, trips[k].L, blocks, d_in, src_mode, stride, work, nullptr, n, m, trips[k].s0, cnt, d_tw, 0, 0, FFTX_KEEP_ALL))
            return false;
    const int Lm = trips[nt - 1].L;
    if (!fftx_launch_mid<1>(g, st, Lm, blocks, work, z, n, m, cnt, d_tw)) return false;
    int s0 = Lm;
    for (int j = nt - 2; j >= 0; j--) {  // the inverse: L_last was the middle trip's, then the forward's plan backwards
        const int L = trips[j].L;
        (void)fftx_launch_fast_u<1>(g, st, L, blocks, nullptr, 0, 0, z, nullptr, n, m, s0, cnt, d_tw, 1, 0, j == 0 ? keep : FFTX_KEEP_ALL);
        s0 += L;
    }
    return true;
}

// fft_autocorrelation for `cnt` windows, exactly: answer = IFFT( | FFT(x) / N | ), fft.c:49-64.
// z: cnt*n complex (the result), mag: cnt*n COMPLEX points of scratch (the six-trip form keeps its cnt*n magnitudes there).
static int fftx_correlate_keep(tsdrgpu_t *g, hipStream_t st, const float *d_in, int in_is_iq, long long stride, int cnt, uint32_t n,
                               const double2 *d_tw, float2 *z, float *mag, const FftxKeep &keep)
{
    int m = 0;
    while ((1u << m) < n) m++;
    int rc;
    if (m == 0) return tsdr_fail(g, TSDRGPU_EINVAL, "exact FFT", "transform too short");
    if (fftx_correlate_fused(g, st, d_in, in_is_iq, stride, cnt, n, m, d_tw, z, reinterpret_cast<float2 *>(mag), keep)) {
        if (hipGetLastError() != hipSuccess) return tsdr_fail(g, TSDRGPU_EHIP, "exact FFT", "launch");
        return TSDRGPU_OK;
    }
    if ((rc = fftx_transform(g, st, d_in, in_is_iq, stride, z, mag, n, m, cnt, d_tw, 0, 1))) return rc;
    return fftx_transform(g, st, mag, 0, (long long)n, z, mag, n, m, cnt, d_tw, 1, 0, keep);
}

int fftx_correlate(tsdrgpu_t *g, hipStream_t st, const float *d_in, int in_is_iq, long long stride, int cnt, uint32_t n,
                   const double2 *d_tw, float2 *z, float *mag)
{
    return fftx_correlate_keep(g, st, d_in, in_is_iq, stride, cnt, n, d_tw, z, mag, FFTX_KEEP_ALL);
}

// fft_autocorrelation + accummulate for `cnt` windows, exactly.  z: cnt*n complex, mag: cnt*n complex points of scratch.  Only the lag
// windows (and lag 0) of the correlations are stored, except for window `full_w` of the batch (-1: none), kept whole.
int fftx_autocorr(tsdrgpu_t *g, hipStream_t st, const float *d_in, int in_is_iq, long long stride, int cnt, uint32_t n,
                  const double2 *d_tw, float2 *z, float *mag, int frame_lo, int frame_len, int line_lo, int line_len, double *d_plots,
                  unsigned long long calls_before, int mode, int full_w)
{
    FftxKeep keep;
    keep.on = 1;
    keep.full_w = full_w;
    keep.lo0 = (unsigned)frame_lo;
    keep.hi0 = (unsigned)(frame_lo + frame_len);
    keep.lo1 = (unsigned)line_lo;
    keep.hi1 = (unsigned)(line_lo + line_len);
    const int rc = fftx_correlate_keep(g, st, d_in, in_is_iq, stride, cnt, n, d_tw, z, mag, keep);
    if (rc) return rc;
    const int L = frame_len + line_len + 1;  // + the lag-0 entry
    TSDR_LAUNCH(g, PROF_ACCUMULATE, st, k_fftx_accumulate, (L + 255) / 256, 256, z, n, cnt, frame_lo, frame_len, line_lo, line_len, d_plots,
                calls_before, mode);
    if (hipGetLastError() != hipSuccess) return tsdr_fail(g, TSDRGPU_EHIP, "exact FFT", "accumulate");
    return TSDRGPU_OK;
}

// fft_perform (fft.c:96-176) on n complex points: d_z -> d_work (n complex), exactly
int fftx_perform(tsdrgpu_t *g, hipStream_t st, const float2 *d_z, float2 *d_work, uint32_t n, const double2 *d_tw, int inverse)
{
    int m = 0;
    while ((1u << m) < n) m++;
    if (m == 0) return TSDRGPU_OK;
    return fftx_transform(g, st, (const float *)d_z, 2, (long long)n, d_work, nullptr, n, m, 1, d_tw, inverse, inverse ? 0 : 2);
}

// ---------------------------------------------------------------------------
// Super-bandwidth stitch (superbandwidth.c:67-152, fft.c:69-93) with the exact FFT: the same sequence as
// tsdrgpu_superb_stitch, every float operation in the reference's order (this file is -ffp-contract=off),
// so hop offsets AND the stitched signal are bit-identical.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_x_abs_diff(const float2 *__restrict__ z, float2 *__restrict__ out, unsigned n)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float2 c = z[i];
    const float cur = sqrtf(c.x * c.x + c.y * c.y);
    float prev;
    if (i == 0) prev = c.x * c.x + c.y * c.y;  // |z0|^2, superbandwidth.c:70
    else {
        const float2 p = z[i - 1];
        prev = sqrtf(p.x * p.x + p.y * p.y);
    }
    out[i] = make_float2(cur - prev, 0.f);
}

__global__ __launch_bounds__(256) void k_x_mul_conj(float2 *__restrict__ a, const float2 *__restrict__ b, unsigned n)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float2 x = a[i], y = b[i];
    a[i] = make_float2(x.x * y.x + x.y * y.y, x.x * y.y - x.y * y.x);  // fft.c:80-89
}

__global__ __launch_bounds__(1024) void k_x_argmax_abs(const float2 *__restrict__ z, unsigned n, int *__restrict__ out_floats)
{
    float best = -1.f;
    int at = 0x7fffffff;
    for (unsigned i = threadIdx.x; i < n; i += blockDim.x) {
        const float2 c = z[i];
        const float v = sqrtf(c.x * c.x + c.y * c.y);
        if (v > best) { best = v; at = (int)i; }
    }
    __shared__ float sb[16];
    __shared__ int si[16];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_down(best, o, 64);
        const int oi = __shfl_down(at, o, 64);
        if (ob > best || (ob == best && oi < at)) { best = ob; at = oi; }
    }
    if ((threadIdx.x & 63) == 0) { sb[threadIdx.x >> 6] = best; si[threadIdx.x >> 6] = at; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; w++)
            if (sb[w] > best || (sb[w] == best && si[w] < at)) { best = sb[w]; at = si[w]; }
        *out_floats = 2 * at;  // first maximum, offset in floats (superbandwidth.c:100-116)
    }
}

__global__ __launch_bounds__(256) void k_x_rotate(const float *__restrict__ in, float *__restrict__ out, unsigned nfloats,
                                                  const int *__restrict__ off)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nfloats) return;
    unsigned s = i + (unsigned)*off;
    if (s >= nfloats) s -= nfloats;
    out[i] = in[s];
}

static uint32_t x_pow2_floor(uint32_t v)
{
    uint32_t m = 0;
    while ((v /= 2) != 0) m++;
    return 1u << m;
}

extern "C" int tsdrgpu_superb_stitch_exact(tsdrgpu_t *g, float *const *d_hops, int nhops, int gathered, int samples_in_frame,
                                           float *d_out, int32_t *h_offsets, uint32_t *h_total)
{
    if (!g || !d_hops || nhops < 1 || nhops > 64 || gathered < 2 || samples_in_frame < 1 || !d_out)
        return g ? tsdr_fail(g, TSDRGPU_EINVAL, "tsdrgpu_superb_stitch_exact", "bad argument") : TSDRGPU_EINVAL;
    const uint32_t per = x_pow2_floor((uint32_t)gathered);
    const uint32_t total = (uint32_t)nhops * per;
    const uint32_t nfl = per * 2;
    const int bsize = ((int)nfl / samples_in_frame) * samples_in_frame;
    if (bsize < 2) return tsdr_fail(g, TSDRGPU_EINVAL, "tsdrgpu_superb_stitch_exact", "hop shorter than one frame");
    const uint32_t bn = x_pow2_floor((uint32_t)bsize) / 2;
    const uint32_t nfft = x_pow2_floor(total);
    hipStream_t st = g->stream;
    HIP_TRY(g, hipStreamSynchronize(st));
    double2 *tw_b = nullptr, *tw_p = nullptr, *tw_t = nullptr;
    float2 *w = nullptr;
    int *d_off = nullptr;
    int rc = TSDRGPU_OK;
    const size_t big = nfft > per ? nfft : per;
    do {
        if ((rc = fftx_build_table(g, bn, &tw_b)) || (rc = fftx_build_table(g, per, &tw_p)) || (rc = fftx_build_table(g, nfft, &tw_t))) break;
        if (hipMalloc(&w, sizeof(float2) * (4 * (size_t)bn + 2 * big)) != hipSuccess || hipMalloc(&d_off, sizeof(int) * nhops) != hipSuccess) {
            rc = tsdr_fail(g, TSDRGPU_ENOMEM, "tsdrgpu_superb_stitch_exact", "work buffers");
            break;
        }
        float2 *A = w, *B = w + bn, *FA = w + 2 * (size_t)bn, *FB = w + 3 * (size_t)bn, *T1 = w + 4 * (size_t)bn, *T2 = T1 + big;
        (void)hipMemsetAsync(d_off, 0, sizeof(int) * nhops, st);
        for (int i = 1; i < nhops && !rc; i++) {
            TSDR_LAUNCH(g, PROF_SUPERB_MISC, st, k_x_abs_diff, (bn + 255) / 256, 256, (const float2 *)d_hops[0], A, bn);
            TSDR_LAUNCH(g, PROF_SUPERB_MISC, st, k_x_abs_diff, (bn + 255) / 256, 256, (const float2 *)d_hops[i], B, bn);
            if ((rc = fftx_perform(g, st, A, FA, bn, tw_b, 0)) || (rc = fftx_perform(g, st, B, FB, bn, tw_b, 0))) break;
            TSDR_LAUNCH(g, PROF_SUPERB_MISC, st, k_x_mul_conj, (bn + 255) / 256, 256, FA, FB, bn);
            if ((rc = fftx_perform(g, st, FA, A, bn, tw_b, 1))) break;
            TSDR_LAUNCH(g, PROF_ARGMAX, st, k_x_argmax_abs, 1, 1024, A, bn, d_off + i);
            TSDR_LAUNCH(g, PROF_SUPERB_MISC, st, k_x_rotate, (nfl + 255) / 256, 256, d_hops[i], (float *)T1, nfl, d_off + i);
            if ((rc = fftx_perform(g, st, T1, T2, per, tw_p, 0))) break;
            (void)hipMemcpyAsync(d_hops[i], T2, sizeof(float2) * per, hipMemcpyDeviceToDevice, st);
        }
        if (rc) break;
        if ((rc = fftx_perform(g, st, (const float2 *)d_hops[0], T2, per, tw_p, 0))) break;
        (void)hipMemcpyAsync(d_hops[0], T2, sizeof(float2) * per, hipMemcpyDeviceToDevice, st);
        for (int i = 0; i < nhops; i++)
            (void)hipMemcpyAsync(d_out + (size_t)i * per * 2, d_hops[i], sizeof(float2) * per, hipMemcpyDeviceToDevice, st);
        // inverse FFT over the largest power of two <= total (fft_perform truncates, fft.c:101-105)
        (void)hipMemcpyAsync(T1, d_out, sizeof(float2) * nfft, hipMemcpyDeviceToDevice, st);
        if ((rc = fftx_perform(g, st, T1, T2, nfft, tw_t, 1))) break;
        (void)hipMemcpyAsync(d_out, T2, sizeof(float2) * nfft, hipMemcpyDeviceToDevice, st);
        if (h_offsets && hipMemcpyAsync(h_offsets, d_off, sizeof(int) * nhops, hipMemcpyDeviceToHost, st) != hipSuccess)
            rc = tsdr_fail(g, TSDRGPU_EHIP, "tsdrgpu_superb_stitch_exact", "copy back");
    } while (0);
    if (hipStreamSynchronize(st) != hipSuccess && !rc) rc = tsdr_fail(g, TSDRGPU_EHIP, "tsdrgpu_superb_stitch_exact", "stream");
    if (hipGetLastError() != hipSuccess && !rc) rc = tsdr_fail(g, TSDRGPU_EHIP, "tsdrgpu_superb_stitch_exact", "kernel");
    (void)hipFree(tw_b);
    (void)hipFree(tw_p);
    (void)hipFree(tw_t);
    (void)hipFree(w);
    (void)hipFree(d_off);
    if (rc) return rc;
    if (h_total) *h_total = total;
    return TSDRGPU_OK;
}
