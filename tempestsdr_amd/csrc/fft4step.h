// fft4step.h — register DFTs, twiddle helpers and the THREE-TRIP ("four-step") plan of the
// autocorrelation (fft_autocorrelation, TempestSDR/src/fft.c:49-64) for gfx950.
//
// The capture window (real, N samples) is transformed as nh = N/2 complex points
// z[m] = x[2m] + i x[2m+1] (tsdrgpu_fft.hip explains the packed-real split).  With nh = N1 * N2,
// N2 = 4096, input index n = N2*n1 + n2 and output index k = k1 + N1*k2:
//
//   Z[k1 + N1 k2] = sum_n2 w_N2^(n2 k2) [ w_nh^(n2 k1) sum_n1 z[N2 n1 + n2] w_N1^(n1 k1) ]
//
//   trip 1  k_ac_cols<.., false>   column DFTs of length N1 over n1 (rows N2 apart, tiles of C >= 16
//                                  neighbouring columns = runs of >= 128 bytes), am_demod fused into the
//                                  load; Y[k1][n2] stored at [k1*N2 + n2]
//   trip 2  k_ac_rows              for the row pair (k1, N1-k1), 16 KiB contiguous each: twiddle
//                                  w_nh^(n2 k1), DFT-4096 in LDS, the packed-real split / 1/N /
//                                  magnitude / re-pack on the pairs Z[k] <-> Z[nh-k] (which live in
//                                  exactly these two rows), the DFT-4096 of the inverse transform and
//                                  its twiddle; stored in place
//   trip 3  k_ac_cols<.., true>    column DFTs over k1; conjugate; only the two lag windows the
//                                  detector reads (frameratedetector.c:115-118) are stored
//
// i.e. 8N + 4N | 4N + 4N | 4N + (lags) bytes per window with fused demodulation — below SURVEY 8(d)'s
// one-pass-per-transform figure of 28N + 16L, where the previous plan (three radix-128 Stockham passes
// each way, the middle two fused) made five trips.
//
// Everything here is plain C++ over float2 so that tests/emu can run the very same kernels on the CPU
// (threads = pthreads, __shared__ = static, __syncthreads = barrier) against numpy before a GPU sees them.
#pragma once

#if defined(__HIP_DEVICE_COMPILE__) && !defined(AC4_SCALAR_COMPLEX)
// Complex arithmetic on register PAIRS with the packed float32 instructions of gfx950 (VOP3P, full rate: two flops per
// lane and issue).  A sum or difference is one v_pk_add_f32; multiplying by -i is not an operation at all but a pair
// of operand modifiers on the addition that consumes it (op_sel swaps the halves, neg_hi / neg_lo flips one sign); a
// complex product is v_pk_mul_f32 + v_pk_fma_f32 with the broadcast, the swap and the sign as modifiers.  The compiler
// does not find these forms by itself: from scalar code its SLP vectoriser reaches v_pk_* but pays for gathering the
// pairs with register moves (725 v_mov_b32 among k_ac_rows' 2 481 VALU instructions), and from <2 x float> code it
// materialises every swap and negation (a v_pk_add with 0 plus two moves per multiplication by -i) — hence the few
// lines of inline assembly.  Operand modifiers: op_sel[i] / op_sel_hi[i] = which half of source i feeds the low / high
// result lane (0 low, 1 high; defaults 0 / 1); neg_lo[i] / neg_hi[i] negate source i for that lane.
typedef float ac4_v2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ ac4_v2 ac4_v(float2 a) { return __builtin_bit_cast(ac4_v2, a); }
__device__ __forceinline__ float2 ac4_f(ac4_v2 a) { return __builtin_bit_cast(float2, a); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return ac4_f(ac4_v(a) + ac4_v(b)); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return ac4_f(ac4_v(a) - ac4_v(b)); }
// a + (-i) b = (a.x + b.y, a.y - b.x)
__device__ __forceinline__ float2 cadd_mi(float2 a, float2 b)
{
    ac4_v2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(ac4_v(a)), "v"(ac4_v(b)));
    return ac4_f(r);
}
// a - (-i) b = (a.x - b.y, a.y + b.x)
__device__ __forceinline__ float2 csub_mi(float2 a, float2 b)
{
    ac4_v2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(ac4_v(a)), "v"(ac4_v(b)));
    return ac4_f(r);
}
// a * (-i) = (a.y, -a.x), where no addition follows that could absorb it: one multiplication by the pair (1, -1)
__device__ __forceinline__ float2 mul_mi(float2 a)
{
    ac4_v2 r;
    const ac4_v2 c = {1.0f, -1.0f};
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(r) : "v"(ac4_v(a)), "v"(c));
    return ac4_f(r);
}
// (a.x b.x - a.y b.y, a.x b.y + a.y b.x)
__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
    ac4_v2 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(ac4_v(a)), "v"(ac4_v(b)));  // (a.x b.x, a.x b.y)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"          // + (a.y (-b.y), a.y b.x)
        : "=v"(r)
        : "v"(ac4_v(a)), "v"(ac4_v(b)), "v"(t));
    return ac4_f(r);
}
#else
__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }  // a * (-i)
__device__ __forceinline__ float2 cadd_mi(float2 a, float2 b) { return make_float2(a.x + b.y, a.y - b.x); }  // a + (-i) b
__device__ __forceinline__ float2 csub_mi(float2 a, float2 b) { return make_float2(a.x - b.y, a.y + b.x); }  // a - (-i) b
#endif

// ---------------------------------------------------------------------------
// small DFTs in registers (forward, e^{-2 pi i/R}), outputs in natural order
// ---------------------------------------------------------------------------
__device__ __forceinline__ void dft2(float2 &a, float2 &b)
{
    const float2 t = a;
    a = cadd(t, b);
    b = csub(t, b);
}

__device__ __forceinline__ void dft4(float2 &a0, float2 &a1, float2 &a2, float2 &a3)
{
    const float2 s02 = cadd(a0, a2), d02 = csub(a0, a2);
    const float2 s13 = cadd(a1, a3), d13 = csub(a1, a3);
    a0 = cadd(s02, s13);
    a1 = cadd_mi(d02, d13);  // d02 + (-i) d13
    a2 = csub(s02, s13);
    a3 = csub_mi(d02, d13);
}

template <int R>
__device__ __forceinline__ void dft_reg(float2 (&v)[R]);

template <>
__device__ __forceinline__ void dft_reg<1>(float2 (&v)[1]) {}
template <>
__device__ __forceinline__ void dft_reg<2>(float2 (&v)[2]) { dft2(v[0], v[1]); }
template <>
__device__ __forceinline__ void dft_reg<4>(float2 (&v)[4]) { dft4(v[0], v[1], v[2], v[3]); }

template <>
__device__ __forceinline__ void dft_reg<8>(float2 (&v)[8])
{
    // 8 = 2 x 4: X[k1 + 2*k2] = sum_{n2<4} w8^{n2*k1} (sum_{n1<2} x[4*n1+n2] w2^{n1 k1}) w4^{n2 k2}
    const float h = 0.70710678118654752440f;
#pragma unroll
    for (int n2 = 0; n2 < 4; n2++) dft2(v[n2], v[n2 + 4]);
    v[5] = cmul(v[5], make_float2(h, -h));
    v[6] = mul_mi(v[6]);
    v[7] = cmul(v[7], make_float2(-h, -h));
    dft4(v[0], v[1], v[2], v[3]);  // k1 = 0 -> X[0], X[2], X[4], X[6]
    dft4(v[4], v[5], v[6], v[7]);  // k1 = 1 -> X[1], X[3], X[5], X[7]
    const float2 x0 = v[0], x2 = v[1], x4 = v[2], x6 = v[3];
    const float2 x1 = v[4], x3 = v[5], x5 = v[6], x7 = v[7];
    v[0] = x0; v[1] = x1; v[2] = x2; v[3] = x3; v[4] = x4; v[5] = x5; v[6] = x6; v[7] = x7;
}

template <>
__device__ __forceinline__ void dft_reg<16>(float2 (&v)[16])
{
    // 16 = 4 x 4: a[n2][k1] = DFT4 over n1 of x[4*n1+n2]; times w16^{n2*k1}; X[k1+4*k2] = DFT4 over n2
    const float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f, h = 0.70710678118654752440f;
#pragma unroll
    for (int n2 = 0; n2 < 4; n2++) dft4(v[n2], v[n2 + 4], v[n2 + 8], v[n2 + 12]);
    v[5] = cmul(v[5], make_float2(c1, -s1));     // n2=1,k1=1: w^1
    v[6] = cmul(v[6], make_float2(h, -h));       // n2=2,k1=1: w^2
    v[7] = cmul(v[7], make_float2(s1, -c1));     // n2=3,k1=1: w^3
    v[9] = cmul(v[9], make_float2(h, -h));       // n2=1,k1=2: w^2
    v[10] = mul_mi(v[10]);                       // n2=2,k1=2: w^4
    v[11] = cmul(v[11], make_float2(-h, -h));    // n2=3,k1=2: w^6
    v[13] = cmul(v[13], make_float2(s1, -c1));   // n2=1,k1=3: w^3
    v[14] = cmul(v[14], make_float2(-h, -h));    // n2=2,k1=3: w^6
    v[15] = cmul(v[15], make_float2(-c1, s1));   // n2=3,k1=3: w^9
#pragma unroll
    for (int k1 = 0; k1 < 4; k1++) dft4(v[4 * k1], v[4 * k1 + 1], v[4 * k1 + 2], v[4 * k1 + 3]);
    float2 t[16];
#pragma unroll
    for (int k1 = 0; k1 < 4; k1++)
#pragma unroll
        for (int k2 = 0; k2 < 4; k2++) t[k1 + 4 * k2] = v[4 * k1 + k2];
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = t[i];
}

// ---------------------------------------------------------------------------
// twiddles: exp(-2 pi i e / span) for a power-of-two span, from an exactly representable dyadic fraction
// ---------------------------------------------------------------------------
// WIDE: spans above 2^24 (the 2^25-point stitch transform, column length 2048): r itself may not be a float, r - span
// (|.| <= 2^24) is, and the angle differs by a whole turn
template <bool WIDE = false>
__device__ __forceinline__ float2 tw_exact(unsigned e, unsigned mask, float inv)
{
    float sn, cs;
    const unsigned r = e & mask;
    const float f = (WIDE && r > (mask >> 1)) ? -(float)(mask - r + 1u) : (float)r;
    sincospif(f * inv, &sn, &cs);
    return make_float2(cs, sn);
}

// pw[i] = w^(e1*i), i < M (a power of two): log2(M) accurate evaluations, the rest complex products of
// depth <= log2(M)-1 (a few f32 ulps)
template <int M, bool WIDE = false>
__device__ __forceinline__ void tw_powers(float2 (&pw)[M], unsigned e1, unsigned mask, float inv)
{
    pw[0] = make_float2(1.f, 0.f);
#pragma unroll
    for (int bit = 1; bit < M; bit <<= 1) {
        pw[bit] = tw_exact<WIDE>(e1 * (unsigned)bit, mask, inv);
#pragma unroll
        for (int i = bit + 1; i < 2 * bit; i++) pw[i] = cmul(pw[bit], pw[i - bit]);
    }
}

// Output filter of a transform's last pass: when `on`, only outputs whose index lies in one of two
// ranges are stored (the autocorrelation reads nothing but its two lag windows, frameratedetector.c:
// 115-118) plus point 0 (lag 0, the scale of the argmax certificate), except for transform `full_b`
// of the batch, which is stored whole.
// (Round 5 also built the fold the review asked for — accummulate inside trip 3: a workgroup per column tile walking the launch's
// windows in window order, the running means of its lags in LDS slots, bit-identical plots, no lag-window store / re-load, no
// k_accumulate launch — and measured it SLOWER: 0.2672 / 0.2695 ms per pass for the two column trips against 0.2399 + 0.0221
// with the separate kernel at N = 2^22, 0.70 against 0.59 + 0.04 at 2^23.  A workgroup that walks the windows leaves a CU
// (N1 tiles x threads) / CUs = 8 waves where the flat grid keeps 16 resident, and trip 3 lives on memory-level parallelism.
// Commit 0de10d1 has the kernel.)
struct FftKeep {
    int on;
    int full_b;
    unsigned lo0, hi0, lo1, hi1;
};

// Magnitudes of the float32 (tolerance-stated) form take the hardware square root (1 ulp) instead of the
// correctly rounded sequence the build flags give sqrtf()
#if defined(__HIP_DEVICE_COMPILE__)
#define AC_SQRT(x) __builtin_amdgcn_sqrtf(x)
#else
#define AC_SQRT(x) sqrtf(x)
#endif

// Packed-real split of the autocorrelation (see tsdrgpu_fft.hip): A = Z[k], bm = Z[nh-k],
// wk = exp(-i pi k/nh); returns Zin[k] (*zk) and Zin[nh-k] (*zkm) of the inverse transform's input.
__device__ __forceinline__ void ac_split_pair(float2 a, float2 bm, float2 wk, unsigned nh, float2 *zk, float2 *zkm)
{
    const float inv_n = 1.0f / (float)(2 * nh);
    const float2 b = make_float2(bm.x, -bm.y);
    const float cs = wk.x, sn = wk.y;
    const float2 sum = make_float2(0.5f * (a.x + b.x), 0.5f * (a.y + b.y));
    const float2 dif = make_float2(0.5f * (a.x - b.x), 0.5f * (a.y - b.y));
    const float2 t = cmul(make_float2(cs, sn), dif);
    const float2 xk = make_float2(sum.x + t.y, sum.y - t.x);
    const float2 xm = make_float2(sum.x - t.y, -(sum.y + t.x));
    const float mk = AC_SQRT(xk.x * xk.x + xk.y * xk.y) * inv_n;
    const float mm = AC_SQRT(xm.x * xm.x + xm.y * xm.y) * inv_n;
    const float s = mk + mm, d = mk - mm;
    *zk = make_float2(s + sn * d, cs * d);
    *zkm = make_float2(s - sn * d, cs * d);
}

// ---------------------------------------------------------------------------
// trips 1 and 3: column DFTs of length N1 = 2^LOGN1 (16..1024), rows N2 = nh/N1 elements apart.
// A workgroup owns C neighbouring columns (C = 16 for N1 >= 256, else 4096/N1: at least 4096 points);
// every thread holds 16 points.  Stockham passes of radix {2,4,8 first if log2(N1) is no multiple of
// 4, then 16 ...} with the tile exchanged through LDS as [point][column] between passes: the first pass
// loads from global memory, the last one stores to it, so a 512-point column costs two LDS round trips.
//   IN_MODE 0: complex input (trip 3)
//   IN_MODE 3: real samples packed two per point; IN_MODE 4: the same, demodulated from interleaved IQ
//              (am_demod, TSDRLibrary.c:244-262) — trip 1
//   LAST: conjugate the result and store only what `keep` asks for (trip 3)
// ---------------------------------------------------------------------------
#define AC4_ROW 4096u  // N2: the row length of the plan

// Columns per tile: 16 (runs of 128 bytes of complex points) up to column length 512, where the tile is 64 KiB and two
// workgroups share a CU; at 1024 sixteen columns are 128 KiB — one workgroup of 1024 threads per CU, every barrier a
// stall of the whole CU — so there the tile is 8 columns wide: 64 KiB and 512 threads again (trip 1's points are 16
// bytes of IQ, so its runs stay 128 bytes; trip 3 reads 64-byte runs).  Measured at N = 2^23: 0.642 -> 0.587 ms per 17
// windows; at column length 512 eight columns change nothing (0.2489 vs 0.2486 ms).
#ifndef AC4_C8_FROM
#define AC4_C8_FROM 1024u
#endif
template <int LOGN1>
struct ColGeom {
    static constexpr unsigned N1 = 1u << LOGN1;
    // (column length 2048 — the super-bandwidth stitch of hops of 2^23 points — keeps 8 columns: 128 KiB, one workgroup of
    // 1024 threads per CU)
    // (measured and dropped: 4 columns at column length 2048 — 64 + 16 KiB, two workgroups of 512 threads per CU again, but runs
    // of 32 bytes: the stitch's three big trips 155 / 175 -> 177 / 237 us)
    static constexpr unsigned C = (N1 >= AC4_C8_FROM) ? 8u : (N1 >= 256u) ? 16u : 4096u / N1;
    static constexpr unsigned NT = N1 * C / 16u;  // threads per workgroup
    static constexpr unsigned Q = N1 / 16u;
    static constexpr int R0 = (LOGN1 % 4) ? (1 << (LOGN1 % 4)) : 16;  // radix of the first pass
    static constexpr int NP = (LOGN1 + 3) / 4;                       // passes
};

// am_demod's re*re + im*im (TSDRLibrary.c:244-262) with the reference's bits: two rounded products and their rounded sum, no
// contraction.  For the side store of trip 1 into the retention ring: what an epoch is replayed from must lead to the
// reference's own demodulated samples, and the correctly rounded root of THIS value (taken by the replay's first trip,
// tsdrgpu_fftx.hip src_mode 3) is exactly that.
__device__ __forceinline__ float ac4_sumsq_exact(float re, float im)
{
#if defined(__HIP_DEVICE_COMPILE__)
    // (HIP's __fmul_rn / __fadd_rn are plain operators, and this translation unit is built with contraction on: the pragma
    // is what keeps re*re + im*im from becoming fma(re, re, im*im))
    float x;
    {
#pragma clang fp contract(off)
        const float rr = re * re, ii = im * im;
        x = rr + ii;
    }
    return x;
#else
    volatile float a = re * re, b = im * im;  // (volatile: no fused multiply-add on the host either)
    return a + b;
#endif
}
// What the column kernels' variants beyond the autocorrelation's own need (all unused, and compiled away, there):
//   retain  EPI 1: the ring slot of window 0 (2 nh floats per window): trip 1 leaves every sample's re*re + im*im there
//           (tsdrgpu_autocorr_set_certify mode 1; frameratedetector.c:87-126 keeps one copy too)
//   shift   IN_MODE 6: the input rotated left by that many complex points (superbandwidth.c:135-137)
//   pval / pidx  EPI 2: per workgroup the first maximum of |re| and of |im| of the results (superb_bestfit's search,
//           superbandwidth.c:100-116, over correlations that are real sequences packed two per transform)
struct AcColsAux {
    float *retain;
    unsigned shift;
    float *pval;
    int *pidx;
};

template <int IN_MODE>
__device__ __forceinline__ float2 ac4_load(const void *__restrict__ base, long long at, bool al16, unsigned nh, unsigned shift)
{
    (void)nh;
    (void)shift;
    if (IN_MODE == 0) return ((const float2 *)base)[at];
    if (IN_MODE == 3) {
        const float *x = (const float *)base;
        return make_float2(x[2 * at], x[2 * at + 1]);
    }
    if (IN_MODE == 5) {
        // complex_to_abs_diff (superbandwidth.c:67-81): the first difference of the magnitudes, element 0 seeded with
        // |z0|^2 (kept literally), imaginary part 0.  Hardware square root: this feeds a float32 correlation whose peak
        // position is all that is read.
        const float2 *z = (const float2 *)base;
#if defined(__HIPCC__)
        // z[at - 1] and z[at] as one (under-aligned) 16-byte request; point 0 has no predecessor: it reads (z0, z1)
        typedef float float4_d8 __attribute__((ext_vector_type(4), aligned(8)));
        const float4_d8 t = *reinterpret_cast<const float4_d8 *>(z + (at ? at - 1 : 0));
        const float2 c = at ? make_float2(t[2], t[3]) : make_float2(t[0], t[1]);
        const float cur = AC_SQRT(c.x * c.x + c.y * c.y);
        const float prev = at ? AC_SQRT(t[0] * t[0] + t[1] * t[1]) : c.x * c.x + c.y * c.y;
#else
        const float2 c = z[at];
        const float cur = AC_SQRT(c.x * c.x + c.y * c.y);
        float prev;
        if (at == 0) prev = c.x * c.x + c.y * c.y;
        else {
            const float2 p = z[at - 1];
            prev = AC_SQRT(p.x * p.x + p.y * p.y);
        }
#endif
        return make_float2(cur - prev, 0.f);
    }
    if (IN_MODE == 6) {
        unsigned src = (unsigned)at + shift;
        if (src >= nh) src -= nh;
        return ((const float2 *)base)[src];
    }
    // two IQ samples = 16 bytes; a window starts on an 8-byte boundary only (capture lengths are odd), and gfx950
    // moves an under-aligned dwordx4 in one instruction all the same
    (void)al16;
#if defined(__HIPCC__)
    typedef float float4_a8 __attribute__((ext_vector_type(4), aligned(8)));
    const float4_a8 t = *reinterpret_cast<const float4_a8 *>((const float2 *)base + 2 * at);
    // am_demod (TSDRLibrary.c:244-262) inside the float32 transform, whose plots carry a 1e-4 tolerance: the hardware
    // square root (1 ulp, one instruction) instead of the 15-instruction correctly rounded sequence — 32 of them per
    // thread.  (The resampler's and the bit-exact detector's demodulation stay correctly rounded.)
    return make_float2(AC_SQRT(t[0] * t[0] + t[1] * t[1]), AC_SQRT(t[2] * t[2] + t[3] * t[3]));
#else  // tests/emu: the same two samples
    const float2 a = ((const float2 *)base)[2 * at], b = ((const float2 *)base)[2 * at + 1];
    return make_float2(sqrtf(a.x * a.x + a.y * a.y), sqrtf(b.x * b.x + b.y * b.y));
#endif
}

// rtw[i] = w_nh^(ebase + estep i), i < 16
template <bool WIDE = false>
__device__ __forceinline__ void ac4_col_twiddles(float2 (&rtw)[16], unsigned estep, unsigned ebase, unsigned nh)
{
    const unsigned mask = nh - 1u;
    const float inv = -2.0f / (float)nh;
    tw_powers<16, WIDE>(rtw, estep, mask, inv);
    const float2 rbase = tw_exact<WIDE>(ebase, mask, inv);
    rtw[0] = rbase;
#pragma unroll
    for (int i = 1; i < 16; i++) rtw[i] = cmul(rbase, rtw[i]);
}

// EPI 0: results stored (trip 3: filtered by `keep`); 1: as 0, and IN_MODE 4's samples also go to aux.retain, demodulated
// with the reference's bits; 2: LAST only — nothing stored, per workgroup the first maxima of |re| and |im| to aux.pval / pidx
template <int LOGN1, int IN_MODE, bool LAST, int EPI>
__device__ __forceinline__ void ac4_cols_body(const void *__restrict__ xb, float2 *__restrict__ yb, unsigned nh, const FftKeep &keep,
                                              const unsigned b, const AcColsAux &aux)
{
    typedef ColGeom<LOGN1> G;
    constexpr unsigned N1 = G::N1, C = G::C, NT = G::NT, Q = G::Q;
    constexpr int R0 = G::R0, NP = G::NP, G0 = 16 / R0;
    constexpr bool WIDE = LOGN1 >= 11;  // only there can nh (4 x 4096 x N1 in the stitch's last trip) exceed 2^24
    __shared__ float2 L[N1 * C];
    // C == 16 (column lengths >= 256): the powers (w_nh^(col Q))^i, i < 16, of the workgroup's 16 columns are shared by
    // the 32 threads of a column — one accurate evaluation each instead of four evaluations and eleven products per
    // thread (rows padded to 17 entries: sixteen columns' reads hit sixteen different bank pairs).  The table sits behind
    // twN[] in one array so that it costs no byte where it is not used (column length 2048 in tiles of 4 columns: two
    // workgroups of 64 + 16 KiB fill a CU's LDS exactly).
    constexpr bool PTAB = (C == 16u);
    __shared__ float2 twtab[N1 + (PTAB ? 16u * 17u : 0u)];
    float2 *const twN = twtab;
    float2 *const ptw = twtab + N1;
    const unsigned N2 = nh / N1;
    const unsigned tid = threadIdx.x;
    const unsigned c = tid % C, q = tid / C;
    // consecutive tiles go to the same XCD (workgroups are dealt round-robin to the 8 XCDs): the IQ rows
    // of a window start on 8-byte boundaries only, so neighbouring tiles share a cache line at each end
    const unsigned gx = gridDim.x;
    const unsigned tile = (gx % 8u == 0u) ? (blockIdx.x % 8u) * (gx / 8u) + blockIdx.x / 8u : blockIdx.x;
    const unsigned col = tile * C + c;
    for (unsigned e = tid; e < N1; e += NT) {
        float sn, cs;
        sincospif(-2.0f * (float)e / (float)N1, &sn, &cs);
        twN[e] = make_float2(cs, sn);
    }
    if (PTAB && tid < 256u) {
        const unsigned cc = tid >> 4, i = tid & 15u;
        ptw[cc * 17u + i] = tw_exact<WIDE>((tile * C + cc) * Q * i, nh - 1u, -2.0f / (float)nh);
    }
    const bool al16 = (IN_MODE == 4) && (((unsigned long long)xb) & 15ull) == 0ull;

    float2 v[16];
    // ---- pass 0 (Ns = 1, radix R0): butterfly a of this thread is column point jb = q + Q*a
    if (EPI == 1 && IN_MODE == 4) {
        // the two samples of every point: am_demod's sum of squares in the reference's roundings into the ring, its (hardware)
        // root into the transform.  All sixteen requests first.
#if defined(__HIPCC__)
        typedef float float4_r8 __attribute__((ext_vector_type(4), aligned(8)));
        float4_r8 raw[16];
#pragma unroll
        for (int a = 0; a < G0; a++)
#pragma unroll
            for (int t = 0; t < R0; t++) {
                const unsigned row = q + Q * (unsigned)a + (unsigned)t * (N1 / (unsigned)R0);
                raw[a * R0 + t] = *reinterpret_cast<const float4_r8 *>((const float2 *)xb + 2 * ((long long)row * N2 + col));
            }
#endif
        float2 *ring = (float2 *)(aux.retain + (long long)b * (2ll * nh));
#pragma unroll
        for (int a = 0; a < G0; a++)
#pragma unroll
            for (int t = 0; t < R0; t++) {
                const int i = a * R0 + t;
                const unsigned row = q + Q * (unsigned)a + (unsigned)t * (N1 / (unsigned)R0);
#if defined(__HIPCC__)
                const float2 x = make_float2(ac4_sumsq_exact(raw[i][0], raw[i][1]), ac4_sumsq_exact(raw[i][2], raw[i][3]));
#else
                const float2 *p = (const float2 *)xb + 2 * ((long long)row * N2 + col);
                const float2 x = make_float2(ac4_sumsq_exact(p[0].x, p[0].y), ac4_sumsq_exact(p[1].x, p[1].y));
#endif
                ring[(long long)row * N2 + col] = x;                    // the replay takes the (correctly rounded) root
                v[i] = make_float2(AC_SQRT(x.x), AC_SQRT(x.y));         // the float32 transform the hardware's, as ever
            }
    } else {
#pragma unroll
        for (int a = 0; a < G0; a++)
#pragma unroll
            for (int t = 0; t < R0; t++) {
                const unsigned row = q + Q * (unsigned)a + (unsigned)t * (N1 / (unsigned)R0);
                v[a * R0 + t] = ac4_load<IN_MODE>(xb, (long long)row * N2 + col, al16, nh, aux.shift);
            }
    }
    // The twiddle between the column and the row transforms, w_nh^(k1 n2), lives here (trip 2 is the one
    // short of VALU time): on trip 1's results and on trip 3's inputs the thread's 16 rows are q + Q i, so
    // the factors are w^(n2 q) * (w^(n2 Q))^i: five accurate evaluations and products of depth <= 4.
    if (LAST) {
        float2 rtw[16];
        if (PTAB) {
            __syncthreads();  // ptw[] complete (the loads above are in flight meanwhile)
            const float2 rbase = tw_exact<WIDE>(col * q, nh - 1u, -2.0f / (float)nh);
            rtw[0] = rbase;
#pragma unroll
            for (int i = 1; i < 16; i++) rtw[i] = cmul(rbase, ptw[c * 17u + (unsigned)i]);
        } else {
            ac4_col_twiddles<WIDE>(rtw, col * Q, col * q, nh);
        }
#pragma unroll
        for (int a = 0; a < G0; a++)
#pragma unroll
            for (int t = 0; t < R0; t++) v[a * R0 + t] = cmul(v[a * R0 + t], rtw[a + G0 * t]);
    }
#pragma unroll
    for (int a = 0; a < G0; a++) dft_reg<R0>(*reinterpret_cast<float2(*)[R0]>(&v[a * R0]));

    if (NP > 1) {
#pragma unroll
        for (int a = 0; a < G0; a++)
#pragma unroll
            for (int u = 0; u < R0; u++) L[((q + Q * (unsigned)a) * (unsigned)R0 + (unsigned)u) * C + c] = v[a * R0 + u];
        __syncthreads();  // tile and twN[] complete
        unsigned Ns = (unsigned)R0;
#pragma unroll
        for (int pass = 1; pass < NP; pass++) {
#pragma unroll
            for (int t = 0; t < 16; t++) v[t] = L[(q + (unsigned)t * Q) * C + c];
            const unsigned k = q & (Ns - 1u);
            const unsigned unit = N1 / (Ns * 16u);
#pragma unroll
            for (int t = 1; t < 16; t++) v[t] = cmul(v[t], twN[((unsigned)t * k * unit) & (N1 - 1u)]);
            dft_reg<16>(v);
            if (pass < NP - 1) {
                __syncthreads();  // every read of this pass done before the tile is overwritten
#pragma unroll
                for (int u = 0; u < 16; u++) L[((q - k) * 16u + k + (unsigned)u * Ns) * C + c] = v[u];
                __syncthreads();
                Ns *= 16u;
            }
        }
    }
    // ---- store: the last pass has Ns*R = N1, so thread q holds rows q + u*(N1/16) (R0 outputs per
    // butterfly when the only pass is pass 0)
    constexpr int RL = (NP > 1) ? 16 : R0;  // radix of the last pass
    constexpr int GL = 16 / RL;
    if (EPI == 2) {
        // superb_bestfit's search (superbandwidth.c:100-116) over what this workgroup computed: the first maximum of |re| and
        // of |im| (two real correlations per transform), lowest index on ties; one (value, index) pair each per workgroup
        float bre = -1.f, bim = -1.f;
        unsigned are = 0x7fffffffu, aim = 0x7fffffffu;
#pragma unroll
        for (int a = 0; a < GL; a++)
#pragma unroll
            for (int u = 0; u < RL; u++) {
                const unsigned row = q + Q * (unsigned)a + (unsigned)u * (N1 / (unsigned)RL);
                const unsigned m = row * N2 + col;
                const float re = fabsf(v[a * RL + u].x), im = fabsf(v[a * RL + u].y);
                if (re > bre || (re == bre && m < are)) { bre = re; are = m; }
                if (im > bim || (im == bim && m < aim)) { bim = im; aim = m; }
            }
        __syncthreads();  // the tile is free: it carries the tree
        float *rv = (float *)L;
        unsigned *ri = (unsigned *)(rv + 2 * NT);
        rv[tid] = bre; rv[NT + tid] = bim; ri[tid] = are; ri[NT + tid] = aim;
        __syncthreads();
        for (unsigned s = NT / 2u; s > 0u; s >>= 1) {
            if (tid < s) {
#pragma unroll
                for (int w = 0; w < 2; w++) {
                    const float ov = rv[w * NT + tid + s];
                    const unsigned oi = ri[w * NT + tid + s];
                    if (ov > rv[w * NT + tid] || (ov == rv[w * NT + tid] && oi < ri[w * NT + tid])) { rv[w * NT + tid] = ov; ri[w * NT + tid] = oi; }
                }
            }
            __syncthreads();
        }
        if (tid < 2u) {
            const unsigned slot = (2u * b + tid) * gridDim.x + tile;
            aux.pval[slot] = rv[tid * NT];
            aux.pidx[slot] = (int)ri[tid * NT];
        }
        return;
    }
    float2 rtw[16];
    if (!LAST) {
        if (PTAB) {  // (NP > 1 here: the passes' barriers lie between the table's writes and these reads)
            const float2 rbase = tw_exact<WIDE>(col * q, nh - 1u, -2.0f / (float)nh);
            rtw[0] = rbase;
#pragma unroll
            for (int i = 1; i < 16; i++) rtw[i] = cmul(rbase, ptw[c * 17u + (unsigned)i]);
        } else {
            ac4_col_twiddles<WIDE>(rtw, col * Q, col * q, nh);
        }
    }
#pragma unroll
    for (int a = 0; a < GL; a++)
#pragma unroll
        for (int u = 0; u < RL; u++) {
            const unsigned row = q + Q * (unsigned)a + (unsigned)u * (N1 / (unsigned)RL);
            const unsigned m = row * N2 + col;
            float2 o = v[a * RL + u];
            if (!LAST) o = cmul(o, rtw[a + GL * u]);
            if (LAST) {
                o.y = -o.y;
                if (keep.on && (int)b != keep.full_b && !((m >= keep.lo0 && m < keep.hi0) || (m >= keep.lo1 && m < keep.hi1) || m == 0u)) continue;
            }
            yb[m] = o;
        }
}

// (Until round 4 this kernel held the body itself.  The restructuring into ac4_cols_body changed its schedule — 102 registers
// instead of 76, the loads further ahead — and a same-box A/B against the old text, kept behind a macro for that purpose and
// removed since, measured the new form 1-2 % FASTER: 0.2387 / 0.2389 against 0.2414 / 0.2430 ms per pass for the two column
// trips, profiles/round5_ab_runs.txt.)
template <int LOGN1, int IN_MODE, bool LAST>
__global__ __launch_bounds__(ColGeom<LOGN1>::NT, 4) void k_ac_cols(const void *__restrict__ xin, long long in_stride,
                                                                float2 *__restrict__ y, unsigned nh, FftKeep keep)
{
    const unsigned b = blockIdx.y;
    const void *xb;
    if (IN_MODE == 3) xb = (const void *)((const float *)xin + (long long)b * in_stride);
    else xb = (const void *)((const float2 *)xin + (long long)b * in_stride);
    const AcColsAux none = {nullptr, 0u, nullptr, nullptr};
    ac4_cols_body<LOGN1, IN_MODE, LAST, 0>(xb, y + (long long)b * nh, nh, keep, b, none);
}


// trip 1 from interleaved IQ that also fills the retention ring (tsdrgpu_autocorr_set_certify mode 1): see AcColsAux.retain
template <int LOGN1>
__global__ __launch_bounds__(ColGeom<LOGN1>::NT, 4) void k_ac_cols_retain(const void *__restrict__ xin, long long in_stride,
                                                                       float2 *__restrict__ y, unsigned nh, float *__restrict__ retain)
{
    const unsigned b = blockIdx.y;
    const FftKeep all = {0, -1, 0u, 0u, 0u, 0u};
    const AcColsAux aux = {retain, 0u, nullptr, nullptr};
    ac4_cols_body<LOGN1, 4, false, 1>((const void *)((const float2 *)xin + (long long)b * in_stride), y + (long long)b * nh, nh, all, b, aux);
}

// ---------------------------------------------------------------------------
// The super-bandwidth stitch on the same three trips (superb_ondataready / superb_bestfit, superbandwidth.c:83-152;
// fft_crosscorrelation, fft.c:69-93).  Four hops of M = N1 * 4096 complex points each.
//
// alignment (hops 1..3 against hop 0, over bn points each): d_h = complex_to_abs_diff(hop_h) is real, so the three
//   correlations c_i = IFFT(conj(D_0) D_i) are real sequences:
//     trip 1  k_sb_cols<.., 5>      column DFTs of the four d_h (abs-diff fused into the load)
//     trip 2  k_sb_rows<XCORR>      row k1 of the four: DFT-4096 each, Q = D_0 conj(D_2) and P = D_0 conj(D_1 - i D_3)
//                                   at the same k in registers, DFT-4096 of both (the inverse's row half): TWO arrays out
//     trip 3  k_sb_cols_argmax      column DFTs of P and Q: F(P) = c_1 + i c_3, F(Q) = c_2; nothing is stored but each
//                                   workgroup's first maxima of |re| and |im|
// stitch: x = IFFT_4M([X_0 X_1 X_2 X_3]), X_h = FFT_M(rot_h(hop_h)) / M.  As the three-trip plan of ONE 4M-point transform
//   with rows of 16384 = 4 x 4096 points (input index K = k1 + N1 (k2 + 4096 h), output n = 16384 q1 + 4 q2 + s):
//     trip 1  k_sb_cols<.., 6>      column DFTs of the four rotated hops (the rotation is the load's index)
//     trip 2  k_sb_rows<STITCH>     row k1 of the four: DFT-4096 each (the forward transforms' row half); then the 16384-point
//                                   row transform of the inverse as radix 4 across the hops at the same k (registers + one
//                                   exchange), the twiddle w_16384^(k2 s), DFT-4096 for each residue s; stored [q2][s]
//     trip 3  k_ac_cols<.., 0, true> with nh = 4M: exactly the autocorrelation's last trip (twiddle, column DFTs, conjugate),
//                                   natural order
//   = 3 trips over 4M points where the pass-per-radix plan made 4 x 3 + 4.
// ---------------------------------------------------------------------------
// (Measured and not kept: the alignment with the hops PAIRED — two real abs-diff signals per complex transform, the four spectra
// out of the pairs Z[k], Z[bn - k] in a row-pair kernel like k_ac_rows; two column trips instead of four, N1/2 x (4 + 4) row
// transforms instead of N1 x (4 + 2).  Offsets identical, but the first trip still reads all four hops (85 -> 68 us) and the row
// kernel gains nothing (48 us either way, 128 registers + 80 bytes of scratch): 0.672 -> 0.665 ms per stitch.  Not worth a second
// form of the row kernel.)
struct SbHops {
    const void *p[4];
};

template <int LOGN1, int IN_MODE>
__global__ __launch_bounds__(ColGeom<LOGN1>::NT, 4) void k_sb_cols(SbHops hops, float2 *__restrict__ y, unsigned nh, const int *__restrict__ off_floats)
{
    const unsigned b = blockIdx.y;
    const FftKeep all = {0, -1, 0u, 0u, 0u, 0u};
    // rotation: the reference's offset is in floats and even (2 * maxlength, superbandwidth.c:118)
    const AcColsAux aux = {nullptr, (IN_MODE == 6 && off_floats) ? ((unsigned)off_floats[b] >> 1) & (nh - 1u) : 0u, nullptr, nullptr};
    ac4_cols_body<LOGN1, IN_MODE, false, 0>(hops.p[b], y + (long long)b * nh, nh, all, b, aux);
}

template <int LOGN1>
__global__ __launch_bounds__(ColGeom<LOGN1>::NT, 4) void k_sb_cols_argmax(const float2 *__restrict__ x, unsigned nh, float *__restrict__ pval, int *__restrict__ pidx)
{
    const unsigned b = blockIdx.y;
    const FftKeep all = {0, -1, 0u, 0u, 0u, 0u};
    const AcColsAux aux = {nullptr, 0u, pval, pidx};
    ac4_cols_body<LOGN1, 0, true, 2>((const void *)(x + (long long)b * nh), nullptr, nh, all, b, aux);
}

// ---------------------------------------------------------------------------
// trip 2: one workgroup of 512 threads per row pair (k1, N1-k1); threads 0..255 own row k1, 256..511
// its mirror.  Workgroup 0 takes the two rows that mirror onto themselves (k1 = 0 and k1 = N1/2).
// DFT-4096 = three radix-16 Stockham passes; thread j holds points j + 256 t before and after.
// The row buffers are padded by one element per 16 so that the stride-16 stores of the first pass
// spread over the banks.
// ---------------------------------------------------------------------------
// Row buffers are padded by one element per 16 (index p lives at p + (p >> 4)), so that the stride-16
// stores of the first pass spread over the banks.  The padded positions are written out as base + constant
// (e.g. pad(j + 256 t) = j + (j >> 4) + 272 t) so that they become immediate offsets of the DS instructions.
#define AC4_ROWBUF (4096 + 256 + 16)

// keeps the compiler from carrying the first transform's address registers through to the second one
#if defined(__HIP_DEVICE_COMPILE__)
#define AC4_LAUNDER(x) asm volatile("" : "+v"(x))
#else
#define AC4_LAUNDER(x) (void)(x)
#endif

// in: v[t] = x[j + 256 t]; out: v[u] = X[j + 256 u].  All threads of the workgroup must call it (barriers);
// Lr is the caller's row buffer, which must not be in use on entry.  Twiddles come from two 256-entry
// LDS tables, tw256[e] = w_256^e and tw4k[e] = w_4096^e (e < 256): no transcendental per point.
__device__ __forceinline__ void ac4_fft4096(float2 (&v)[16], float2 *Lr, unsigned j, const float2 *tw256, const float2 *tw4k)
{
    const unsigned k = j & 15u, jh = j >> 4;
    float2 *const Lj = Lr + (j + jh);  // pad(j + 256 t) = j + (j >> 4) + 272 t
    dft_reg<16>(v);  // pass 0, Ns = 1: pad(16 j + u) = 17 j + u
    {
        float2 *const Lw = Lr + 17u * j;
#pragma unroll
        for (int u = 0; u < 16; u++) Lw[u] = v[u];
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 16; t++) v[t] = Lj[272 * t];
    // pass 1, Ns = 16: w_256^(t k)
#pragma unroll
    for (int t = 1; t < 16; t++) v[t] = cmul(v[t], tw256[(unsigned)t * k]);
    dft_reg<16>(v);
    __syncthreads();
    {   // pad(256 jh + k + 16 u) = 272 jh + k + 17 u
        float2 *const Lw = Lr + (272u * jh + k);
#pragma unroll
        for (int u = 0; u < 16; u++) Lw[17 * u] = v[u];
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 16; t++) v[t] = Lj[272 * t];
    // pass 2, Ns = 256: w_4096^(t j), t < 16, as powers of b = w_4096^j = w_256^(j >> 4) * w_4096^(j & 15) — two table
    // reads and 15 products of depth <= 4 instead of 30 reads and 30 products
    {
        float2 pw[16];
        pw[1] = cmul(tw256[jh], tw4k[k]);
        pw[2] = cmul(pw[1], pw[1]);
        pw[3] = cmul(pw[2], pw[1]);
        pw[4] = cmul(pw[2], pw[2]);
        pw[5] = cmul(pw[4], pw[1]);
        pw[6] = cmul(pw[4], pw[2]);
        pw[7] = cmul(pw[4], pw[3]);
        pw[8] = cmul(pw[4], pw[4]);
#pragma unroll
        for (int t = 9; t < 16; t++) pw[t] = cmul(pw[8], pw[t - 8]);
#pragma unroll
        for (int t = 1; t < 16; t++) v[t] = cmul(v[t], pw[t]);
    }
    dft_reg<16>(v);
}

// (Round 5 built the form the review suggested — one row pair per workgroup of 256 threads, both rows per thread through one
// 39 KB buffer, four workgroups per CU instead of two — and measured it SLOWER in two same-box A/B pairs: 0.146 / 0.168 against
// 0.127 / 0.132 ms per pass; it needs all 128 registers plus 24 bytes of scratch where this one takes 81.  Dropped.)
// (Round 6 built the PERSISTENT form a third time — a grid of two workgroups per CU walking the row pairs, the next pair's sixteen
// points per thread requested before the current pair is transformed — in two addressings: plain pointers (correct, 128 registers +
// 48 bytes of scratch: the group 0.509-0.533 ms per pass against 0.391-0.393 on the same box, the pass 81.0 against 92.1 GS/s) and
// the rows as buffer resources with scalar offsets (fewer address registers, still 12 spilled, and wrong results on the device).
// What a wave saves in waiting it loses twice over in occupancy-neutral register pressure; dropped for good.
// profiles/round6_ab_runs.txt.)
// one row pair: v[t] = this thread's 16 points of its row (loaded by the caller), transformed, split, transformed back and
// stored in place; `buf` / tables as in k_ac_rows.  All 512 threads of the workgroup must call it.
__device__ __forceinline__ void ac4_rows_pair(float2 (&v)[16], float2 *__restrict__ zrow, float2 (*buf)[AC4_ROWBUF], const float2 *tw256,
                                              const float2 *tw4k, unsigned k1, bool selfpair, unsigned half, unsigned j, unsigned nh, unsigned N1)
{
    float2 *Lr = buf[half];
    ac4_fft4096(v, Lr, j, tw256, tw4k);  // v[u] = Z[k1 + N1 (j + 256 u)]
    __syncthreads();
    // ---- split: Z[k] pairs with Z[nh-k], i.e. (k1, k2) <-> (N1-k1, N2-1-k2) — thread j register u of one
    // row with thread 255-j register 15-u of the other —, for k1 = 0: (0, (N2-k2) mod N2)
    float sn0, cs0;
    sincospif(-(float)(k1 + N1 * j) * (1.0f / (float)nh), &sn0, &cs0);  // exp(-i pi k/nh), k = k1 + N1 (j + 256 u)
    const float2 w0 = make_float2(cs0, sn0);
    float2 *const Lj = Lr + (j + (j >> 4));  // own element u sits at Lj[272 u]
    if (!selfpair) {
        // every thread publishes its registers 8..15, computes the 8 pairs of its registers 0..7 (each pair once:
        // both results come out of one evaluation), and hands the partner's halves back through the slots it read
#pragma unroll
        for (int u = 8; u < 16; u++) Lj[272 * u] = v[u];
        __syncthreads();
        const unsigned pb = AC4_ROW - 1u - j;  // partner of k2 = j + 256 u: pb - 256 u, padded like everything else
        float2 *const Lq = buf[1u - half] + (pb + (pb >> 4)) - 272u * 15u;
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const float2 bm = Lq[272 * (15 - u)];
            const float2 wk = u ? cmul(w0, tw256[8u * (unsigned)u]) : w0;  // times exp(-i pi u/16)
            float2 zk, zkm;
            ac_split_pair(v[u], bm, wk, nh, &zk, &zkm);
            v[u] = make_float2(zk.x, -zk.y);  // conjugated input: inverse = conj(FFT(conj(.)))
            Lq[272 * (15 - u)] = make_float2(zkm.x, -zkm.y);
        }
        __syncthreads();
#pragma unroll
        for (int u = 8; u < 16; u++) v[u] = Lj[272 * u];
    } else {
        // the two rows that mirror onto themselves: every element evaluates its own pair
#pragma unroll
        for (int u = 0; u < 16; u++) Lj[272 * u] = v[u];
        __syncthreads();
        // row 0: partner of k2 is 4096 - k2 (k2 = 0 would wrap, but takes the special formula below)
        const unsigned pb = (half == 0u ? AC4_ROW : AC4_ROW - 1u) - j;
        const float2 *const Lq = Lr + (pb + (pb >> 4)) - 272u * 15u;
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const float2 bm = Lq[272 * (15 - u)];
            const float2 wk = u ? cmul(w0, tw256[8u * (unsigned)u]) : w0;
            float2 zk, zkm;
            ac_split_pair(v[u], bm, wk, nh, &zk, &zkm);
            if (u == 0 && k1 == 0u && j == 0u) {
                const float inv_n = 1.0f / (float)(2u * nh);
                const float m0 = fabsf(v[0].x + v[0].y) * inv_n;  // X[0]  = Re Z0 + Im Z0
                const float mh = fabsf(v[0].x - v[0].y) * inv_n;  // X[nh] = Re Z0 - Im Z0
                zk = make_float2(m0 + mh, m0 - mh);
            }
            v[u] = make_float2(zk.x, -zk.y);
        }
    }
    __syncthreads();  // exchange reads done before the buffers are reused
    unsigned j2 = j;
    AC4_LAUNDER(j2);
    ac4_fft4096(v, Lr, j2, tw256, tw4k);
    // row k1 of the inverse's input side at output position n2 = j + 256 u; trip 3 applies w_nh^(n2 k1) on its
    // loads and the conjugation that completes the inverse after its forward column transform
#pragma unroll
    for (int u = 0; u < 16; u++) zrow[j2 + 256u * (unsigned)u] = v[u];
}

__device__ __forceinline__ void ac4_rows_tables(float2 *tw256, float2 *tw4k, unsigned half, unsigned j)
{
    float sn, cs;
    if (half == 0u) {
        sincospif(-(float)j * (1.0f / 128.0f), &sn, &cs);
        tw256[j] = make_float2(cs, sn);
    } else {
        sincospif(-(float)j * (1.0f / 2048.0f), &sn, &cs);
        tw4k[j] = make_float2(cs, sn);
    }
}

__global__ __launch_bounds__(512, 4) void k_ac_rows(float2 *__restrict__ z, unsigned nh)
{
    __shared__ float2 buf[2][AC4_ROWBUF];
    __shared__ float2 tw256[256], tw4k[256];
    const unsigned N1 = nh / AC4_ROW;
    const unsigned tid = threadIdx.x, half = tid >> 8, j = tid & 255u;
    const unsigned wg = blockIdx.x;
    const bool selfpair = wg == 0;
    const unsigned k1 = selfpair ? (half ? N1 / 2u : 0u) : (half ? N1 - wg : wg);
    float2 *zrow = z + (long long)blockIdx.y * nh + (long long)k1 * AC4_ROW;
    float2 v[16];
#pragma unroll
    for (int t = 0; t < 16; t++) v[t] = zrow[j + 256u * (unsigned)t];  // already times w_nh^(k1 n2) (trip 1)
    ac4_rows_tables(tw256, tw4k, half, j);
    __syncthreads();  // tables ready
    ac4_rows_pair(v, zrow, buf, tw256, tw4k, k1, selfpair, half, j, nh, N1);
}

// ---------------------------------------------------------------------------
// trip 2 of the super-bandwidth stitch's two phases (see k_sb_cols above).  One workgroup of 512 threads per row k1 of the
// FOUR hop arrays w[h][k1][.] (trip 1's output, already times w_nh^(k1 n2)): threads 0..255 take hops 0 and 2 one after
// the other, threads 256..511 hops 1 and 3 — thread j of either half ends up holding X_h[k1 + N1 (j + 256 u)], u < 16,
// of its two hops, so everything that combines the hops AT THE SAME k is register arithmetic plus one exchange of 16
// values per thread between the halves.
//   XCORR   (fft_crosscorrelation, fft.c:69-93, for the three pairs (0, i) at once): half 0 forms Q = D0 conj(D2), half 1
//           P = D0 conj(D1 - i D3) (D0 handed over through LDS); each half transforms its one row; out[a][k1][q2], a = half
//   STITCH  (superbandwidth.c:138-146): U_h = conj(X_h) / M; half 0: U0 +- U2, half 1: U1 +- U3; half 0 gets U1 + U3 and
//           forms T0, T2, half 1 gets U0 - U2 and forms T1, T3 (T_s = sum_h (-i)^(h s) U_h); times w_16384^(s k2);
//           DFT-4096 per residue; out[k1][4 q2 + s]
// `scale`: 1/n of the forward transforms (fft.c:167-175), a power of two.
// ---------------------------------------------------------------------------
#define SB_ROWS_XCORR 0
#define SB_ROWS_STITCH 1

template <int MODE>
__global__ __launch_bounds__(512, 4) void k_sb_rows(const float2 *__restrict__ w, float2 *__restrict__ out, unsigned nh, float scale)
{
    __shared__ float2 buf[2][AC4_ROWBUF];
    __shared__ float2 tw256[256], tw4k[256];
    const unsigned tid = threadIdx.x, half = tid >> 8, j = tid & 255u;
    const unsigned k1 = blockIdx.x;
    float2 *Lr = buf[half];
    const float2 *rowA = w + (long long)half * nh + (long long)k1 * AC4_ROW;  // hop `half`
    const float2 *rowB = rowA + 2ll * nh;                                      // hop `half + 2`
    float2 va[16], vb[16];
#pragma unroll
    for (int t = 0; t < 16; t++) va[t] = rowA[j + 256u * (unsigned)t];
#pragma unroll
    for (int t = 0; t < 16; t++) vb[t] = rowB[j + 256u * (unsigned)t];
    {
        float sn, cs;
        if (half == 0u) {
            sincospif(-(float)j * (1.0f / 128.0f), &sn, &cs);
            tw256[j] = make_float2(cs, sn);
        } else {
            sincospif(-(float)j * (1.0f / 2048.0f), &sn, &cs);
            tw4k[j] = make_float2(cs, sn);
        }
    }
    __syncthreads();  // tables ready
    ac4_fft4096(va, Lr, j, tw256, tw4k);  // va[u] = X_a[k1 + N1 (j + 256 u)]
    __syncthreads();
    unsigned j2 = j;
    AC4_LAUNDER(j2);
    ac4_fft4096(vb, Lr, j2, tw256, tw4k);
    __syncthreads();  // both buffers free
    float2 *const Lown = Lr + (j2 + (j2 >> 4));               // own element u sits at Lown[272 u]
    float2 *const Loth = buf[1u - half] + (j2 + (j2 >> 4));    // the same (j, u) of the other half
    if (MODE == SB_ROWS_XCORR) {
        const float s2 = scale * scale;
        if (half == 0u) {
#pragma unroll
            for (int u = 0; u < 16; u++) Lown[272 * u] = va[u];  // D0 for the other half
        } else {
#pragma unroll
            for (int u = 0; u < 16; u++) {  // conj(D1 - i D3)
                const float2 e = cadd_mi(va[u], vb[u]);
                vb[u] = make_float2(e.x * s2, -e.y * s2);
            }
        }
        __syncthreads();
        if (half == 0u) {
#pragma unroll
            for (int u = 0; u < 16; u++) {  // Q = D0 conj(D2)
                const float2 c = make_float2(vb[u].x * s2, -vb[u].y * s2);
                va[u] = cmul(va[u], c);
            }
        } else {
#pragma unroll
            for (int u = 0; u < 16; u++) va[u] = cmul(Loth[272 * u], vb[u]);  // P = D0 conj(D1 - i D3)
        }
        __syncthreads();  // the exchange's reads are done before the buffers carry the next transform
        unsigned j3 = j2;
        AC4_LAUNDER(j3);
        ac4_fft4096(va, Lr, j3, tw256, tw4k);
        float2 *orow = out + (long long)half * nh + (long long)k1 * AC4_ROW;
#pragma unroll
        for (int u = 0; u < 16; u++) orow[j3 + 256u * (unsigned)u] = va[u];
    } else {
        // U = conj(X) * scale; e = Ua + Ub, o = Ua - Ub
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const float2 a = make_float2(va[u].x * scale, -va[u].y * scale), b = make_float2(vb[u].x * scale, -vb[u].y * scale);
            va[u] = cadd(a, b);
            vb[u] = csub(a, b);
        }
        // half 0 publishes its o (U0 - U2) and needs the other's e (U1 + U3); half 1 the other way round
        if (half == 0u) {
#pragma unroll
            for (int u = 0; u < 16; u++) Lown[272 * u] = vb[u];
        } else {
#pragma unroll
            for (int u = 0; u < 16; u++) Lown[272 * u] = va[u];
        }
        __syncthreads();
        if (half == 0u) {
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const float2 be = Loth[272 * u];
                const float2 ae = va[u];
                va[u] = cadd(ae, be);  // T0
                vb[u] = csub(ae, be);  // T2
            }
        } else {
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const float2 ao = Loth[272 * u];
                const float2 bo = vb[u];
                va[u] = cadd_mi(ao, bo);  // T1 = (U0 - U2) - i (U1 - U3)
                vb[u] = csub_mi(ao, bo);  // T3 = (U0 - U2) + i (U1 - U3)
            }
        }
        // residues: va is s = half, vb is s = half + 2; times w_16384^(s (j + 256 u)) = w_16384^(s j) * w_64^(s u)
        const unsigned sa = half, sb = half + 2u;
        {
            float sn, cs;
            if (sa) {
                sincospif(-(float)(sa * j2) * (1.0f / 8192.0f), &sn, &cs);
                const float2 base = make_float2(cs, sn);
                va[0] = cmul(va[0], base);
#pragma unroll
                for (int u = 1; u < 16; u++) va[u] = cmul(va[u], cmul(base, tw256[(4u * sa * (unsigned)u) & 255u]));
            }
            sincospif(-(float)(sb * j2) * (1.0f / 8192.0f), &sn, &cs);
            const float2 base = make_float2(cs, sn);
            vb[0] = cmul(vb[0], base);
#pragma unroll
            for (int u = 1; u < 16; u++) vb[u] = cmul(vb[u], cmul(base, tw256[(4u * sb * (unsigned)u) & 255u]));
        }
        __syncthreads();  // the exchange's reads are done before the buffers carry the next transform
        float2 *orow = out + (long long)k1 * (4ll * AC4_ROW);
        unsigned j3 = j2;
        AC4_LAUNDER(j3);
        ac4_fft4096(va, Lr, j3, tw256, tw4k);  // va[u] = y_sa[q2 = j + 256 u]
        __syncthreads();
        unsigned j4 = j3;
        AC4_LAUNDER(j4);
        ac4_fft4096(vb, Lr, j4, tw256, tw4k);  // vb[u] = y_sb[q2]
        // The four residues of one q2 are 32 contiguous bytes of the output (index 4 q2 + s), and each half holds two of them
        // for all of its 16 points: half 0 keeps the points of even u and hands the odd ones' (y0, y2) over, half 1 keeps the
        // odd ones and hands the even ones' (y1, y3) over — 16 values each way through the (free) row buffers — so that every
        // thread stores 8 x 32 contiguous bytes.  (8-byte stores 32 bytes apart, what the registers allow without the exchange,
        // quarter the efficiency of every store instruction: 254 -> measured below.)
        __syncthreads();  // the transform's last reads of the row buffers are done
        float2 *const Lo = Lr + (j4 + (j4 >> 4));
        float2 *const Lp = buf[1u - half] + (j4 + (j4 >> 4));
        // (two explicit branches on the wave-uniform `half`: every register index a constant)
        if (half == 0u) {
#pragma unroll
            for (int i = 0; i < 8; i++) {  // gives away its odd u
                Lo[272 * (2 * i)] = va[2 * i + 1];
                Lo[272 * (2 * i + 1)] = vb[2 * i + 1];
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; i++) {  // gives away its even u
                Lo[272 * (2 * i)] = va[2 * i];
                Lo[272 * (2 * i + 1)] = vb[2 * i];
            }
        }
        __syncthreads();
#if defined(__HIPCC__)
        typedef float sb_f4 __attribute__((ext_vector_type(4)));
#define SB_STORE4(dst_, y0_, y1_, y2_, y3_)                                        \
    do {                                                                           \
        sb_f4 lo_ = {(y0_).x, (y0_).y, (y1_).x, (y1_).y}, hi_ = {(y2_).x, (y2_).y, (y3_).x, (y3_).y}; \
        *reinterpret_cast<sb_f4 *>(dst_) = lo_;                                    \
        *reinterpret_cast<sb_f4 *>((dst_) + 2) = hi_;                              \
    } while (0)
#else
#define SB_STORE4(dst_, y0_, y1_, y2_, y3_) \
    do { (dst_)[0] = (y0_); (dst_)[1] = (y1_); (dst_)[2] = (y2_); (dst_)[3] = (y3_); } while (0)
#endif
        if (half == 0u) {
#pragma unroll
            for (int i = 0; i < 8; i++) {  // keeps even u: holds s = 0, 2; receives s = 1, 3
                const float2 y1 = Lp[272 * (2 * i)], y3 = Lp[272 * (2 * i + 1)];
                float2 *dst = orow + 4u * (j4 + 256u * (unsigned)(2 * i));
                SB_STORE4(dst, va[2 * i], y1, vb[2 * i], y3);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; i++) {  // keeps odd u: holds s = 1, 3; receives s = 0, 2
                const float2 y0 = Lp[272 * (2 * i)], y2 = Lp[272 * (2 * i + 1)];
                float2 *dst = orow + 4u * (j4 + 256u * (unsigned)(2 * i + 1));
                SB_STORE4(dst, y0, va[2 * i + 1], y2, vb[2 * i + 1]);
            }
        }
#undef SB_STORE4
    }
}
