// tests/emu/emu_fft.cpp — host build of the three-trip autocorrelation kernels
// (tempestsdr_amd/csrc/fft4step.h): the very code the GPU runs, executed by OS threads, so that the
// CPU-only suite can check the index and twiddle algebra against numpy before a GPU sees it.
// TEST INFRASTRUCTURE ONLY: never loaded by the product.
#include "hipemu.h"

#include "../../tempestsdr_amd/csrc/fft4step.h"

template <int LOGN1>
static void run_plan(const float *in, int in_is_iq, long long stride, int cnt, unsigned nh, float2 *work, float2 *out, FftKeep keep)
{
    typedef ColGeom<LOGN1> G;
    const unsigned N2 = nh >> LOGN1;
    const FftKeep all = {0, -1, 0u, 0u, 0u, 0u};
    if (in_is_iq)
        emu_launch(emu_dim3(N2 / G::C, cnt), G::NT, [&]() { k_ac_cols<LOGN1, 4, false>(in, stride, work, nh, all); });
    else
        emu_launch(emu_dim3(N2 / G::C, cnt), G::NT, [&]() { k_ac_cols<LOGN1, 3, false>(in, stride, work, nh, all); });
    emu_launch(emu_dim3((1u << LOGN1) / 2, cnt), 512, [&]() { k_ac_rows(work, nh); });
    emu_launch(emu_dim3(N2 / G::C, cnt), G::NT, [&]() { k_ac_cols<LOGN1, 0, true>(work, (long long)nh, out, nh, keep); });
}

extern "C" int emu_autocorr4(const float *in, int in_is_iq, long long stride, int cnt, unsigned nh, float *work, float *out,
                             int keep_on, int full_b, unsigned lo0, unsigned hi0, unsigned lo1, unsigned hi1)
{
    FftKeep keep = {keep_on, full_b, lo0, hi0, lo1, hi1};
    unsigned logn1 = 0;
    while ((4096u << logn1) < nh) logn1++;
    if ((4096u << logn1) != nh) return -1;
    float2 *w = (float2 *)work, *o = (float2 *)out;
    switch (logn1) {
        case 4: run_plan<4>(in, in_is_iq, stride, cnt, nh, w, o, keep); break;
        case 5: run_plan<5>(in, in_is_iq, stride, cnt, nh, w, o, keep); break;
        case 6: run_plan<6>(in, in_is_iq, stride, cnt, nh, w, o, keep); break;
        case 7: run_plan<7>(in, in_is_iq, stride, cnt, nh, w, o, keep); break;
        case 8: run_plan<8>(in, in_is_iq, stride, cnt, nh, w, o, keep); break;
        case 9: run_plan<9>(in, in_is_iq, stride, cnt, nh, w, o, keep); break;
        case 10: run_plan<10>(in, in_is_iq, stride, cnt, nh, w, o, keep); break;
        default: return -1;
    }
    return 0;
}

// ---- trip 1 with the retention side store (tsdrgpu_autocorr_set_certify mode 1) -------------------------------------
template <int LOGN1>
static void run_plan_retain(const float *in, long long stride, int cnt, unsigned nh, float2 *work, float2 *out, float *retain)
{
    typedef ColGeom<LOGN1> G;
    const unsigned N2 = nh >> LOGN1;
    const FftKeep all = {0, -1, 0u, 0u, 0u, 0u};
    emu_launch(emu_dim3(N2 / G::C, cnt), G::NT, [&]() { k_ac_cols_retain<LOGN1>(in, stride, work, nh, retain); });
    emu_launch(emu_dim3((1u << LOGN1) / 2, cnt), 512, [&]() { k_ac_rows(work, nh); });
    emu_launch(emu_dim3(N2 / G::C, cnt), G::NT, [&]() { k_ac_cols<LOGN1, 0, true>(work, (long long)nh, out, nh, all); });
}

extern "C" int emu_autocorr4_retain(const float *in, long long stride, int cnt, unsigned nh, float *work, float *out, float *retain)
{
    unsigned logn1 = 0;
    while ((4096u << logn1) < nh) logn1++;
    if ((4096u << logn1) != nh) return -1;
    float2 *w = (float2 *)work, *o = (float2 *)out;
    switch (logn1) {
        case 4: run_plan_retain<4>(in, stride, cnt, nh, w, o, retain); break;
        case 5: run_plan_retain<5>(in, stride, cnt, nh, w, o, retain); break;
        case 6: run_plan_retain<6>(in, stride, cnt, nh, w, o, retain); break;
        default: return -1;
    }
    return 0;
}

// ---- the super-bandwidth stitch's two phases (k_sb_cols / k_sb_rows / k_sb_cols_argmax / k_ac_cols) -----------------
template <int LOGN1>
static int run_sb_xcorr(SbHops hops, unsigned bn, float2 *w, float2 *v, float *pval, int *pidx)
{
    typedef ColGeom<LOGN1> G;
    const unsigned N2 = bn >> LOGN1;
    emu_launch(emu_dim3(N2 / G::C, 4), G::NT, [&]() { k_sb_cols<LOGN1, 5>(hops, w, bn, nullptr); });
    emu_launch(emu_dim3(1u << LOGN1), 512, [&]() { k_sb_rows<SB_ROWS_XCORR>(w, v, bn, 1.0f / (float)bn); });
    emu_launch(emu_dim3(N2 / G::C, 2), G::NT, [&]() { k_sb_cols_argmax<LOGN1>(v, bn, pval, pidx); });
    return (int)(N2 / G::C);
}

template <int LOGN1>
static void run_sb_stitch(SbHops hops, unsigned per, const int *off, float2 *w, float2 *v, float2 *out)
{
    typedef ColGeom<LOGN1> G;
    const unsigned N2 = per >> LOGN1;
    const FftKeep all = {0, -1, 0u, 0u, 0u, 0u};
    emu_launch(emu_dim3(N2 / G::C, 4), G::NT, [&]() { k_sb_cols<LOGN1, 6>(hops, w, per, off); });
    emu_launch(emu_dim3(1u << LOGN1), 512, [&]() { k_sb_rows<SB_ROWS_STITCH>(w, v, per, 1.0f / (float)per); });
    emu_launch(emu_dim3(4u * N2 / G::C, 1), G::NT, [&]() { k_ac_cols<LOGN1, 0, true>(v, 4ll * per, out, 4u * per, all); });
}

// returns the number of partials per slot (slots: array 0 |re|, array 0 |im|, array 1 |re|, array 1 |im|), or -1
extern "C" int emu_sb_xcorr(const float *h0, const float *h1, const float *h2, const float *h3, unsigned bn, float *work, float *v,
                            float *pval, int *pidx)
{
    unsigned logn1 = 0;
    while ((4096u << logn1) < bn) logn1++;
    if ((4096u << logn1) != bn) return -1;
    SbHops hops = {{h0, h1, h2, h3}};
    switch (logn1) {
        case 4: return run_sb_xcorr<4>(hops, bn, (float2 *)work, (float2 *)v, pval, pidx);
        case 5: return run_sb_xcorr<5>(hops, bn, (float2 *)work, (float2 *)v, pval, pidx);
        case 6: return run_sb_xcorr<6>(hops, bn, (float2 *)work, (float2 *)v, pval, pidx);
        default: return -1;
    }
}

extern "C" int emu_sb_stitch(const float *h0, const float *h1, const float *h2, const float *h3, unsigned per, const int *off_floats,
                             float *work, float *v, float *out)
{
    unsigned logn1 = 0;
    while ((4096u << logn1) < per) logn1++;
    if ((4096u << logn1) != per) return -1;
    SbHops hops = {{h0, h1, h2, h3}};
    switch (logn1) {
        case 4: run_sb_stitch<4>(hops, per, off_floats, (float2 *)work, (float2 *)v, (float2 *)out); break;
        case 5: run_sb_stitch<5>(hops, per, off_floats, (float2 *)work, (float2 *)v, (float2 *)out); break;
        case 6: run_sb_stitch<6>(hops, per, off_floats, (float2 *)work, (float2 *)v, (float2 *)out); break;
        default: return -1;
    }
    return 0;
}
