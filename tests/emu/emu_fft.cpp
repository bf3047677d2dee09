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
