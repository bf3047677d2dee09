// tests/emu/emu.cpp — host build of the per-thread math headers that the HIP
// kernels use (tempestsdr_amd/csrc/*_math.h), so the CPU-only test-suite can
// check the closed forms against the oracle before anything runs on a GPU.
// TEST INFRASTRUCTURE ONLY: never loaded by the product.
#include "../../tempestsdr_amd/csrc/resample_math.h"

extern "C" {

// one dsp_resample_process call, evaluated pixel-by-pixel like the kernel does
unsigned emu_resample_chunk(const float *in, unsigned size, double up, double down, double offset_in,
                            double contrib_in, float *out, double *contrib_out, double *offset_out)
{
    RsGeom g;
    g.r = up / down;
    g.rinv = down / up;
    g.size = size;
    g.o = -offset_in * g.r;
    const unsigned n_out = (unsigned)(int)(((double)size - offset_in) * g.r);
    auto load = [&](int j) { return in[j]; };
    // like the kernel: groups of 8 pixels, the first group misaligned by `mis` (here: size % 4)
    const int mis = (int)(size & 3);
    for (int p0 = -mis; p0 < (int)n_out; p0 += 8) {
        float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        rs_area_group<8>(g, p0, (int)n_out, contrib_in, load, [&](int k, float val) { v[k] = val; });
        for (int k = 0; k < 8; k++)
            if (p0 + k >= 0 && p0 + k < (int)n_out) {
                out[p0 + k] = v[k];
                // the per-pixel form must agree with the group form
                float s;
                const float want = rs_area_pixel(g, (unsigned)(p0 + k), contrib_in, load, &s) ? s : 0.0f;
                if (want != v[k]) out[p0 + k] = -12345.0f;
            }
    }
    *contrib_out = rs_contrib_before(g, (int)size, contrib_in, load);
    *offset_out = offset_in + (n_out * (down / up) - size);
    return n_out;
}
}
