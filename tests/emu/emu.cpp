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
    g.size = size;
    g.o = -offset_in * g.r;
    const unsigned n_out = (unsigned)(int)(((double)size - offset_in) * g.r);
    auto load = [&](long long j) { return in[j]; };
    for (unsigned p = 0; p < n_out; p++) {
        float v;
        out[p] = rs_area_pixel(g, p, contrib_in, load, &v) ? v : 0.0f;
    }
    *contrib_out = rs_contrib_before(g, (long long)size, contrib_in, load);
    *offset_out = offset_in + (n_out * (down / up) - size);
    return n_out;
}
}
