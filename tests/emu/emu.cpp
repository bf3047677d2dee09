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
    auto load = [&](int j) { return in[j]; };
    for (unsigned p = 0; p < n_out; p += 4) {  // groups of four share the owner search, like the kernel
        int owner = -1;
        for (unsigned k = p; k < p + 4 && k < n_out; k++) {
            float v;
            out[k] = rs_area_pixel(g, k, contrib_in, load, &v, &owner) ? v : 0.0f;
        }
    }
    *contrib_out = rs_contrib_before(g, (int)size, contrib_in, load);
    *offset_out = offset_in + (n_out * (down / up) - size);
    return n_out;
}
}
