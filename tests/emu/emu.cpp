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
// the same call evaluated the way k_rs_area_up does: waves of 64 lanes, lane l of a wave that starts at sample
// `base` holds sample base + l - 2 (two ghost lanes hand pnext / fired / tail to the first real one), every value a
// lane takes from its left neighbour goes through the arrays below exactly like the DPP wave shift.
// Returns the number of lanes that needed the slow chain replay.
unsigned emu_resample_chunk_up(const float *in, unsigned size_u, double up, double down, double offset_in,
                               double contrib_in, float *out)
{
    RsGeom g;
    g.r = up / down;
    g.rinv = down / up;
    g.size = size_u;
    g.o = -offset_in * g.r;
    const int size = (int)size_u;
    const int n_out = (int)(((double)size - offset_in) * g.r);
    auto load = [&](int j) { return in[j]; };
    unsigned slow = 0;
    for (int p = 0; p < n_out; p++) out[p] = 0.0f;  // (the kernel zero-fills [pix_in(size), n_out) in its copy-out)
    for (int base = 0; base < size; base += 62) {
        RsUpGeom a[64];
        int pnext[64], pin[64], fired[64];
        double tail[64], val[64];
        float vf[64];
        for (int l = 0; l < 64; l++) {
            const int id = base + l - 2;
            a[l] = rs_up_geom(g, id);
            pnext[l] = (int)a[l].pnext;
            const int idc = id < 0 ? 0 : (id >= size ? size - 1 : id);
            vf[l] = load(idc);
            val[l] = (double)vf[l];
            tail[l] = rs_up_tail(g, a[l], val[l]);
        }
        for (int l = 0; l < 64; l++) {
            const int id = base + l - 2;
            pin[l] = (id <= 0) ? 0 : (l ? pnext[l - 1] : -777);  // lane 0 keeps garbage
            fired[l] = rs_up_fired(a[l], (double)pin[l]) ? 1 : 0;
        }
        for (int l = 2; l < 64; l++) {
            const int id = base + l - 2;
            if (id >= size) break;
            double cin;
            if (id == 0) cin = contrib_in;
            else if (fired[l - 1]) cin = 0.0 + tail[l - 1];
            else if (fired[l]) { cin = rs_contrib_before(g, id, contrib_in, load); slow++; }
            else cin = 0.0;  // unused
            const float first = rs_up_first(a[l], (double)pin[l], cin, val[l]);
            const int cnt = pnext[l] - pin[l];
            for (int c = 0; c < cnt; c++) {
                const int p = pin[l] + c;
                if (p < n_out) out[p] = (c == 0 && fired[l]) ? first : vf[l];
            }
        }
    }
    return slow;
}
}
