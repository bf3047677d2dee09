// resample_math.h — closed-form, per-output-pixel evaluation of the reference's
// sequential area resampler (dsp_resample_process, TempestSDR/src/dsp.c:250-307).
//
// The reference walks the INPUT samples in order, carrying `pix` (next output
// pixel) and `contrib` (unfinished pixel).  Which sample emits which pixel, and
// through which branch, depends only on (r, o, size) — never on the data — so a
// thread can recover, for ONE output pixel p, exactly the f64 expression the
// sequential loop would have evaluated:
//
//   hi_m1(id) = (id*r + o) + r - 1.0        ("idcheck2", dsp.c:283)
//   E(p)      = min{ id : p < hi_m1(id) }   sample during which p is stored
//   pix_in(id)= first pixel not stored before sample id
//             = min{ q >= 0 : q >= hi_m1(id-1) }
//   p is stored by the "finish a straddling pixel" branch (dsp.c:288-292) iff
//   p == pix_in(E) and p < lo(E); then value = contrib + v*((1.0-lo)+p), where
//   contrib is rebuilt by replaying the tail terms (dsp.c:299-302) of the
//   samples since the last sample that took that branch.  Otherwise the pixel
//   lies wholly inside sample E and the value is v (dsp.c:294-297).
//
// Every comparison and arithmetic expression below is the reference's own f64
// expression (compile with -ffp-contract=off), so results — including the
// aligned edge cases where r*size is an integer — are bit-identical.
//
// All functions are host+device so the CPU-only test-suite can exercise the
// same code (tests/emu/) that the HIP kernels run.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define TSDR_HD __host__ __device__ inline
#else
#define TSDR_HD inline
#endif

struct RsChunk {
    long long in_off;   // first input sample of the chunk in the input stream
    long long out_off;  // first output pixel of the chunk in the output stream
    unsigned size;      // input samples in the chunk
    unsigned n_out;     // announced output count, (int)((size-offset)*r)   dsp.c:262
    double o;           // -offset*r  ("offset_sample", dsp.c:272)
};

struct RsGeom {
    double r;     // upsample_by/downsample_by
    double rinv;  // ~1/r, used only to seed searches
    double o;
    unsigned size;
};

// sample indices fit 32 bits (chunk sizes are uint32_t in the reference and far
// below 2^31 in practice): int -> double conversions are single instructions
TSDR_HD double rs_lo(const RsGeom &g, int id) { return (double)id * g.r + g.o; }
TSDR_HD double rs_hi(const RsGeom &g, int id) { return rs_lo(g, id) + g.r; }
TSDR_HD double rs_him1(const RsGeom &g, int id) { return rs_lo(g, id) + g.r - 1.0; }

// smallest integer q >= 0 with q >= x  (q compared as double, like `pid < idcheck2`)
TSDR_HD double rs_first_not_below(double x) { return (x > 0.0) ? ceil(x) : 0.0; }

TSDR_HD double rs_pix_in(const RsGeom &g, int id)
{
    return (id <= 0) ? 0.0 : rs_first_not_below(rs_him1(g, id - 1));
}

// did sample `id` take the straddling-pixel branch (dsp.c:288)?
TSDR_HD bool rs_fired(const RsGeom &g, int id)
{
    const double pin = rs_pix_in(g, id);
    return pin < rs_lo(g, id) && pin < rs_him1(g, id);
}

// E(p) searched from a starting guess; returns g.size when no sample of this chunk stores p
TSDR_HD int rs_owner_from(const RsGeom &g, double p, int id)
{
    const int size = (int)g.size;
    if (id < 0) id = 0;
    if (id > size) id = size;
    while (id > 0 && p < rs_him1(g, id - 1)) id--;
    while (id < size && !(p < rs_him1(g, id))) id++;
    return id;
}

TSDR_HD int rs_owner(const RsGeom &g, double p)
{
    // only a starting point for the exact search below, so a reciprocal multiply is as good as the division
    const double guess = (p + 1.0 - g.r - g.o) * g.rinv;
    int id = (guess < 0.0) ? 0 : ((guess >= 2147483000.0) ? (int)g.size : (int)guess + 1);
    return rs_owner_from(g, p, id);
}

// `contrib` as the reference holds it when sample `id` begins (id may be
// g.size: the value carried out of the chunk).  in(j) returns sample j as float.
// *used_in (optional) tells whether the chunk's incoming contrib took part.
template <class In>
TSDR_HD double rs_contrib_before(const RsGeom &g, int id, double contrib_in, In in,
                                 bool *used_in = nullptr)
{
    int j0 = id - 1;
    while (j0 >= 0 && !rs_fired(g, j0)) j0--;
    if (used_in) *used_in = (j0 < 0);
    double contrib = (j0 >= 0) ? 0.0 : contrib_in;
    for (int j = (j0 >= 0 ? j0 : 0); j < id; j++) {
        const double v = (double)in(j);
        const double pix = rs_pix_in(g, j + 1);
        const double lo = rs_lo(g, j), hi = lo + g.r;
        if (pix < hi && pix > lo)
            contrib += (hi - pix) * v;
        else
            contrib += g.r * v;
    }
    return contrib;
}

// Value of output pixel p of the chunk; returns false when the reference's
// loop never stores it (aligned edge case) — the caller then writes 0.0f.
// `*owner` carries the search position from one pixel to the next (pass a
// negative value for "no hint").
template <class In>
TSDR_HD bool rs_area_pixel(const RsGeom &g, unsigned p, double contrib_in, In in, float *out, int *owner = nullptr)
{
    const double pd = (double)p;
    const int id = (owner && *owner >= 0) ? rs_owner_from(g, pd, *owner) : rs_owner(g, pd);
    if (owner) *owner = id;
    if (id >= (int)g.size) return false;
    const float vf = in(id);
    const double lo = rs_lo(g, id);
    const bool first = (id == 0) ? (p == 0) : (p == 0 || (pd - 1.0) < rs_him1(g, id - 1));
    if (first && pd < lo) {
        const double contrib = rs_contrib_before(g, id, contrib_in, in);
        *out = (float)(contrib + (double)vf * (1.0 - lo + pd));
    } else {
        *out = vf;  // (float)(double)vf
    }
    return true;
}

// A thread's share of the work: pixels [p0, p0+NPIX) of the chunk (clipped to
// [0, n_out)).  Instead of evaluating each pixel on its own, jump into the
// reference's loop at the sample that stores the first pixel — its state there
// (`pid`, `contrib`) is known in closed form — and replay the loop body
// (dsp.c:280-303) literally over the handful of samples that touch the group.
// lo/hi/him1 and the demodulated sample are computed once per sample.
// put(k, value) receives pixel p0+k; pixels the reference never stores are not
// reported (the caller pre-fills zeros).  `ipix` shadows the double `pix` so
// that the group-window tests are integer compares.
template <int NPIX, class In, class Put>
TSDR_HD void rs_area_group(const RsGeom &g, int p0, int n_out, double contrib_in, In in, Put put)
{
    const int pfirst = p0 < 0 ? 0 : p0;
    const int pend = (p0 + NPIX < n_out) ? (p0 + NPIX) : n_out;
    if (pfirst >= pend) return;
    const int size = (int)g.size;
    int id = rs_owner(g, (double)pfirst);
    if (id >= size) return;
    double pix = rs_pix_in(g, id);
    int ipix = (int)pix;
    double contrib = rs_contrib_before(g, id, contrib_in, in);
    for (; id < size; id++) {
        const double lo = (double)id * g.r + g.o;
        const double hi = lo + g.r;
        const double him1 = lo + g.r - 1.0;
        const float vf = in(id);
        const double val = (double)vf;
        if (pix < lo && pix < him1) {
            if (ipix >= pfirst) put(ipix - p0, (float)(contrib + val * (1.0 - lo + pix)));
            contrib = 0;
            pix += 1.0;
            if (++ipix >= pend) return;
        }
        while (pix < him1) {
            if (ipix >= pfirst) put(ipix - p0, vf);
            pix += 1.0;
            if (++ipix >= pend) return;
        }
        if (pix < hi && pix > lo)
            contrib += (hi - pix) * val;
        else
            contrib += g.r * val;
    }
}

// ---------------------------------------------------------------------------
// Sample-parallel form (k_rs_area_up).  One lane per INPUT sample: sample id
// stores exactly the pixels [pix_in(id), pix_in(id+1)) — branch A of the loop
// body (dsp.c:288-292) stores the first of them when it fires, branch B
// (dsp.c:294-297) the rest as copies of the sample — and then adds one tail term
// to contrib (dsp.c:299-302).  Everything but contrib depends on id alone, and
// contrib on entry is 0.0 + tail(id-1) whenever sample id-1 fired branch A (with
// r >= 1 practically always), so a lane needs three values from its left
// neighbour: pix_in(id) (= the neighbour's pnext), whether the neighbour fired,
// and its tail term.  When the neighbour did not fire, rs_contrib_before replays
// the chain.  The expressions are the loop body's own.
// ---------------------------------------------------------------------------
struct RsUpGeom {
    double lo, hi, him1;
    double pnext;  // pix when the sample is done = pix_in(id+1)
};

TSDR_HD RsUpGeom rs_up_geom(const RsGeom &g, int id)
{
    RsUpGeom a;
    a.lo = (double)id * g.r + g.o;
    a.hi = a.lo + g.r;
    a.him1 = a.lo + g.r - 1.0;
    a.pnext = rs_first_not_below(a.him1);
    return a;
}
// branch A taken?  pin = pix on entry
TSDR_HD bool rs_up_fired(const RsUpGeom &a, double pin) { return pin < a.lo && pin < a.him1; }
// the term dsp.c:299-302 adds after the stores (pix == pnext there)
TSDR_HD double rs_up_tail(const RsGeom &g, const RsUpGeom &a, double val)
{
    return (a.pnext < a.hi && a.pnext > a.lo) ? (a.hi - a.pnext) * val : g.r * val;
}
// the straddling pixel's value (dsp.c:289)
TSDR_HD float rs_up_first(const RsUpGeom &a, double pin, double contrib, double val)
{
    return (float)(contrib + val * (1.0 - a.lo + pin));
}

// nearest-neighbour branch, dsp.c:274-276
TSDR_HD long long rs_nearest_src(unsigned size, unsigned n_out, unsigned p)
{
    return (long long)(((uint64_t)size * p) / n_out);
}
