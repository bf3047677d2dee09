// CPU restatement of two instruction sequences the HIP kernels run bare (tests/test_emu_arith.py): the numerator's half of an
// IEEE f32 division with the divisor's half prepared once (NormDiv, tsdrgpu_frame.hip) and the two-sided correction of an
// approximate square root (demod1, tsdrgpu_core.hip).  Both claim "the same bits as / and sqrtf" inside their guards, whatever
// the hardware's 1-ulp approximations (v_rcp_f32, v_sqrt_f32) return: so the starting approximation is an ARGUMENT here and the
// tests perturb it by an ulp either way.  Compile with -ffp-contract=off; fmaf is the C library's (single rounding).
#include <cmath>
#include <cstdint>
#include <cstring>

static inline float as_f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline uint32_t as_u(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

extern "C" {
// norm_div_setup + norm_div with r0 = the reciprocal approximation of d displaced by `ulps`
float emu_norm_div(float n, float d, int ulps)
{
    const float r0 = as_f(as_u(1.0f / d) + (uint32_t)ulps);
    const float e = fmaf(-d, r0, 1.0f);
    const float r = fmaf(e, r0, r0);
    float q = n * r;
    q = fmaf(fmaf(-d, q, n), r, q);
    return fmaf(fmaf(-d, q, n), r, q);
}
// mismatches of emu_norm_div against n / d over `count` numerators n = v - lastmin, v in [-250, 250] hashed from `seed`
long emu_norm_div_sweep(float lastmin, float d, int ulps, long count, uint64_t seed, float *bad_n)
{
    long bad = 0;
    uint64_t z = seed;
    for (long i = 0; i < count; i++) {
        z = z * 6364136223846793005ull + 1442695040888963407ull;
        // a third of the pixels close to the minimum (tiny numerators), the rest anywhere in [-250, 250]
        const uint32_t h = (uint32_t)(z >> 33);
        float v;
        if (h % 3u == 0u) v = as_f(as_u(lastmin) + (h >> 8) % 4096u);            // within 4096 ulps above lastmin's magnitude
        else v = ((float)(h & 0xffffffu) / 16777216.0f - 0.5f) * 500.0f;
        if (!(std::fabs(v) <= 250.0f)) continue;
        const float n = v - lastmin;
        const float want = n / d, got = emu_norm_div(n, d, ulps);
        if (as_u(want) != as_u(got)) { if (!bad && bad_n) *bad_n = n; bad++; }
    }
    return bad;
}
// demod1's bare path: s0 = the square root approximation displaced by `ulps`
float emu_sqrt_fix(float x, int ulps)
{
    float s = as_f(as_u(std::sqrt(x)) + (uint32_t)ulps);
    const float sm = as_f(as_u(s) - 1u), sp = as_f(as_u(s) + 1u);
    const float rm = fmaf(-sm, s, x), rp = fmaf(-sp, s, x);
    s = (rm <= 0.0f) ? sm : s;
    s = (rp > 0.0f) ? sp : s;
    return s;
}
long emu_sqrt_fix_sweep(uint32_t lo_bits, uint32_t hi_bits, uint32_t step, int ulps, uint32_t *bad_x)
{
    long bad = 0;
    for (uint64_t b = lo_bits; b < hi_bits; b += step) {
        const float x = as_f((uint32_t)b);
        if (as_u(std::sqrt(x)) != as_u(emu_sqrt_fix(x, ulps))) { if (!bad && bad_x) *bad_x = (uint32_t)b; bad++; }
    }
    return bad;
}
}
