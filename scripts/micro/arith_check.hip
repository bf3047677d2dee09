// Device check behind DESIGN.md "Instruction diets": on this GPU, for the hard cases of both sequences,
//   (a) the compiler's f32 `/` gives the correctly rounded quotient (the host's n / d),
//   (b) the bare numerator half with the divisor's half prepared once (NormDiv, tsdrgpu_frame.hip) gives the same bits as `/`,
//   (c) sqrtf gives the correctly rounded root, and the bare two-sided correction (demod1, tsdrgpu_core.hip) the same bits.
// Hard cases for (a)/(b): divisors whose mantissa is all ones (the quotient of a power of two then lies 2^-48 beside a rounding
// midpoint, and an approximate reciprocal one ulp low lands exactly on it — tests/test_emu_arith.py), beside random ones.
// Prints "<n> mismatches"; exit status 0 only when all three counts are zero.  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

struct NormDiv { float d, r; };
__device__ __forceinline__ NormDiv nd_setup(float span)
{
    NormDiv nd;
    nd.d = span;
    const float r0 = __builtin_amdgcn_rcpf(span);
    const float e = __builtin_fmaf(-span, r0, 1.0f);
    nd.r = __builtin_fmaf(e, r0, r0);
    return nd;
}
__device__ __forceinline__ float nd_div(const NormDiv &nd, float n)
{
    float q = n * nd.r;
    q = __builtin_fmaf(__builtin_fmaf(-nd.d, q, n), nd.r, q);
    return __builtin_fmaf(__builtin_fmaf(-nd.d, q, n), nd.r, q);
}
__global__ void k_div(const float *n, const float *d, float *q_plain, float *q_bare, int count)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    q_plain[i] = n[i] / d[i];
    q_bare[i] = nd_div(nd_setup(d[i]), n[i]);
}
__global__ void k_sqrt(const float *x, float *s_plain, float *s_bare, int count)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const float v = x[i];
    s_plain[i] = sqrtf(v);
    float s = __builtin_amdgcn_sqrtf(v);
    const float sm = __uint_as_float(__float_as_uint(s) - 1u), sp = __uint_as_float(__float_as_uint(s) + 1u);
    const float rm = __builtin_fmaf(-sm, s, v), rp = __builtin_fmaf(-sp, s, v);
    s = (rm <= 0.0f) ? sm : s;
    s = (rp > 0.0f) ? sp : s;
    s_bare[i] = s;
}
static uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static float fl(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

int main()
{
    std::vector<float> n, d;
    uint64_t z = 12345;
    auto rnd = [&]() { z = z * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(z >> 32); };
    // all-ones divisors in every binade of the guard x numerators: powers of two, their neighbours, random
    for (int e = -20; e <= 20; e++) {
        const float dd = fl(((uint32_t)(127 + e - 1) << 23) | 0x7fffffu);
        for (int k = -44; k <= 11; k++) {
            const float p = ldexpf(1.0f, k);
            for (int u = -2; u <= 2; u++) { n.push_back(fl(bits(p) + (uint32_t)u)); d.push_back(dd); }
            n.push_back(-p); d.push_back(dd);
        }
        for (int j = 0; j < 2000; j++) { n.push_back(ldexpf((float)(rnd() & 0xffffff) / 16777216.0f + 0.5f, (int)(rnd() % 55) - 44)); d.push_back(dd); }
    }
    // random divisors and numerators inside the guard
    for (int j = 0; j < 4000000; j++) {
        d.push_back(ldexpf((float)(rnd() & 0xffffff) / 16777216.0f + 0.5f, (int)(rnd() % 41) - 19));
        const float m = ldexpf((float)(rnd() & 0xffffff) / 16777216.0f + 0.5f, (int)(rnd() % 55) - 43);
        n.push_back((rnd() & 1) ? m : -m);
    }
    const int count = (int)n.size();
    float *dn, *dd_, *dq, *db;
    hipMalloc(&dn, 4 * count); hipMalloc(&dd_, 4 * count); hipMalloc(&dq, 4 * count); hipMalloc(&db, 4 * count);
    hipMemcpy(dn, n.data(), 4 * count, hipMemcpyHostToDevice);
    hipMemcpy(dd_, d.data(), 4 * count, hipMemcpyHostToDevice);
    k_div<<<(count + 255) / 256, 256>>>(dn, dd_, dq, db, count);
    std::vector<float> q(count), b(count);
    hipMemcpy(q.data(), dq, 4 * count, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), db, 4 * count, hipMemcpyDeviceToHost);
    long bad_plain = 0, bad_bare = 0;
    for (int i = 0; i < count; i++) {
        const float want = n[i] / d[i];
        if (bits(q[i]) != bits(want)) { if (bad_plain < 5) printf("plain  n=%08x d=%08x got %08x want %08x\n", bits(n[i]), bits(d[i]), bits(q[i]), bits(want)); bad_plain++; }
        if (bits(b[i]) != bits(q[i])) { if (bad_bare < 5) printf("bare   n=%08x d=%08x got %08x plain %08x\n", bits(n[i]), bits(d[i]), bits(b[i]), bits(q[i])); bad_bare++; }
    }
    printf("division: %d cases, plain vs host: %ld mismatches, bare vs plain: %ld mismatches\n", count, bad_plain, bad_bare);
    // square roots: every 257th float of [2^-96, inf), all of one even and one odd binade
    std::vector<float> x;
    for (uint64_t u = 0x0f800000u; u < 0x7f800000u; u += 257) x.push_back(fl((uint32_t)u));
    for (uint32_t u = 0x3f800000u; u < 0x40800000u; u++) x.push_back(fl(u));
    const int cs = (int)x.size();
    float *dx, *ds, *dsb;
    hipMalloc(&dx, 4ull * cs); hipMalloc(&ds, 4ull * cs); hipMalloc(&dsb, 4ull * cs);
    hipMemcpy(dx, x.data(), 4ull * cs, hipMemcpyHostToDevice);
    k_sqrt<<<(cs + 255) / 256, 256>>>(dx, ds, dsb, cs);
    std::vector<float> s(cs), sb(cs);
    hipMemcpy(s.data(), ds, 4ull * cs, hipMemcpyDeviceToHost);
    hipMemcpy(sb.data(), dsb, 4ull * cs, hipMemcpyDeviceToHost);
    long bad_s = 0, bad_sb = 0;
    for (int i = 0; i < cs; i++) {
        const float want = sqrtf(x[i]);
        if (bits(s[i]) != bits(want)) bad_s++;
        if (bits(sb[i]) != bits(s[i])) { if (bad_sb < 5) printf("sqrt bare x=%08x got %08x plain %08x\n", bits(x[i]), bits(sb[i]), bits(s[i])); bad_sb++; }
    }
    printf("square root: %d cases, sqrtf vs host: %ld mismatches, bare vs sqrtf: %ld mismatches\n", cs, bad_s, bad_sb);
    const long total = bad_plain + bad_bare + bad_s + bad_sb;
    printf("%ld mismatches\n", total);
    return total ? 1 : 0;
}
