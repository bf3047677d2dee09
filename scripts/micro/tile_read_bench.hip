// tile_read_bench.hip — how fast can 60 frames of 2962 x 1125 floats (800 MB, cold) be READ by 256-thread workgroups
// that each take 32 KB (32 dword loads per lane, all issued before the first use, like k_frame_stats), as a function
// of the SHAPE of those 32 KB: 256 columns x 32 rows (k_frame_stats' tile), 512 x 16, 1024 x 8, 2048 x 4, or 32 KB of
// consecutive addresses.  Same instruction mix in all shapes; only the addresses differ.
// Build: hipcc --offload-arch=gfx950 -O2 -o tile_read_bench tile_read_bench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int TW, int TH, bool XCD>
__global__ __launch_bounds__(256) void k_tile(const float *__restrict__ frames, int W, int H, int tiles_x, int tiles_y, float *out)
{
    const unsigned total = gridDim.x, l = blockIdx.x;
    const unsigned logical = (XCD && total % 8u == 0u) ? (l % 8u) * (total / 8u) + l / 8u : l;
    const int tx = logical % tiles_x, ty = (logical / tiles_x) % tiles_y, f = logical / (tiles_x * tiles_y);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float *src = frames + (long long)f * W * H;
    constexpr int R = TH / 4, C = TW / 64;
    float v[R][C];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int y = ty * TH + wave + 4 * r;
        const float *row = src + (long long)(y < H ? y : 0) * W;
#pragma unroll
        for (int j = 0; j < C; j++) {
            const int x = tx * TW + lane + 64 * j;
            v[r][j] = row[x < W ? x : W - 1];
        }
    }
    float m = 0.f;
#pragma unroll
    for (int r = 0; r < R; r++)
#pragma unroll
        for (int j = 0; j < C; j++) m = fmaxf(m, v[r][j]);
    if (m == 12345.678f) out[l] = m;
}

// the 256 x 32 tile with (a) a cap on the workgroups per CU through LDS, (b) k_frame_stats' own reductions and stores
#include "../../tempestsdr_amd/csrc/wave_reduce.h"
template <int LDS_KB, bool REDUCE, int MODE>  // MODE: 0 all stores, 1 no stores, 2 no row stores, 3 no column stores, 4 row sums through LDS (one 128-byte store per tile)
__global__ __launch_bounds__(256) void k_tile_like_stats(const float *__restrict__ frames, int W, int H, int tiles_x, int tiles_y, float *out,
                                                         float *__restrict__ colp, float *__restrict__ rowp)
{
    constexpr int TW = 256, TH = 32;
    __shared__ float pad[LDS_KB * 256];
    const unsigned total = gridDim.x, l = blockIdx.x;
    const unsigned logical = (total % 8u == 0u) ? (l % 8u) * (total / 8u) + l / 8u : l;
    const int tx = logical % tiles_x, ty = (logical / tiles_x) % tiles_y, f = logical / (tiles_x * tiles_y);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float *src = frames + (long long)f * W * H;
    constexpr int R = TH / 4, C = TW / 64;
    float v[R][C];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int y = ty * TH + wave + 4 * r;
        const float *row = src + (long long)(y < H ? y : 0) * W;
#pragma unroll
        for (int j = 0; j < C; j++) {
            const int x = tx * TW + lane + 64 * j;
            v[r][j] = row[x < W ? x : W - 1];
        }
    }
    float m = 0.f;
    if (!REDUCE) {
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int j = 0; j < C; j++) m = fmaxf(m, v[r][j]);
        if (m == 12345.678f) { out[l] = m; pad[threadIdx.x] = m; }
        return;
    }
    float cns[4] = {0, 0, 0, 0};
    float lo = INFINITY, hi = -INFINITY;
#pragma unroll
    for (int r = 0; r < R; r++) {
        float rns = 0.f;
#pragma unroll
        for (int j = 0; j < C; j++) {
            cns[j] += v[r][j];
            rns += v[r][j];
            lo = fminf(lo, v[r][j]);
            hi = fmaxf(hi, v[r][j]);
        }
        rns = wave_sum(rns);
        const int y = ty * TH + wave + 4 * r;
        if (MODE == 0 || MODE == 3) {
            if (lane == 0 && y < H) rowp[((long long)(f * tiles_x + tx) * 3) * H + y] = rns;
        } else if (MODE == 4) {
            if (lane == 0) pad[1040 + wave + 4 * r] = rns;
        } else if (rns == 12345.678f) out[l] = rns;
    }
#pragma unroll
    for (int j = 0; j < 4; j++) pad[wave * 256 + lane + 64 * j] = cns[j];
    lo = wave_min(lo);
    hi = wave_max(hi);
    if (lane == 0) { pad[1024 + wave] = lo; pad[1028 + wave] = hi; }
    __syncthreads();
    const int x = tx * TW + threadIdx.x;
    const float cs = pad[threadIdx.x] + pad[256 + threadIdx.x] + pad[512 + threadIdx.x] + pad[768 + threadIdx.x];
    if (MODE == 0 || MODE == 2 || MODE == 4) {
        if (x < W) colp[((long long)(f * tiles_y + ty) * 3) * W + x] = cs;
    } else if (MODE == 5) {
        if (x < W) __builtin_nontemporal_store(cs, &colp[((long long)(f * tiles_y + ty) * 3) * W + x]);
    } else if (MODE == 6) {  // 1 KB-aligned blocks per tile
        colp[((long long)(f * tiles_y + ty) * tiles_x + tx) * 256 + threadIdx.x] = cs;
    } else if (MODE == 7) {
        __builtin_nontemporal_store(cs, &colp[((long long)(f * tiles_y + ty) * tiles_x + tx) * 256 + threadIdx.x]);
    } else if (MODE == 8) {  // a quarter of the bytes
        if (threadIdx.x < 64) colp[((long long)(f * tiles_y + ty) * tiles_x + tx) * 64 + threadIdx.x] = cs;
    } else if (MODE == 9) {  // four times the bytes
        for (int q = 0; q < 4; q++) colp[(((long long)(f * tiles_y + ty) * tiles_x + tx) * 4 + q) * 256 + threadIdx.x] = cs;
    } else if (MODE == 10) {  // the same stores, but into 1 MB that stays in the L2s
        colp[(l & 1023u) * 256 + threadIdx.x] = cs;
    } else if (cs == 12345.678f) out[l] = cs;
    if (MODE == 4 && threadIdx.x < 32 && ty * TH + (int)threadIdx.x < H)
        rowp[((long long)(f * tiles_x + tx) * 3) * H + ty * TH + threadIdx.x] = pad[1040 + threadIdx.x];
    if (MODE == 1) {
        if (pad[1024] == 12345.678f) out[l] = pad[1028];
    } else if (threadIdx.x == 0) {
        out[2 * l] = fminf(fminf(pad[1024], pad[1025]), fminf(pad[1026], pad[1027]));
        out[2 * l + 1] = fmaxf(fmaxf(pad[1028], pad[1029]), fmaxf(pad[1030], pad[1031]));
    }
}

// 32 KB of consecutive addresses per workgroup (the frames as one array)
template <bool XCD>
__global__ __launch_bounds__(256) void k_linear(const float *__restrict__ p, size_t n, float *out)
{
    const unsigned total = gridDim.x, l = blockIdx.x;
    const unsigned logical = (XCD && total % 8u == 0u) ? (l % 8u) * (total / 8u) + l / 8u : l;
    const size_t base = (size_t)logical * 8192;
    float v[32];
#pragma unroll
    for (int j = 0; j < 32; j++) {
        const size_t i = base + threadIdx.x + 256 * j;
        v[j] = p[i < n ? i : n - 1];
    }
    float m = 0.f;
#pragma unroll
    for (int j = 0; j < 32; j++) m = fmaxf(m, v[j]);
    if (m == 12345.678f) out[l] = m;
}

// the same bytes with one dwordx4 per lane: 8 loads per lane, 4 KB per workgroup and instruction
template <bool XCD>
__global__ __launch_bounds__(256) void k_linear4(const float4 *__restrict__ p, size_t n4, float *out)
{
    const unsigned total = gridDim.x, l = blockIdx.x;
    const unsigned logical = (XCD && total % 8u == 0u) ? (l % 8u) * (total / 8u) + l / 8u : l;
    const size_t base = (size_t)logical * 2048;
    float4 v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const size_t i = base + threadIdx.x + 256 * j;
        v[j] = p[i < n4 ? i : n4 - 1];
    }
    float m = 0.f;
#pragma unroll
    for (int j = 0; j < 8; j++) m = fmaxf(fmaxf(m, v[j].x), fmaxf(v[j].y, fmaxf(v[j].z, v[j].w)));
    if (m == 12345.678f) out[l] = m;
}

static float4 *flushbuf;
static float *outbuf;
__global__ void k_flush(const float4 *p, size_t n4, float *out)
{
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = p[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) out[0] = acc;
}

template <class F>
static void timeit(const char *name, double bytes, F launch)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f, tot = 0;
    const int reps = 8;
    for (int r = 0; r < reps; r++) {
        hipLaunchKernelGGL(k_flush, 2048, 256, 0, 0, (const float4 *)flushbuf, ((size_t)1 << 30) / 16, outbuf);  // cold start
        hipEventRecord(e0, 0);
        launch();
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float t;
        hipEventElapsedTime(&t, e0, e1);
        if (r) { tot += t; if (t < best) best = t; }
    }
    printf("%-34s %8.4f ms avg  %8.4f ms best  %7.2f TB/s avg\n", name, tot / (reps - 1), best, bytes / (tot / (reps - 1) * 1e-3) / 1e12);
}

int main()
{
    const int W = 2962, H = 1125, F = 60;
    const size_t n = (size_t)W * H * F;
    float *frames;
    hipMalloc(&frames, n * 4 + 64);
    hipMalloc(&flushbuf, (size_t)1 << 30);
    hipMalloc(&outbuf, 4 << 20);
    hipMemset(frames, 0, n * 4);
    hipMemset(flushbuf, 0, (size_t)1 << 30);
    const double bytes = (double)n * 4;
#define TILE(TW, TH, X)                                                                                                   \
    {                                                                                                                     \
        const int tx = (W + TW - 1) / TW, ty = (H + TH - 1) / TH;                                                         \
        char nm[64];                                                                                                      \
        snprintf(nm, sizeof nm, "tile %4d x %2d %s (%d wgs)", TW, TH, X ? "xcd" : "   ", tx * ty * F);                     \
        timeit(nm, bytes, [&]() { hipLaunchKernelGGL((k_tile<TW, TH, X>), tx * ty * F, 256, 0, 0, frames, W, H, tx, ty, outbuf); }); \
    }
    TILE(256, 32, true)
    TILE(256, 32, false)
    TILE(512, 16, true)
    TILE(1024, 8, true)
    TILE(1024, 8, false)
    TILE(2048, 4, true)
    {
        const int tx = (W + 255) / 256, ty = (H + 31) / 32;
        float *colp, *rowp;
        hipMalloc(&colp, (size_t)F * ty * tx * 256 * 4 * 4 + (1 << 20));
        hipMalloc(&rowp, (size_t)F * tx * 3 * H * 4);
#define LIKE(KB, RED, MODE, label) timeit(label, bytes, [&]() { hipLaunchKernelGGL((k_tile_like_stats<KB, RED, MODE>), tx * ty * F, 256, 0, 0, frames, W, H, tx, ty, outbuf, colp, rowp); });
        LIKE(5, false, 0, "256x32, <= 8 wg/CU by LDS (20 KB)")
        LIKE(10, false, 0, "256x32, <= 4 wg/CU by LDS (40 KB)")
        LIKE(5, true, 0, "256x32 + sums + all stores")
        LIKE(10, true, 0, "256x32 + sums + all stores, 4 wg/CU")
        LIKE(5, true, 1, "256x32 + sums, no stores")
        LIKE(5, true, 2, "256x32 + sums, column stores only")
        LIKE(5, true, 3, "256x32 + sums, row stores only")
        LIKE(5, true, 4, "256x32 + sums, row sums via LDS")
        LIKE(5, true, 5, "256x32 + sums, col stores nontemporal")
        LIKE(5, true, 6, "256x32 + sums, col stores aligned 1 KB")
        LIKE(5, true, 7, "256x32 + sums, col aligned + nontemporal")
        LIKE(5, true, 8, "256x32 + sums, 256 B of col stores")
        LIKE(5, true, 9, "256x32 + sums, 4 KB of col stores")
        LIKE(5, true, 10, "256x32 + sums, col stores into 1 MB")
    }
    {
        const unsigned g = (unsigned)((n + 8191) / 8192);
        timeit("linear 32 KB, dword, xcd", bytes, [&]() { hipLaunchKernelGGL((k_linear<true>), (g + 7) & ~7u, 256, 0, 0, frames, n, outbuf); });
        timeit("linear 32 KB, dword", bytes, [&]() { hipLaunchKernelGGL((k_linear<false>), g, 256, 0, 0, frames, n, outbuf); });
        timeit("linear 32 KB, dwordx4, xcd", bytes, [&]() { hipLaunchKernelGGL((k_linear4<true>), (g + 7) & ~7u, 256, 0, 0, (const float4 *)frames, n / 4, outbuf); });
        timeit("linear 32 KB, dwordx4", bytes, [&]() { hipLaunchKernelGGL((k_linear4<false>), g, 256, 0, 0, (const float4 *)frames, n / 4, outbuf); });
        timeit("grid-stride dwordx4 (2048 wgs)", bytes, [&]() { hipLaunchKernelGGL(k_flush, 2048, 256, 0, 0, (const float4 *)frames, n / 4, outbuf); });
    }
    return 0;
}
