// mall_bench.hip — what is a producer -> consumer hand-off through HBM worth when the data still sits in the 256 MiB
// Infinity Cache?  Kernel W writes S bytes, kernel R reads them back (R timed), for a range of S; and the same with a
// 1 GiB "flush" read in between.  Build: hipcc --offload-arch=gfx950 -O2 -o mall_bench mall_bench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__global__ void k_write(float4 *p, size_t n4, float v)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) p[i] = make_float4(v, v, v, v);
}
__global__ void k_read(const float4 *p, size_t n4, float *out)
{
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = p[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) out[0] = acc;
}
__global__ void k_copy(const float4 *p, float4 *q, size_t n4)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) q[i] = p[i];
}

int main()
{
    const size_t cap = (size_t)2 << 30;
    float4 *a, *b, *flush;
    float *out;
    hipMalloc(&a, cap); hipMalloc(&b, cap); hipMalloc(&flush, (size_t)1 << 30); hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * 8, block = 256;
    const size_t sizes_mb[] = {16, 32, 64, 96, 128, 160, 192, 224, 256, 320, 512, 1024};
    printf("%8s %14s %14s %14s %14s\n", "MB", "read after W", "read, flushed", "copy after W", "copy, flushed");
    for (size_t s : sizes_mb) {
        const size_t n4 = (s << 20) / 16;
        float ms[4] = {0, 0, 0, 0};
        const int reps = 6;
        for (int mode = 0; mode < 4; mode++) {
            float tot = 0;
            for (int r = 0; r < reps; r++) {
                hipLaunchKernelGGL(k_write, grid, block, 0, 0, a, n4, (float)r);
                if (mode & 1) hipLaunchKernelGGL(k_read, grid, block, 0, 0, (const float4 *)flush, ((size_t)1 << 30) / 16, out);
                hipEventRecord(e0, 0);
                if (mode < 2) hipLaunchKernelGGL(k_read, grid, block, 0, 0, (const float4 *)a, n4, out);
                else hipLaunchKernelGGL(k_copy, grid, block, 0, 0, (const float4 *)a, b, n4);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float t;
                hipEventElapsedTime(&t, e0, e1);
                if (r) tot += t;
            }
            ms[mode] = tot / (reps - 1);
        }
        const double gb = (double)(s << 20) / 1e9;
        printf("%8zu %9.0f GB/s %9.0f GB/s %9.0f GB/s %9.0f GB/s\n", s, gb / (ms[0] * 1e-3), gb / (ms[1] * 1e-3), 2 * gb / (ms[2] * 1e-3), 2 * gb / (ms[3] * 1e-3));
    }
    return 0;
}
