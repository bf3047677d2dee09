// wave_reduce_check.hip — lane 0 of the DPP / permlane-swap wave reductions of tsdrgpu_frame.hip must equal, bit for
// bit, lane 0 of the __shfl_down trees they replace (same additions in the same order).
// Build: hipcc --offload-arch=gfx950 -O3 -o wave_reduce_check wave_reduce_check.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define WAVE_REDUCE_CHECK 1
#include "../../tempestsdr_amd/csrc/wave_reduce.h"

__device__ __forceinline__ float ref_sum(float v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64); return v; }
__device__ __forceinline__ float ref_min(float v) { for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_down(v, o, 64)); return v; }
__device__ __forceinline__ float ref_max(float v) { for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_down(v, o, 64)); return v; }

__global__ void k(const float *in, float *out)
{
    const float v = in[blockIdx.x * 64 + threadIdx.x];
    const float a = wave_sum(v), b = ref_sum(v), c = wave_min(v), d = ref_min(v), e = wave_max(v), f = ref_max(v);
    if (threadIdx.x == 0) {
        float *o = out + 6 * blockIdx.x;
        o[0] = a; o[1] = b; o[2] = c; o[3] = d; o[4] = e; o[5] = f;
    }
}

int main()
{
    const int nw = 4096;
    float *h = (float *)malloc(nw * 64 * 4), *ho = (float *)malloc(nw * 6 * 4);
    srand(7);
    for (int i = 0; i < nw * 64; i++) {
        const int k = rand() % 16;
        h[i] = (k == 0) ? -0.0f : (k == 1 ? 1e30f * (rand() % 3 - 1) : (float)rand() / RAND_MAX * ((rand() & 1) ? 1.0f : -1000.0f));
    }
    float *d, *o;
    hipMalloc(&d, nw * 64 * 4); hipMalloc(&o, nw * 6 * 4);
    hipMemcpy(d, h, nw * 64 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, nw, 64, 0, 0, d, o);
    hipMemcpy(ho, o, nw * 6 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int w = 0; w < nw; w++)
        for (int q = 0; q < 3; q++)
            if (memcmp(&ho[6 * w + 2 * q], &ho[6 * w + 2 * q + 1], 4)) bad++;
    printf("wave reductions: %d waves x {sum, min, max}: %d mismatches\n", nw, bad);
    return bad != 0;
}
