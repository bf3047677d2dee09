// tests/emu/hipemu.h — a minimal CPU stand-in for the HIP kernel language, so that the kernels in
// tempestsdr_amd/csrc/*.h can be run by the CPU-only test-suite: a workgroup is a set of OS threads,
// __shared__ is a static array (workgroups run one after the other), __syncthreads() a barrier.
// TEST INFRASTRUCTURE ONLY: never loaded by the product.
#pragma once
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <string.h>

#include <functional>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)

struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
static inline float2 make_float2(float x, float y) { float2 r; r.x = x; r.y = y; return r; }
static inline float4 make_float4(float x, float y, float z, float w) { float4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }

struct emu_dim3 {
    unsigned x, y, z;
    emu_dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
static thread_local emu_dim3 threadIdx, blockIdx;
static emu_dim3 blockDim, gridDim;
static pthread_barrier_t emu_barrier;
static inline void __syncthreads() { pthread_barrier_wait(&emu_barrier); }

static inline void sincospif(float x, float *s, float *c)
{
    // exact argument reduction like the device function: x is a dyadic fraction in the callers
    double r = fmod((double)x, 2.0);
    *s = (float)sin(M_PI * r);
    *c = (float)cos(M_PI * r);
}

// Runs `body()` (a kernel call with its arguments bound) for every thread of every workgroup of the grid.
static inline void emu_launch(emu_dim3 grid, unsigned block, const std::function<void()> &body)
{
    gridDim = grid;
    blockDim = emu_dim3(block);
    pthread_barrier_init(&emu_barrier, nullptr, block);
    std::vector<std::thread> th;
    for (unsigned t = 0; t < block; t++)
        th.emplace_back([&, t]() {
            for (unsigned bz = 0; bz < grid.z; bz++)
                for (unsigned by = 0; by < grid.y; by++)
                    for (unsigned bx = 0; bx < grid.x; bx++) {
                        threadIdx = emu_dim3(t, 0, 0);
                        blockIdx = emu_dim3(bx, by, bz);
                        body();
                        pthread_barrier_wait(&emu_barrier);  // static __shared__ is reused by the next workgroup
                    }
        });
    for (auto &x : th) x.join();
    pthread_barrier_destroy(&emu_barrier);
}
