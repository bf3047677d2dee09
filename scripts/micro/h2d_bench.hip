// Micro-benchmark: how fast can ONE host thread hand 2 MB blocks of page-locked (hipHostRegister) memory to the
// device when it must wait for each block before reusing the buffer?  (engine.c on_block_any)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <sys/mman.h>
static double now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void k_copy(const float4 *__restrict__ s, float4 *__restrict__ a, float4 *__restrict__ b, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float4 v = s[i]; a[i] = v; b[i] = v; }
}
int main(int argc, char **argv)
{
    const size_t bytes = argc > 1 ? atol(argv[1]) : (2u << 20);
    const int iters = 2000, nblk = 64;
    char *host = (char *)mmap(nullptr, bytes * nblk, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    memset(host, 1, bytes * nblk);
    CK(hipHostRegister(host, bytes * nblk, hipHostRegisterDefault));
    char *pinned; CK(hipHostMalloc((void **)&pinned, bytes * nblk, hipHostMallocDefault));
    char *dev, *dev2; CK(hipMalloc((void **)&dev, bytes * nblk)); CK(hipMalloc((void **)&dev2, bytes * nblk));
    hipStream_t st[4]; for (int i = 0; i < 4; i++) CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
    hipEvent_t ev[4]; for (int i = 0; i < 4; i++) CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
    for (int src = 0; src < 2; src++) {
        char *h = src ? pinned : host;
        for (int split = 1; split <= 4; split *= 2) for (int spin = 0; spin < 2; spin++) {
            double t0 = now();
            for (int it = 0; it < iters; it++) {
                const size_t off = (size_t)(it % nblk) * bytes, part = bytes / split;
                for (int s = 0; s < split; s++) CK(hipMemcpyAsync(dev + off + s * part, h + off + s * part, part, hipMemcpyHostToDevice, st[s]));
                for (int s = 0; s < split; s++) {
                    if (spin) while (hipStreamQuery(st[s]) == hipErrorNotReady) {}
                    else CK(hipStreamSynchronize(st[s]));
                }
            }
            double dt = now() - t0;
            printf("%s memcpyAsync split=%d %s: %.1f us/block, %.1f GB/s\n", src ? "hipHostMalloc " : "hipHostRegister", split, spin ? "spin" : "sync", dt / iters * 1e6, bytes * (double)iters / dt / 1e9);
        }
        // kernel copy straight from host memory into two device buffers
        void *dp; CK(hipHostGetDevicePointer(&dp, h, 0));
        for (int blocks = 64; blocks <= 1024; blocks *= 4) for (int spin = 0; spin < 2; spin++) {
            double t0 = now();
            for (int it = 0; it < iters; it++) {
                const size_t off = (size_t)(it % nblk) * bytes;
                hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, st[0], (const float4 *)((char *)dp + off), (float4 *)(dev + off), (float4 *)(dev2 + off), bytes / 16);
                if (spin) while (hipStreamQuery(st[0]) == hipErrorNotReady) {} else CK(hipStreamSynchronize(st[0]));
            }
            double dt = now() - t0;
            printf("%s kernel copy2 blocks=%d %s: %.1f us/block, %.1f GB/s\n", src ? "hipHostMalloc " : "hipHostRegister", blocks, spin ? "spin" : "sync", dt / iters * 1e6, bytes * (double)iters / dt / 1e9);
        }
    }
    // D2H of a 13.3 MB frame, one stream vs split
    const size_t fb = 13329000;
    char *hf; CK(hipHostMalloc((void **)&hf, fb * 4, hipHostMallocDefault));
    for (int split = 1; split <= 4; split *= 2) {
        double t0 = now();
        const int n = 300;
        for (int it = 0; it < n; it++) {
            const size_t part = (fb / split) & ~(size_t)15;
            for (int s = 0; s < split; s++) CK(hipMemcpyAsync(hf + s * part, dev + s * part, s == split - 1 ? fb - s * part : part, hipMemcpyDeviceToHost, st[s]));
            for (int s = 0; s < split; s++) CK(hipStreamSynchronize(st[s]));
        }
        double dt = now() - t0;
        printf("D2H frame split=%d: %.1f us/frame, %.1f GB/s\n", split, dt / n * 1e6, fb * (double)n / dt / 1e9);
    }
    // both directions at once
    {
        double t0 = now();
        const int n = 300;
        for (int it = 0; it < n; it++) {
            CK(hipMemcpyAsync(hf, dev, fb, hipMemcpyDeviceToHost, st[0]));
            for (int k = 0; k < 6; k++) CK(hipMemcpyAsync(dev2 + (size_t)k * bytes, host + (size_t)k * bytes, bytes, hipMemcpyHostToDevice, st[1]));
            CK(hipStreamSynchronize(st[0])); CK(hipStreamSynchronize(st[1]));
        }
        double dt = now() - t0;
        printf("duplex: D2H frame + 6 H2D blocks: %.1f us, %.1f GB/s each way\n", dt / n * 1e6, fb * (double)n / dt / 1e9);
    }
    return 0;
}
