// Micro-benchmark: H2D blocks (2 MB, waited for one by one) and D2H frames (13.3 MB) from two host threads on two
// streams at once: what does each direction get?  Run under different GPU_MAX_HW_QUEUES.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <pthread.h>
#include <unistd.h>
#include <atomic>
static double now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static std::atomic<int> stop{0};
static char *g_host, *g_dev, *g_hostd, *g_devd;
static std::atomic<long> n_h2d{0}, n_d2h{0};
static hipStream_t s_up, s_down;
static void *th_h2d(void *) {
    int i = 0;
    while (!stop) { CK(hipMemcpyAsync(g_dev + (size_t)(i % 32) * (2u << 20), g_host + (size_t)(i % 32) * (2u << 20), 2u << 20, hipMemcpyHostToDevice, s_up)); CK(hipStreamSynchronize(s_up)); i++; n_h2d++; }
    return nullptr;
}
static void *th_d2h(void *) {
    hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    while (!stop) { CK(hipMemcpyAsync(g_hostd, g_devd, 13329000, hipMemcpyDeviceToHost, s_down)); CK(hipEventRecord(ev, s_down)); CK(hipEventSynchronize(ev)); n_d2h++; }
    return nullptr;
}
__global__ void k_tiny(float *p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.0f; }
int main(int argc, char **argv)
{
    const int nextra = argc > 1 ? atoi(argv[1]) : 0;  // extra streams created first (shifts which hardware queues the two get)
    CK(hipSetDevice(0));
    CK(hipHostMalloc((void **)&g_host, 64u << 20, hipHostMallocDefault));
    CK(hipHostMalloc((void **)&g_hostd, 16u << 20, hipHostMallocDefault));
    CK(hipMalloc((void **)&g_dev, 64u << 20)); CK(hipMalloc((void **)&g_devd, 16u << 20));
    float *p; CK(hipMalloc((void **)&p, 4096));
    hipStream_t extra[8], comp;
    CK(hipStreamCreateWithFlags(&comp, hipStreamNonBlocking));
    for (int i = 0; i < nextra; i++) { CK(hipStreamCreateWithFlags(&extra[i], hipStreamNonBlocking)); hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, extra[i], p); CK(hipStreamSynchronize(extra[i])); }
    CK(hipStreamCreateWithFlags(&s_up, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s_down, hipStreamNonBlocking));
    for (int mode = 0; mode < 3; mode++) {
        stop = 0; n_h2d = 0; n_d2h = 0;
        pthread_t a, b;
        if (mode != 1) pthread_create(&a, nullptr, th_h2d, nullptr);
        if (mode != 0) pthread_create(&b, nullptr, th_d2h, nullptr);
        double t0 = now();
        // a compute stream doing small kernels meanwhile
        while (now() - t0 < 1.5) { for (int i = 0; i < 50; i++) hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, comp, p); CK(hipStreamSynchronize(comp)); }
        double dt = now() - t0;
        stop = 1;
        if (mode != 1) pthread_join(a, nullptr);
        if (mode != 0) pthread_join(b, nullptr);
        printf("extra=%d %s: H2D %.1f GB/s  D2H %.1f GB/s\n", nextra, mode == 0 ? "H2D only" : (mode == 1 ? "D2H only" : "both    "), n_h2d * 2.097152e-3 / dt, n_d2h * 13.329e-3 / dt);
    }
    return 0;
}
