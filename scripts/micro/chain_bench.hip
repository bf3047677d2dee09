// Micro-benchmark: how long does a chain of tiny dependent kernels take on one stream while other engines of the
// device are busy (H2D / D2H DMA on their own streams, a long kernel on another stream)?
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <pthread.h>
#include <unistd.h>
#include <atomic>
static double now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void k_tiny(float *p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.0f; }
__global__ void k_big(float4 *p, size_t n, int rep)
{
    for (int r = 0; r < rep; r++)
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float4 v = p[i]; v.x += 1.f; p[i] = v; }
}
static std::atomic<int> stop{0};
static char *g_host, *g_dev, *g_hostd, *g_devd;
static float4 *g_big;
static int mode_h2d, mode_d2h, mode_big, reg_mode;
static void *dma_h2d(void *) {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    int i = 0;
    while (!stop) { CK(hipMemcpyAsync(g_dev + (size_t)(i % 32) * (2u << 20), g_host + (size_t)(i % 32) * (2u << 20), 2u << 20, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); i++; }
    return nullptr;
}
static void *dma_d2h(void *) {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    while (!stop) { CK(hipMemcpyAsync(g_hostd, g_devd, 13329000, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); }
    return nullptr;
}
static void *big_kernel(void *) {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    while (!stop) { hipLaunchKernelGGL(k_big, dim3(2048), dim3(256), 0, s, g_big, (size_t)(64u << 20) / 16, 4); CK(hipStreamSynchronize(s)); }
    return nullptr;
}
int main(int argc, char **argv)
{
    CK(hipSetDevice(0));
    CK(hipHostMalloc((void **)&g_host, 64u << 20, hipHostMallocDefault));
    CK(hipHostMalloc((void **)&g_hostd, 16u << 20, hipHostMallocDefault));
    CK(hipMalloc((void **)&g_dev, 64u << 20)); CK(hipMalloc((void **)&g_devd, 16u << 20)); CK(hipMalloc((void **)&g_big, 64u << 20));
    float *p; CK(hipMalloc((void **)&p, 4096)); CK(hipMemset(p, 0, 4096));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char *names[] = {"alone", "+H2D", "+D2H", "+H2D+D2H", "+big kernel", "+all"};
    const int cfg[][3] = {{0,0,0},{1,0,0},{0,1,0},{1,1,0},{0,0,1},{1,1,1}};
    for (int c = 0; c < 6; c++) {
        stop = 0;
        pthread_t th[3]; int nt = 0;
        if (cfg[c][0]) pthread_create(&th[nt++], nullptr, dma_h2d, nullptr);
        if (cfg[c][1]) pthread_create(&th[nt++], nullptr, dma_d2h, nullptr);
        if (cfg[c][2]) pthread_create(&th[nt++], nullptr, big_kernel, nullptr);
        usleep(200000);
        const int N = 2000;
        for (int w = 0; w < 200; w++) hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, st, p);
        CK(hipStreamSynchronize(st));
        double t0 = now();
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < N; i++) hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, st, p);
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        double dt = now() - t0;
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        // launch + sync one at a time (host-latency bound)
        double t2 = now();
        for (int i = 0; i < 300; i++) { hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, st, p); CK(hipStreamSynchronize(st)); }
        double d1 = (now() - t2) / 300;
        {   // hipExtLaunchKernelGGL with null start/stop events (what TSDR_LAUNCH expands to)
            double t3 = now();
            for (int i = 0; i < N; i++) hipExtLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, st, nullptr, nullptr, 0, p);
            CK(hipStreamSynchronize(st));
            double dx = (now() - t3) / N;
            // kernels alternating with small async device-to-device copies and small pinned host-to-device copies
            t3 = now();
            for (int i = 0; i < N; i++) {
                hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, st, p);
                if ((i & 3) == 0) CK(hipMemcpyAsync(g_dev + 4096, g_dev, 1024, hipMemcpyDeviceToDevice, st));
                if ((i & 3) == 2) CK(hipMemcpyAsync(g_dev + 8192, g_host, 2048, hipMemcpyHostToDevice, st));
            }
            CK(hipStreamSynchronize(st));
            double dy = (now() - t3) / N;
            printf("%-14s hipExtLaunch chain: %.2f us/kernel; kernels + small D2D/H2D copies every other: %.2f us/kernel\n", names[c], dx * 1e6, dy * 1e6);
        }
        {   // the same chain with an event recorded after every kernel (what the engine's slot / batch bookkeeping does)
            hipEvent_t ev[8]; for (int i = 0; i < 8; i++) CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
            hipStream_t other; CK(hipStreamCreateWithFlags(&other, hipStreamNonBlocking));
            double t3 = now();
            for (int i = 0; i < N; i++) { hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, st, p); CK(hipEventRecord(ev[i & 7], st)); }
            CK(hipStreamSynchronize(st));
            double d2 = (now() - t3) / N;
            // ... and another stream waiting on every second of them, with its own tiny kernel
            t3 = now();
            for (int i = 0; i < N; i++) {
                hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, st, p); CK(hipEventRecord(ev[i & 7], st));
                if (i & 1) { CK(hipStreamWaitEvent(other, ev[i & 7], 0)); hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, other, p + 64); }
            }
            CK(hipStreamSynchronize(st)); CK(hipStreamSynchronize(other));
            double d3 = (now() - t3) / N;
            // ... ping-pong: the main stream also waits for the other one (like COMPUTE waiting for the detector's lane)
            hipEvent_t back; CK(hipEventCreateWithFlags(&back, hipEventDisableTiming));
            t3 = now();
            for (int i = 0; i < N; i++) {
                hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, st, p); CK(hipEventRecord(ev[i & 7], st));
                if ((i & 7) == 7) { CK(hipStreamWaitEvent(other, ev[i & 7], 0)); hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, other, p + 64); CK(hipEventRecord(back, other)); CK(hipStreamWaitEvent(st, back, 0)); }
            }
            CK(hipStreamSynchronize(st)); CK(hipStreamSynchronize(other));
            double d4 = (now() - t3) / N;
            printf("%-14s with an event after each kernel: %.2f us/kernel; + second stream waiting on them: %.2f; + ping-pong every 8th: %.2f\n", names[c], d2 * 1e6, d3 * 1e6, d4 * 1e6);
        }
        printf("%-14s chain of %d tiny kernels: %.2f us/kernel on the GPU (events), %.2f us/kernel wall; launch+sync %.1f us\n", names[c], N, ms * 1e3 / N, dt * 1e6 / N, d1 * 1e6);
        stop = 1;
        for (int i = 0; i < nt; i++) pthread_join(th[i], nullptr);
    }
    return 0;
}
