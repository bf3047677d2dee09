// How long does hipMalloc take as a function of size (and does it stall a busy stream)?  Decides how the detector's
// retention ring (tsdrgpu_autocorr_set_certify mode 1, up to 32 GiB) is allocated.
//   hipcc --offload-arch=gfx950 -O2 -o malloc_bench malloc_bench.hip && ./malloc_bench
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void spin(float *p, int n) { float a = p[threadIdx.x]; for (int i = 0; i < n; i++) a = a * 1.0001f + 0.5f; p[threadIdx.x] = a; }
int main()
{
    hipFree(0);
    const size_t GiB = (size_t)1 << 30;
    for (int rep = 0; rep < 2; rep++)
        for (size_t g : {(size_t)1, (size_t)4, (size_t)16, (size_t)32}) {
            void *p = nullptr;
            double t0 = now();
            hipError_t e = hipMalloc(&p, g * GiB);
            double t1 = now();
            hipMemset(p, 0, 64);
            hipDeviceSynchronize();
            double t2 = now();
            hipFree(p);
            double t3 = now();
            printf("hipMalloc %2zu GiB: %8.2f ms (rc %d), first touch %6.2f ms, hipFree %8.2f ms\n", g, (t1 - t0) * 1e3, (int)e, (t2 - t1) * 1e3, (t3 - t2) * 1e3);
        }
    // 32 x 1 GiB
    {
        void *p[32];
        double t0 = now();
        for (int i = 0; i < 32; i++) hipMalloc(&p[i], GiB);
        double t1 = now();
        printf("32 x hipMalloc 1 GiB: %8.2f ms total\n", (t1 - t0) * 1e3);
        for (int i = 0; i < 32; i++) hipFree(p[i]);
    }
    // does a hipMalloc on another thread delay kernels queued on a stream?
    {
        hipStream_t st;
        hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        float *d;
        hipMalloc(&d, 4096);
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        for (int with = 0; with < 2; with++) {
            std::thread th;
            void *big = nullptr;
            hipEventRecord(a, st);
            if (with) th = std::thread([&] { hipSetDevice(0); hipMalloc(&big, 8 * GiB); });
            for (int i = 0; i < 2000; i++) spin<<<1, 64, 0, st>>>(d, 100);
            hipEventRecord(b, st);
            hipEventSynchronize(b);
            if (with) th.join();
            float ms = 0;
            hipEventElapsedTime(&ms, a, b);
            printf("2000 small kernels on a stream %s a concurrent hipMalloc(8 GiB): %.2f ms\n", with ? "WITH" : "without", ms);
            if (big) hipFree(big);
        }
    }
    // virtual memory management: reserve 32 GiB of address space, map 1 GiB at a time
    {
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = 0;
        size_t gran = 0;
        hipError_t e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum);
        printf("VMM granularity %zu (rc %d)\n", gran, (int)e);
        void *va = nullptr;
        double t0 = now();
        e = hipMemAddressReserve(&va, 32 * GiB, 0, nullptr, 0);
        double t1 = now();
        printf("hipMemAddressReserve 32 GiB: %.2f ms (rc %d)\n", (t1 - t0) * 1e3, (int)e);
        if (e == hipSuccess) {
            hipMemGenericAllocationHandle_t h[4];
            for (int i = 0; i < 4; i++) {
                double a0 = now();
                hipError_t e1 = hipMemCreate(&h[i], GiB, &prop, 0);
                hipError_t e2 = hipMemMap((char *)va + i * GiB, GiB, 0, h[i], 0);
                hipMemAccessDesc acc = {};
                acc.location = prop.location;
                acc.flags = hipMemAccessFlagsProtReadWrite;
                hipError_t e3 = hipMemSetAccess((char *)va + i * GiB, GiB, &acc, 1);
                double a1 = now();
                printf("  map segment %d: %.2f ms (rc %d %d %d)\n", i, (a1 - a0) * 1e3, (int)e1, (int)e2, (int)e3);
            }
            hipMemset(va, 1, 4 * GiB);
            printf("  memset over the 4 mapped GiB: rc %d\n", (int)hipDeviceSynchronize());
            for (int i = 0; i < 4; i++) { hipMemUnmap((char *)va + i * GiB, GiB); hipMemRelease(h[i]); }
            hipMemAddressFree(va, 32 * GiB);
        }
    }
    return 0;
}
