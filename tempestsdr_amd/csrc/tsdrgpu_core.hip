// tsdrgpu_core.hip — context, memory, AM demodulation and the fractional area
// resampler for gfx950.  See include/tsdrgpu.h for the contract of each entry
// point and the reference code it replaces.
#include "tsdrgpu_internal.h"
#include "resample_math.h"

// ---------------------------------------------------------------------------
// context / memory
// ---------------------------------------------------------------------------
// Hardware queues.  The HIP runtime spreads a process's streams over GPU_MAX_HW_QUEUES hardware queues (4 by default,
// read when the runtime initialises, i.e. at the process's first HIP call).  The streaming engine keeps three lanes
// busy at once (COMPUTE, UPLOAD, DOWNLOAD); measured on MI355X through the tsdr_* API it sustains 5.0-5.1 GS/s when
// they share two hardware queues and 1.8-3.6 GS/s when each lane has a queue of its own (with more queues active the
// command processor's switching between them adds tens of microseconds to every small kernel of the frame path).
// This library does NOT touch the process environment: a launcher that wants the streaming optimum exports
// GPU_MAX_HW_QUEUES=2 before the process starts (tempestsdr_amd/tsdrlib.py and bench.py's e2e leg do), or sets
// TSDR_GPU_HW_QUEUES=2, which libTSDRLibrary.so's tsdr_init forwards if the host has not chosen a value (host/tsdr_api.c).

// the text of the calling thread's last failing call (the engine drives one context from five threads)
char *tsdr_errbuf(void)
{
    static thread_local char buf[512];
    return buf;
}

extern "C" int tsdrgpu_create(tsdrgpu_t **out, int device)
{
    if (!out) return TSDRGPU_EINVAL;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return TSDRGPU_EHIP;  // no CPU fallback
    if (device < 0 || device >= count) return TSDRGPU_EINVAL;
    tsdrgpu_t *g = (tsdrgpu_t *)calloc(1, sizeof(tsdrgpu_t));
    if (!g) return TSDRGPU_ENOMEM;
    g->device = device;
    // The side stream carries short, latency-bound kernels (the sync detector's chain) beside bandwidth-bound ones
    // on the main stream: it gets the highest priority, so its workgroups take the slots the big kernels free
    // instead of queueing behind their whole grids (measured: the chain's re-collapse of two strips took 0.5 ms
    // beside the FFT trips at equal priority).
    int prio_lo = 0, prio_hi = 0;
    if (hipSetDevice(device) == hipSuccess) (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&g->prop, device) != hipSuccess ||
        hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithPriority(&g->stream2, hipStreamNonBlocking, prio_hi) != hipSuccess ||
        hipStreamCreateWithPriority(&g->bg, hipStreamNonBlocking, prio_lo) != hipSuccess ||
        hipStreamCreateWithFlags(&g->up, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&g->down, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&g->fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreate(&g->t0) != hipSuccess || hipEventCreate(&g->t1) != hipSuccess) {
        free(g);
        return TSDRGPU_EHIP;
    }
    *out = g;
    return TSDRGPU_OK;
}

extern "C" void tsdrgpu_destroy(tsdrgpu_t *g)
{
    if (!g) return;
    hipSetDevice(g->device);
    hipStreamSynchronize(g->stream);
    hipStreamSynchronize(g->stream2);
    hipStreamSynchronize(g->bg);
    hipStreamDestroy(g->bg);
    hipStreamSynchronize(g->up);
    hipStreamSynchronize(g->down);
    hipStreamDestroy(g->up);
    hipStreamDestroy(g->down);
    hipStreamDestroy(g->stream2);
    hipEventDestroy(g->fork);
    for (int i = 0; i < g->cap_spans; i++) {
        if (g->spans[i].a) hipEventDestroy(g->spans[i].a);
        if (g->spans[i].b) hipEventDestroy(g->spans[i].b);
    }
    free(g->spans);
    if (g->fft_ws) hipFree(g->fft_ws);
    if (g->superb_ws) hipFree(g->superb_ws);
    if (g->superb_h_off) hipHostFree(g->superb_h_off);
    if (g->fftx_tw) hipFree(g->fftx_tw);
    hipEventDestroy(g->t0);
    hipEventDestroy(g->t1);
    hipStreamDestroy(g->stream);
    free(g);
}

extern "C" const char *tsdrgpu_last_error(tsdrgpu_t *g) { return g ? tsdr_errbuf() : "no context"; }
extern "C" void *tsdrgpu_stream(tsdrgpu_t *g) { return g ? (void *)g->stream : nullptr; }

extern "C" int tsdrgpu_sync(tsdrgpu_t *g)
{
    if (!g) return TSDRGPU_EINVAL;
    HIP_TRY(g, hipStreamSynchronize(g->stream));
    HIP_TRY(g, hipStreamSynchronize(g->stream2));
    HIP_TRY(g, hipStreamSynchronize(g->bg));
    return TSDRGPU_OK;
}

// ---- TSDRGPU_REDZONES (tsdrgpu_internal.h) --------------------------------------------------------------------------------
#include <mutex>
#include <unordered_map>
namespace {
constexpr size_t RZ = 4096;
struct RzRec { void *base; size_t bytes; };
std::mutex rz_lock;
std::unordered_map<void *, RzRec> *rz_live;  // (leaked on purpose: frees may come from static destructors)
int rz_mode = -1;                            // -1 not read yet, 0 off, 1 abort on a violation, 2 report
int rz_on()
{
    if (rz_mode < 0) {
        const char *e = getenv("TSDRGPU_REDZONES");
        rz_mode = (e && (e[0] == '1' || e[0] == '2')) ? e[0] - '0' : 0;
    }
    return rz_mode;
}
}  // namespace

hipError_t tsdr_redzone_malloc(void **p, size_t n)
{
    if (!rz_on()) return (hipMalloc)(p, n);
    char *base = nullptr;
    hipError_t rc = (hipMalloc)((void **)&base, n + 2 * RZ);
    if (rc != hipSuccess) return rc;
    // (synchronous memsets: a debug mode)
    if (hipMemset(base, 0xFF, RZ) != hipSuccess || hipMemset(base + RZ + n, 0xFF, RZ) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        (void)(hipFree)(base);
        return hipErrorOutOfMemory;
    }
    std::lock_guard<std::mutex> hold(rz_lock);
    if (!rz_live) rz_live = new std::unordered_map<void *, RzRec>();
    (*rz_live)[base + RZ] = RzRec{base, n};
    *p = base + RZ;
    return hipSuccess;
}

hipError_t tsdr_redzone_free(void *p)
{
    if (!p || !rz_on()) return (hipFree)(p);
    RzRec rec{nullptr, 0};
    {
        std::lock_guard<std::mutex> hold(rz_lock);
        if (rz_live) {
            auto it = rz_live->find(p);
            if (it != rz_live->end()) { rec = it->second; rz_live->erase(it); }
        }
    }
    if (!rec.base) return (hipFree)(p);  // not ours (allocated before the mode was read: cannot happen) — hand it on
    static thread_local unsigned char host[2 * RZ];
    (void)hipDeviceSynchronize();
    if (hipMemcpy(host, rec.base, RZ, hipMemcpyDeviceToHost) == hipSuccess &&
        hipMemcpy(host + RZ, (char *)rec.base + RZ + rec.bytes, RZ, hipMemcpyDeviceToHost) == hipSuccess) {
        for (int side = 0; side < 2; side++) {
            size_t bad = 0, first = 0;
            for (size_t i = 0; i < RZ; i++)
                if (host[side * RZ + i] != 0xFF) { if (!bad) first = i; bad++; }
            if (bad) {
                fprintf(stderr, "tsdrgpu: RED ZONE VIOLATION: %zu bytes written %s a device allocation of %zu bytes (first at byte %+ld from that edge)\n",
                        bad, side ? "above" : "below", rec.bytes, side ? (long)first : (long)first - (long)RZ);
                fflush(stderr);
                if (const char *log = getenv("TSDRGPU_REDZONE_LOG")) {  // (a test run collects the reports of all its processes here)
                    if (FILE *f = fopen(log, "a")) {
                        fprintf(f, "%zu bytes written %s a device allocation of %zu bytes (first at byte %+ld from that edge)\n", bad,
                                side ? "above" : "below", rec.bytes, side ? (long)first : (long)first - (long)RZ);
                        fclose(f);
                    }
                }
                if (rz_on() == 1) abort();
            }
        }
    }
    return (hipFree)(rec.base);
}

extern "C" int tsdrgpu_device_name(tsdrgpu_t *g, char *buf, size_t buflen)
{
    if (!g || !buf || !buflen) return TSDRGPU_EINVAL;
    snprintf(buf, buflen, "%s (%s, %d CUs)", g->prop.name, g->prop.gcnArchName, g->prop.multiProcessorCount);
    return TSDRGPU_OK;
}

extern "C" int tsdrgpu_alloc(tsdrgpu_t *g, void **d_ptr, size_t bytes)
{
    if (!g || !d_ptr) return TSDRGPU_EINVAL;
    HIP_TRY(g, hipSetDevice(g->device));
    if (hipMalloc(d_ptr, bytes ? bytes : 1) != hipSuccess) return tsdr_fail(g, TSDRGPU_ENOMEM, "hipMalloc", "out of device memory");
    return TSDRGPU_OK;
}
extern "C" int tsdrgpu_free(tsdrgpu_t *g, void *d_ptr)
{
    if (!g) return TSDRGPU_EINVAL;
    if (d_ptr) HIP_TRY(g, hipFree(d_ptr));
    return TSDRGPU_OK;
}
extern "C" int tsdrgpu_alloc_host(tsdrgpu_t *g, void **h_ptr, size_t bytes)
{
    if (!g || !h_ptr) return TSDRGPU_EINVAL;
    HIP_TRY(g, hipSetDevice(g->device));
    if (hipHostMalloc(h_ptr, bytes ? bytes : 1, hipHostMallocPortable) != hipSuccess)
        return tsdr_fail(g, TSDRGPU_ENOMEM, "hipHostMalloc", "out of pinned memory");
    return TSDRGPU_OK;
}
extern "C" int tsdrgpu_free_host(tsdrgpu_t *g, void *h_ptr)
{
    if (!g) return TSDRGPU_EINVAL;
    if (h_ptr) HIP_TRY(g, hipHostFree(h_ptr));
    return TSDRGPU_OK;
}
extern "C" int tsdrgpu_upload(tsdrgpu_t *g, void *d_dst, const void *h_src, size_t bytes)
{
    if (!g) return TSDRGPU_EINVAL;
    if (bytes) HIP_TRY(g, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, g->stream));
    return TSDRGPU_OK;
}
extern "C" int tsdrgpu_download(tsdrgpu_t *g, void *h_dst, const void *d_src, size_t bytes)
{
    if (!g) return TSDRGPU_EINVAL;
    if (bytes) HIP_TRY(g, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, g->stream));
    return TSDRGPU_OK;
}
extern "C" int tsdrgpu_copy(tsdrgpu_t *g, void *d_dst, const void *d_src, size_t bytes)
{
    if (!g) return TSDRGPU_EINVAL;
    if (bytes) HIP_TRY(g, hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, g->stream));
    return TSDRGPU_OK;
}
// one launch that copies a block to two destinations (the engine appends every plugin block to the resampler's and to
// the detector's sample stream): dwordx4 when all three pointers allow, dwords otherwise
__global__ __launch_bounds__(256) void k_copy2(const float *__restrict__ src, float *__restrict__ d1, float *__restrict__ d2, long long n, int vec)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    if (vec) {
        const long long nq = n / 4;
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += stride) {
            const float4 v = reinterpret_cast<const float4 *>(src)[i];
            reinterpret_cast<float4 *>(d1)[i] = v;
            reinterpret_cast<float4 *>(d2)[i] = v;
        }
        for (long long i = nq * 4 + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) { d1[i] = src[i]; d2[i] = src[i]; }
    } else {
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) { const float v = src[i]; d1[i] = v; d2[i] = v; }
    }
}
extern "C" int tsdrgpu_copy2(tsdrgpu_t *g, void *d_dst1, void *d_dst2, const void *d_src, size_t bytes)
{
    if (!g || (bytes & 3)) return g ? tsdr_fail(g, TSDRGPU_EINVAL, "tsdrgpu_copy2", "size must be a multiple of 4") : TSDRGPU_EINVAL;
    if (!bytes) return TSDRGPU_OK;
    const long long n = (long long)(bytes / 4);
    const int vec = ((((uintptr_t)d_dst1) | ((uintptr_t)d_dst2) | ((uintptr_t)d_src)) & 15) == 0;
    long long blocks = (n / (vec ? 4 : 1) + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    TSDR_LAUNCH(g, PROF_EXTRAS, g->stream, k_copy2, (unsigned)blocks, 256, (const float *)d_src, (float *)d_dst1, (float *)d_dst2, n, vec);
    KERNEL_CHECK(g, "k_copy2");
    return TSDRGPU_OK;
}
// n source blocks, back to back, into two destinations with ONE launch (the streaming engine appends every block the
// source queued since the last look to both sample streams; one launch per block made the COMPUTE lane launch bound)
#define GATHER_MAX 32
struct GatherArgs {
    const float *src[GATHER_MAX];
    long long off[GATHER_MAX + 1];  // floats: block b lands at dst + off[b]
};
__global__ __launch_bounds__(256) void k_gather2(GatherArgs a, float *__restrict__ d1, float *__restrict__ d2)
{
    const int b = blockIdx.y;
    const float *__restrict__ src = a.src[b];
    const long long n = a.off[b + 1] - a.off[b];
    float *o1 = d1 + a.off[b];
    float *o2 = d2 ? d2 + a.off[b] : nullptr;
    const bool vec = ((((uintptr_t)src) | ((uintptr_t)o1) | ((uintptr_t)o2)) & 15) == 0;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long first = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long done = 0;
    if (vec) {
        const long long nq = n / 4;
        for (long long i = first; i < nq; i += stride) {
            const float4 v = reinterpret_cast<const float4 *>(src)[i];
            reinterpret_cast<float4 *>(o1)[i] = v;
            if (o2) reinterpret_cast<float4 *>(o2)[i] = v;
        }
        done = nq * 4;
    }
    for (long long i = done + first; i < n; i += stride) {
        const float v = src[i];
        o1[i] = v;
        if (o2) o2[i] = v;
    }
}
extern "C" int tsdrgpu_gather2(tsdrgpu_t *g, void *d_dst1, void *d_dst2, const void *const *d_srcs, const size_t *bytes, int n)
{
    if (!g || !d_dst1 || !d_srcs || !bytes || n < 0 || n > GATHER_MAX) return g ? tsdr_fail(g, TSDRGPU_EINVAL, "tsdrgpu_gather2", "bad argument") : TSDRGPU_EINVAL;
    if (n == 0) return TSDRGPU_OK;
    GatherArgs a;
    memset(&a, 0, sizeof(a));
    long long off = 0, longest = 0;
    for (int b = 0; b < n; b++) {
        if (!d_srcs[b] || (bytes[b] & 3)) return tsdr_fail(g, TSDRGPU_EINVAL, "tsdrgpu_gather2", "block sizes must be multiples of 4");
        a.src[b] = (const float *)d_srcs[b];
        a.off[b] = off;
        const long long nf = (long long)(bytes[b] / 4);
        off += nf;
        if (nf > longest) longest = nf;
    }
    a.off[n] = off;
    if (off == 0) return TSDRGPU_OK;
    long long blocks = (longest / 4 + 255) / 256;
    if (blocks > 512) blocks = 512;
    if (blocks < 1) blocks = 1;
    TSDR_LAUNCH(g, PROF_EXTRAS, g->stream, k_gather2, dim3((unsigned)blocks, (unsigned)n), 256, a, (float *)d_dst1, (float *)d_dst2);
    KERNEL_CHECK(g, "k_gather2");
    return TSDRGPU_OK;
}
extern "C" int tsdrgpu_zero(tsdrgpu_t *g, void *d_ptr, size_t bytes)
{
    if (!g) return TSDRGPU_EINVAL;
    if (bytes) HIP_TRY(g, hipMemsetAsync(d_ptr, 0, bytes, g->stream));
    return TSDRGPU_OK;
}
// ---- lanes and events -------------------------------------------------------------------------------
hipStream_t tsdr_lane_stream(tsdrgpu_t *g, int lane)
{
    switch (lane) {
        case TSDRGPU_LANE_COMPUTE: return g->stream;
        case TSDRGPU_LANE_SIDE: return g->stream2;
        case TSDRGPU_LANE_BACKGROUND: return g->bg;
        case TSDRGPU_LANE_UPLOAD: return g->up;
        case TSDRGPU_LANE_DOWNLOAD: return g->down;
    }
    return nullptr;
}
static hipStream_t lane_stream(tsdrgpu_t *g, int lane) { return tsdr_lane_stream(g, lane); }
extern "C" int tsdrgpu_bind_thread(tsdrgpu_t *g)
{
    if (!g) return TSDRGPU_EINVAL;
    HIP_TRY(g, hipSetDevice(g->device));
    return TSDRGPU_OK;
}
extern "C" int tsdrgpu_event_create(tsdrgpu_t *g, tsdrgpu_event_t **out)
{
    if (!g || !out) return TSDRGPU_EINVAL;
    hipEvent_t ev = nullptr;
    HIP_TRY(g, hipSetDevice(g->device));
    HIP_TRY(g, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    *out = (tsdrgpu_event_t *)ev;
    return TSDRGPU_OK;
}
extern "C" void tsdrgpu_event_destroy(tsdrgpu_t *g, tsdrgpu_event_t *ev)
{
    (void)g;
    if (ev) (void)hipEventDestroy((hipEvent_t)ev);
}
extern "C" int tsdrgpu_event_record(tsdrgpu_t *g, tsdrgpu_event_t *ev, int lane)
{
    if (!g || !ev || !lane_stream(g, lane)) return g ? tsdr_fail(g, TSDRGPU_EINVAL, "tsdrgpu_event_record", "bad lane") : TSDRGPU_EINVAL;
    HIP_TRY(g, hipEventRecord((hipEvent_t)ev, lane_stream(g, lane)));
    return TSDRGPU_OK;
}
extern "C" int tsdrgpu_lane_wait(tsdrgpu_t *g, int lane, tsdrgpu_event_t *ev)
{
    if (!g || !ev || !lane_stream(g, lane)) return g ? tsdr_fail(g, TSDRGPU_EINVAL, "tsdrgpu_lane_wait", "bad lane") : TSDRGPU_EINVAL;
    HIP_TRY(g, hipStreamWaitEvent(lane_stream(g, lane), (hipEvent_t)ev, 0));
    return TSDRGPU_OK;
}
extern "C" int tsdrgpu_event_sync(tsdrgpu_t *g, tsdrgpu_event_t *ev)
{
    if (!g || !ev) return TSDRGPU_EINVAL;
    HIP_TRY(g, hipEventSynchronize((hipEvent_t)ev));
    return TSDRGPU_OK;
}
extern "C" int tsdrgpu_event_done(tsdrgpu_t *g, tsdrgpu_event_t *ev)
{
    if (!g || !ev) return TSDRGPU_EINVAL;
    const hipError_t e = hipEventQuery((hipEvent_t)ev);
    if (e == hipSuccess) return 1;
    if (e == hipErrorNotReady) return 0;
    return tsdr_fail(g, TSDRGPU_EHIP, "hipEventQuery", hipGetErrorString(e));
}
extern "C" int tsdrgpu_lane_sync(tsdrgpu_t *g, int lane)
{
    if (!g || !lane_stream(g, lane)) return TSDRGPU_EINVAL;
    HIP_TRY(g, hipStreamSynchronize(lane_stream(g, lane)));
    return TSDRGPU_OK;
}
extern "C" int tsdrgpu_upload_lane(tsdrgpu_t *g, void *d_dst, const void *h_src, size_t bytes)
{
    if (!g) return TSDRGPU_EINVAL;
    if (bytes) HIP_TRY(g, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, g->up));
    return TSDRGPU_OK;
}
extern "C" int tsdrgpu_download_lane(tsdrgpu_t *g, void *h_dst, const void *d_src, size_t bytes)
{
    if (!g) return TSDRGPU_EINVAL;
    if (bytes) HIP_TRY(g, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, g->down));
    return TSDRGPU_OK;
}
extern "C" int tsdrgpu_host_register(tsdrgpu_t *g, void *h_ptr, size_t bytes)
{
    if (!g || !h_ptr || !bytes) return TSDRGPU_EINVAL;
    HIP_TRY(g, hipSetDevice(g->device));
    const hipError_t e = hipHostRegister(h_ptr, bytes, hipHostRegisterPortable);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // not sticky: the caller falls back to a bounce buffer
        return tsdr_fail(g, TSDRGPU_EHIP, "hipHostRegister", hipGetErrorString(e));
    }
    return TSDRGPU_OK;
}
extern "C" int tsdrgpu_host_unregister(tsdrgpu_t *g, void *h_ptr)
{
    if (!g || !h_ptr) return TSDRGPU_EINVAL;
    const hipError_t e = hipHostUnregister(h_ptr);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return tsdr_fail(g, TSDRGPU_EHIP, "hipHostUnregister", hipGetErrorString(e));
    }
    return TSDRGPU_OK;
}

extern "C" int tsdrgpu_timer_start(tsdrgpu_t *g)
{
    if (!g) return TSDRGPU_EINVAL;
    HIP_TRY(g, hipEventRecord(g->t0, g->stream));
    return TSDRGPU_OK;
}
extern "C" int tsdrgpu_timer_stop_ms(tsdrgpu_t *g, float *ms)
{
    if (!g || !ms) return TSDRGPU_EINVAL;
    HIP_TRY(g, hipEventRecord(g->t1, g->stream));
    HIP_TRY(g, hipEventSynchronize(g->t1));
    HIP_TRY(g, hipEventElapsedTime(ms, g->t0, g->t1));
    return TSDRGPU_OK;
}

// ---------------------------------------------------------------------------
// event profiler
// ---------------------------------------------------------------------------
static const char *const kStageNames[PROF_COUNT] = {"k_demod", "k_rs_tail+k_rs_chain", "k_rs_area", "k_rs_nearest",
                                                    "k_frame_stats", "k_frame_reduce", "k_chain", "k_frame_pass",
                                                    "k_fft_lds", "k_ac_mid", "k_ac_cols", "k_ac_rows", "k_accumulate", "superb_misc", "k_argmax", "extras"};

void prof_pair(tsdrgpu_t *g, int stage, hipEvent_t *a, hipEvent_t *b)
{
    *a = *b = nullptr;
    if (!g || !g->prof_on) return;
    if (g->nspans == g->cap_spans) {
        const int cap = g->cap_spans ? g->cap_spans * 2 : 1024;
        ProfSpan *n = (ProfSpan *)realloc(g->spans, sizeof(ProfSpan) * cap);
        if (!n) return;
        for (int i = g->cap_spans; i < cap; i++) n[i].a = n[i].b = nullptr;
        g->spans = n;
        g->cap_spans = cap;
    }
    ProfSpan &sp = g->spans[g->nspans];
    if (!sp.a && (hipEventCreate(&sp.a) != hipSuccess || hipEventCreate(&sp.b) != hipSuccess)) return;
    sp.stage = stage;
    *a = sp.a;
    *b = sp.b;
    g->nspans++;
}

extern "C" int tsdrgpu_profile_begin(tsdrgpu_t *g)
{
    if (!g) return TSDRGPU_EINVAL;
    HIP_TRY(g, hipStreamSynchronize(g->stream));
    HIP_TRY(g, hipStreamSynchronize(g->stream2));
    HIP_TRY(g, hipStreamSynchronize(g->bg));
    g->nspans = 0;
    g->prof_on = 1;
    return TSDRGPU_OK;
}

extern "C" int tsdrgpu_profile_end(tsdrgpu_t *g, tsdrgpu_profile_entry_t *h_entries, int max_entries, int *h_count)
{
    if (!g) return TSDRGPU_EINVAL;
    g->prof_on = 0;
    HIP_TRY(g, hipStreamSynchronize(g->stream));
    HIP_TRY(g, hipStreamSynchronize(g->stream2));
    HIP_TRY(g, hipStreamSynchronize(g->bg));
    double total[PROF_COUNT] = {0};
    int launches[PROF_COUNT] = {0};
    for (int i = 0; i < g->nspans; i++) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, g->spans[i].a, g->spans[i].b) == hipSuccess) {
            total[g->spans[i].stage] += ms;
            launches[g->spans[i].stage]++;
        }
    }
    int n = 0;
    for (int s = 0; s < PROF_COUNT && n < max_entries; s++) {
        if (!launches[s]) continue;
        if (h_entries) {
            snprintf(h_entries[n].name, sizeof(h_entries[n].name), "%s", kStageNames[s]);
            h_entries[n].total_ms = total[s];
            h_entries[n].launches = launches[s];
        }
        n++;
    }
    if (h_count) *h_count = n;
    g->nspans = 0;
    return TSDRGPU_OK;
}

// ---------------------------------------------------------------------------
// staging ring
// ---------------------------------------------------------------------------
int staging_init(tsdrgpu_t *g, StagingRing *r)
{
    memset(r, 0, sizeof(*r));
    for (int i = 0; i < StagingRing::SLOTS; i++) HIP_TRY(g, hipEventCreateWithFlags(&r->ev[i], hipEventDisableTiming));
    return TSDRGPU_OK;
}
void staging_free(StagingRing *r)
{
    for (int i = 0; i < StagingRing::SLOTS; i++) {
        if (r->h[i]) hipHostFree(r->h[i]);
        if (r->d[i]) hipFree(r->d[i]);
        if (r->ev[i]) hipEventDestroy(r->ev[i]);
    }
    memset(r, 0, sizeof(*r));
}
int staging_acquire(tsdrgpu_t *g, StagingRing *r, size_t bytes)
{
    const int s = r->next;
    r->next = (r->next + 1) % StagingRing::SLOTS;
    if (r->used[s]) {
        HIP_TRY(g, hipEventSynchronize(r->ev[s]));
        r->used[s] = false;
    }
    if (r->cap[s] < bytes) {
        if (r->h[s]) hipHostFree(r->h[s]);
        if (r->d[s]) hipFree(r->d[s]);
        r->h[s] = r->d[s] = nullptr;
        r->cap[s] = 0;
        size_t cap = bytes < 4096 ? 4096 : bytes + bytes / 2;
        if (hipHostMalloc(&r->h[s], cap, hipHostMallocDefault) != hipSuccess) return tsdr_fail(g, TSDRGPU_ENOMEM, "staging", "pinned");
        if (hipMalloc(&r->d[s], cap) != hipSuccess) return tsdr_fail(g, TSDRGPU_ENOMEM, "staging", "device");
        r->cap[s] = cap;
    }
    return s;
}
// A few kilobytes of chunk table per call: copied by a kernel that reads the pinned host buffer directly.  (As a
// hipMemcpyAsync it went through a copy engine, and the kernels behind it in the lane started ~29 us later —
// rocprofv3 timeline of bench.py, scripts/bench_gaps.py.)
__global__ __launch_bounds__(256) void k_stage_copy(const unsigned *__restrict__ h, unsigned *__restrict__ d, unsigned n)
{
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) d[i] = h[i];
}
int staging_push(tsdrgpu_t *g, StagingRing *r, int slot, size_t bytes)
{
    if (bytes > (1u << 20) || (bytes & 3)) {
        HIP_TRY(g, hipMemcpyAsync(r->d[slot], r->h[slot], bytes, hipMemcpyHostToDevice, g->stream));
        return TSDRGPU_OK;
    }
    const unsigned n = (unsigned)(bytes / 4);
    hipLaunchKernelGGL(k_stage_copy, dim3((n + 255) / 256 < 64 ? (n + 255) / 256 : 64), dim3(256), 0, g->stream, (const unsigned *)r->h[slot],
                       (unsigned *)r->d[slot], n);
    HIP_TRY(g, hipGetLastError());
    return TSDRGPU_OK;
}
int staging_release(tsdrgpu_t *g, StagingRing *r, int slot)
{
    HIP_TRY(g, hipEventRecord(r->ev[slot], g->stream));
    r->used[slot] = true;
    return TSDRGPU_OK;
}

// ---------------------------------------------------------------------------
// a1  AM demodulation: out[i] = sqrtf(I*I + Q*Q)     TSDRLibrary.c:244-262
// HBM-bound: 8 B read + 4 B written per sample.  16-byte loads/stores when the
// pointers allow; products, sum and sqrt are separate correctly-rounded f32
// operations (-ffp-contract=off), so the result is bit-identical to the CPU's.
// ---------------------------------------------------------------------------
// am_demod's sqrtf(re*re + im*im), TSDRLibrary.c:244-262.  hipcc's correctly rounded sqrtf is 21 instructions: v_sqrt_f32 (1 ulp),
// the two-sided correction (the neighbours' residuals by fma, two selects), a rescaling by 2^32 for arguments below 2^-96 and
// a class test for 0 / inf — the last two a third of it, for arguments a sample stream practically never holds.  When every
// lane of the wave has 2^-96 <= x < inf (one subtract and one unsigned compare on the bits, one ballot) the same correction
// runs bare: the same instructions on the same values as the library routine, so the same bits (the resampler's time is its
// VALU count).  Any other wave — zeros of an int8 recording, subnormal amplitudes, infinities — takes sqrtf as before.
__device__ __forceinline__ float demod1(float re, float im)
{
    const float x = re * re + im * im;
    const unsigned b = __float_as_uint(x);
    const bool plain = (b - 0x0f800000u) < (0x7f800000u - 0x0f800000u);
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(!plain) == 0ull, 1)) {
        float s = __builtin_amdgcn_sqrtf(x);
        const float sm = __uint_as_float(__float_as_uint(s) - 1u), sp = __uint_as_float(__float_as_uint(s) + 1u);
        const float rm = __builtin_fmaf(-sm, s, x), rp = __builtin_fmaf(-sp, s, x);
        s = (rm <= 0.0f) ? sm : s;
        s = (rp > 0.0f) ? sp : s;
        return s;
    }
    return sqrtf(x);
}

__global__ __launch_bounds__(256) void k_demod_vec4(const float4 *__restrict__ iq, float4 *__restrict__ out, long long nquads)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nquads; i += stride) {
        const float4 a = iq[2 * i];
        const float4 b = iq[2 * i + 1];
        float4 o;
        o.x = demod1(a.x, a.y);
        o.y = demod1(a.z, a.w);
        o.z = demod1(b.x, b.y);
        o.w = demod1(b.z, b.w);
        out[i] = o;
    }
}

__global__ __launch_bounds__(256) void k_demod_scalar(const float2 *__restrict__ iq, float *__restrict__ out, long long first, long long n)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = first + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float2 a = iq[i];
        out[i] = demod1(a.x, a.y);
    }
}

static unsigned stream_grid(long long work_items, unsigned block, const tsdrgpu_t *g)
{
    long long blocks = (work_items + block - 1) / block;
    const long long cap = (long long)g->prop.multiProcessorCount * 16;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

extern "C" int tsdrgpu_am_demod(tsdrgpu_t *g, const float *d_iq, float *d_out, int64_t n)
{
    if (!g || !d_iq || !d_out || n < 0) return TSDRGPU_EINVAL;
    if (n == 0) return TSDRGPU_OK;
    long long done = 0;
    if ((((uintptr_t)d_iq) & 15) == 0 && (((uintptr_t)d_out) & 15) == 0 && n >= 4) {
        const long long nquads = n / 4;
        TSDR_LAUNCH(g, PROF_DEMOD, g->stream, k_demod_vec4, stream_grid(nquads, 256, g), 256, (const float4 *)d_iq, (float4 *)d_out, nquads);
        KERNEL_CHECK(g, "k_demod_vec4");
        done = nquads * 4;
    }
    if (done < n) {
        TSDR_LAUNCH(g, PROF_DEMOD, g->stream, k_demod_scalar, stream_grid(n - done, 256, g), 256, (const float2 *)d_iq, d_out, done, n);
        KERNEL_CHECK(g, "k_demod_scalar");
    }
    return TSDRGPU_OK;
}

// ---------------------------------------------------------------------------
// a2  fractional area resampler                        dsp.c:250-307
// One thread per output pixel; the closed form in resample_math.h reproduces
// the sequential loop's f64 expressions.  A chunk table (host-built, because the
// per-chunk phase recurrence is pure f64 bookkeeping, dsp.c:262,272,306) tells
// each block which dsp_resample_process call it belongs to.
// ---------------------------------------------------------------------------
// Frame tracking records.  RsChunkFrame: where a chunk's first pixel falls (frame index counted from
// the frame the call starts in, offset inside that frame).  RsBlockMM: min/max of the non-sentinel
// pixels one wave of a k_rs_area workgroup emitted, split over the (at most two, frames hold >= 4096
// pixels) frames the workgroup's 2048-pixel span touches; f0 < 0 = the workgroup emitted nothing.
struct RsChunkFrame {
    long long rem;
    int f;
    int pad;
};
struct RsBlockMM {
    int f0;
    float mn0, mx0, mn1, mx1;
};
struct RsFrameRange {
    int c_lo, c_hi;  // chunks that overlap the frame
};

// wave64 min/max with DPP row shifts + row broadcasts, result in lane 63.  The values travel as order-preserving
// integer keys (sign-magnitude -> two's complement: the floats' order, -0 below +0, no NaN among pixels): an integer
// min/max takes the DPP operand itself — six instructions per reduction — where the float form cost a v_mov_dpp, a
// canonicalising v_max and the min per step (27 instructions; the frame tracking's two reductions were half of what it
// added to the resampler's instruction count).
__device__ __forceinline__ int rs_fkey(float v)
{
    const int b = __builtin_bit_cast(int, v);
    return b ^ ((b >> 31) & 0x7fffffff);
}
__device__ __forceinline__ float rs_funkey(int k) { return __builtin_bit_cast(float, k ^ ((k >> 31) & 0x7fffffff)); }
template <int CTRL>
__device__ __forceinline__ int rs_dpp(int v, int identity)  // lanes without a source lane read the operation's identity
{
    return __builtin_amdgcn_update_dpp(identity, v, CTRL, 0xf, 0xf, false);
}
__device__ __forceinline__ float rs_wave_min(float v)
{
    int k = rs_fkey(v);
    k = min(k, rs_dpp<0x111>(k, 0x7fffffff));  // row_shr:1
    k = min(k, rs_dpp<0x112>(k, 0x7fffffff));  // row_shr:2
    k = min(k, rs_dpp<0x114>(k, 0x7fffffff));  // row_shr:4
    k = min(k, rs_dpp<0x118>(k, 0x7fffffff));  // row_shr:8   -> lane 15 of each row holds the row's min
    k = min(k, rs_dpp<0x142>(k, 0x7fffffff));  // row_bcast:15
    k = min(k, rs_dpp<0x143>(k, 0x7fffffff));  // row_bcast:31 -> lane 63 holds the wave's min
    return rs_funkey(k);
}
__device__ __forceinline__ float rs_wave_max(float v)
{
    int k = rs_fkey(v);
    k = max(k, rs_dpp<0x111>(k, (int)0x80000000));
    k = max(k, rs_dpp<0x112>(k, (int)0x80000000));
    k = max(k, rs_dpp<0x114>(k, (int)0x80000000));
    k = max(k, rs_dpp<0x118>(k, (int)0x80000000));
    k = max(k, rs_dpp<0x142>(k, (int)0x80000000));
    k = max(k, rs_dpp<0x143>(k, (int)0x80000000));
    return rs_funkey(k);
}
// frame index / offset of the pixel `add` pixels after (f, rem)
__device__ __forceinline__ void rs_frame_of(long long P, int f, long long rem, long long add, int *fo, long long *ro)
{
    long long pos = rem + add;
    if (pos >= P) {
        if (pos < 2 * P) { pos -= P; f += 1; }
        else { const long long q = pos / P; pos -= q * P; f += (int)q; }
    }
    *fo = f;
    *ro = pos;
}

struct tsdrgpu_resampler {
    tsdrgpu_t *g;
    double offset;      // dsp_resample_t.offset (host side: data independent)
    double *d_contrib;  // dsp_resample_t.contrib (device side: data dependent)
    double *d_cin;      // per-chunk incoming contrib
    double *d_tail;     // per-chunk outgoing contrib when it does not depend on the incoming one
    unsigned char *d_need;
    int cap_chunks;
    StagingRing ring;
    // frame tracking (tsdrgpu_resampler_track_frames): per-frame min/max of the emitted pixels
    long long frame_pixels;  // 0 = off
    long long phase;         // pixels of the current incomplete frame emitted by earlier calls
    RsBlockMM *d_slots;      // one record per k_rs_area workgroup
    size_t cap_slots;
    float *d_fmin, *d_fmax;  // frames touched by the last call, in order
    int cap_frames;
    float *d_carry;          // [2][2]: min/max of the incomplete frame, double buffered by call parity
    int parity;
    int last_complete;       // frames completed by the last call (-1: no tracked call yet)
};

template <bool IQ>
struct SampleLoad {
    const float *base;
    __device__ __forceinline__ float operator()(int j) const
    {
        if (IQ) {
            const float2 s = ((const float2 *)base)[j];
            return demod1(s.x, s.y);
        }
        return base[j];
    }
};

// phase 1: every chunk's outgoing contrib, in parallel
template <bool IQ>
__global__ void k_rs_tail(const RsChunk *__restrict__ chunks, int nchunks, double r, double rinv, const float *__restrict__ in,
                          double *__restrict__ tail, unsigned char *__restrict__ need)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nchunks) return;
    const RsChunk ch = chunks[c];
    RsGeom g;
    g.r = r;
    g.rinv = rinv;
    g.o = ch.o;
    g.size = ch.size;
    SampleLoad<IQ> ld{in + (IQ ? 2 : 1) * ch.in_off};
    bool used = false;
    tail[c] = rs_contrib_before(g, (int)ch.size, 0.0, ld, &used);
    need[c] = used ? 1 : 0;
}

// phase 2: chain them.  Normally every chunk finishes at least one pixel, so
// its outgoing contrib does not depend on the incoming one and the chain is a
// shift; chunks that do depend on it (need[c]) force the scalar replay.
template <bool IQ>
__global__ __launch_bounds__(256) void k_rs_chain(const RsChunk *__restrict__ chunks, int nchunks, double r, double rinv,
                                                  const float *__restrict__ in, const double *__restrict__ tail,
                                                  const unsigned char *__restrict__ need, double *__restrict__ cin,
                                                  double *__restrict__ contrib_state)
{
    int any = 0;
    for (int c = threadIdx.x; c < nchunks; c += blockDim.x) any |= need[c];
    any = __syncthreads_or(any);
    if (!any) {
        const double first = *contrib_state;
        for (int c = threadIdx.x; c < nchunks; c += blockDim.x) cin[c] = c ? tail[c - 1] : first;
        __syncthreads();
        if (threadIdx.x == 0) *contrib_state = tail[nchunks - 1];
        return;
    }
    if (threadIdx.x != 0) return;
    double c_in = *contrib_state;
    for (int c = 0; c < nchunks; c++) {
        cin[c] = c_in;
        if (need[c]) {
            const RsChunk ch = chunks[c];
            RsGeom g;
            g.r = r;
    g.rinv = rinv;
            g.o = ch.o;
            g.size = ch.size;
            SampleLoad<IQ> ld{in + (IQ ? 2 : 1) * ch.in_off};
            c_in = rs_contrib_before(g, (int)ch.size, c_in, ld);
        } else {
            c_in = tail[c];
        }
    }
    *contrib_state = c_in;
}

// Each thread produces eight consecutive pixels that start on a 16-byte boundary
// of the output stream (two dwordx4 stores) by replaying the reference loop over
// the few samples that touch them (rs_area_group).  Groups cut by the chunk's
// ends fall back to dword stores.
#define RS_NPIX 8
template <bool IQ, bool MM>
__global__ __launch_bounds__(256) void k_rs_area(const RsChunk *__restrict__ chunks, double r, double rinv, const float *__restrict__ in,
                                                 const double *__restrict__ cin, float *__restrict__ out,
                                                 const RsChunkFrame *__restrict__ cframes, long long P, RsBlockMM *__restrict__ slots)
{
    const RsChunk ch = chunks[blockIdx.y];
    RsGeom g;
    g.r = r;
    g.rinv = rinv;
    g.o = ch.o;
    g.size = ch.size;
    SampleLoad<IQ> ld{in + (IQ ? 2 : 1) * ch.in_off};
    const double c_in = cin[blockIdx.y];
    float *dst = out + ch.out_off;
    const int mis = (int)((((uintptr_t)dst) >> 2) & 3);  // dst's offset inside its 16-byte group
    const int n_out = (int)ch.n_out;
    const int ngroups = (n_out + mis + RS_NPIX - 1) / RS_NPIX;
    // emitted pixels are staged in LDS as [k][thread] (a lane only ever touches its own column, so no
    // barrier is needed and equal k across lanes is conflict free); a dynamic register index would
    // cost a compare+select per register instead of one ds_write
    __shared__ float stage[RS_NPIX][256];
    // frame tracking: min/max of this workgroup's pixels for the frame its first pixel lies in (fb) and
    // the next one.  The host sizes the grid so that the loop below runs at most once per thread.
    float mn0 = INFINITY, mx0 = -INFINITY, mn1 = INFINITY, mx1 = -INFINITY;
    int fb = -1;          // frame of the workgroup's first pixel
    int pbf = 0;          // that pixel
    int to_b = 0x3fffffff;  // pixels from it to the next frame boundary: pixel p is in frame fb iff p - pbf < to_b
    if (MM) {
        const int pb = RS_NPIX * (int)(blockIdx.x * blockDim.x) - mis;
        pbf = pb < 0 ? 0 : pb;
        if (pb < n_out) {
            const RsChunkFrame cf = cframes[blockIdx.y];
            long long remb;
            rs_frame_of(P, cf.f, cf.rem, pbf, &fb, &remb);
            const long long tb = P - remb;
            to_b = tb > 0x3fffffffLL ? 0x3fffffff : (int)tb;
        }
    }
    const bool crosses = MM && to_b < RS_NPIX * 256 + RS_NPIX;  // uniform: the span reaches into frame fb+1
    for (int grp = blockIdx.x * blockDim.x + threadIdx.x; grp < ngroups; grp += gridDim.x * blockDim.x) {
        const int p0 = RS_NPIX * grp - mis;
#pragma unroll
        for (int k = 0; k < RS_NPIX; k++) stage[k][threadIdx.x] = 0.0f;
        rs_area_group<RS_NPIX>(g, p0, n_out, c_in, ld, [&](int k, float val) { stage[k][threadIdx.x] = val; });
        float v[RS_NPIX];
#pragma unroll
        for (int k = 0; k < RS_NPIX; k++) v[k] = stage[k][threadIdx.x];
        if (MM) {
            // usual case: a whole group inside one frame and no sentinel among its pixels -> min/max of
            // the eight values (v_min3/v_max3), no per-pixel tests
            bool fast = !crosses && p0 >= 0 && p0 + RS_NPIX <= n_out;
            if (fast) {
                const float m = fminf(fminf(fminf(v[0], v[1]), fminf(v[2], v[3])), fminf(fminf(v[4], v[5]), fminf(v[6], v[7])));
                const float M = fmaxf(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])), fmaxf(fmaxf(v[4], v[5]), fmaxf(v[6], v[7])));
                if (M > 250.0f || m < -250.0f) {
                    fast = false;
                } else {
                    mn0 = fminf(mn0, m);
                    mx0 = fmaxf(mx0, M);
                }
            }
            if (!fast) {
#pragma unroll
                for (int k = 0; k < RS_NPIX; k++) {
                    const int p = p0 + k;
                    const float val = v[k];
                    const bool ok = p >= 0 && p < n_out && !((val > 250.0f) || (val < -250.0f));  // dsp.c:57
                    const bool first = p - pbf < to_b;
                    mn0 = fminf(mn0, (ok && first) ? val : INFINITY);
                    mx0 = fmaxf(mx0, (ok && first) ? val : -INFINITY);
                    mn1 = fminf(mn1, (ok && !first) ? val : INFINITY);
                    mx1 = fmaxf(mx1, (ok && !first) ? val : -INFINITY);
                }
            }
        }
        if (p0 >= 0 && p0 + RS_NPIX <= n_out) {
            *reinterpret_cast<float4 *>(dst + p0) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4 *>(dst + p0 + 4) = make_float4(v[4], v[5], v[6], v[7]);
        } else {
#pragma unroll
            for (int k = 0; k < RS_NPIX; k++)
                if (p0 + k >= 0 && p0 + k < n_out) dst[p0 + k] = v[k];
        }
    }
    if (MM) {
        // one record per wave (no workgroup barrier, no LDS): lane 63 holds the DPP reductions
        mn0 = rs_wave_min(mn0); mx0 = rs_wave_max(mx0);
        if (crosses) { mn1 = rs_wave_min(mn1); mx1 = rs_wave_max(mx1); }
        if ((threadIdx.x & 63) == 63) {
            RsBlockMM o;
            o.f0 = fb;
            o.mn0 = mn0; o.mx0 = mx0; o.mn1 = mn1; o.mx1 = mx1;
            slots[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6)] = o;
        }
    }
}

// ---------------------------------------------------------------------------
// Row-band form (tsdrgpu_resample_band): the same chunks, but only the pixels of rows
// [y0, y0 + rows) of every frame are computed and stored, into a band buffer that holds frame after frame of
// rows*width pixels.  A band entry = the part of one chunk's output that falls into the band of one frame (a chunk is
// shorter than a frame, so at most two per chunk).  Every thread enters the reference loop at its own pixel group
// through the closed form (rs_area_group), exactly like k_rs_area, so the values are those of the full run bit for bit.
// ---------------------------------------------------------------------------
struct RsBandEntry {
    int chunk;          // index into the chunk table
    int p_lo, p_hi;     // chunk-relative pixels [p_lo, p_hi) of this entry
    int pad;
    long long dst_off;  // where pixel p_lo goes in the band buffer
};

template <bool IQ>
__global__ __launch_bounds__(256) void k_rs_area_band(const RsChunk *__restrict__ chunks, const RsBandEntry *__restrict__ ents, double r,
                                                      double rinv, const float *__restrict__ in, const double *__restrict__ cin,
                                                      float *__restrict__ out)
{
    const RsBandEntry e = ents[blockIdx.y];
    const RsChunk ch = chunks[e.chunk];
    RsGeom g;
    g.r = r;
    g.rinv = rinv;
    g.o = ch.o;
    g.size = ch.size;
    SampleLoad<IQ> ld{in + (IQ ? 2 : 1) * ch.in_off};
    const double c_in = cin[e.chunk];
    float *dst = out + e.dst_off - e.p_lo;  // pixel p of the chunk goes to dst[p] (only p_lo <= p < p_hi is ever touched)
    const int mis = (int)((((uintptr_t)dst) >> 2) & 3);
    const int n_out = (int)ch.n_out;
    const int g_lo = (e.p_lo + mis) / RS_NPIX, g_hi = (e.p_hi + mis + RS_NPIX - 1) / RS_NPIX;
    __shared__ float stage[RS_NPIX][256];
    for (int grp = g_lo + (int)(blockIdx.x * blockDim.x + threadIdx.x); grp < g_hi; grp += (int)(gridDim.x * blockDim.x)) {
        const int p0 = RS_NPIX * grp - mis;
#pragma unroll
        for (int k = 0; k < RS_NPIX; k++) stage[k][threadIdx.x] = 0.0f;
        rs_area_group<RS_NPIX>(g, p0, n_out, c_in, ld, [&](int k, float val) { stage[k][threadIdx.x] = val; });
        float v[RS_NPIX];
#pragma unroll
        for (int k = 0; k < RS_NPIX; k++) v[k] = stage[k][threadIdx.x];
        if (p0 >= e.p_lo && p0 + RS_NPIX <= e.p_hi) {
            *reinterpret_cast<float4 *>(dst + p0) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4 *>(dst + p0 + 4) = make_float4(v[4], v[5], v[6], v[7]);
        } else {
#pragma unroll
            for (int k = 0; k < RS_NPIX; k++)
                if (p0 + k >= e.p_lo && p0 + k < e.p_hi) dst[p0 + k] = v[k];
        }
    }
}

// ---------------------------------------------------------------------------
// k_rs_area_up: the same resampler, one lane per INPUT sample (resample_math.h, "sample-parallel form").
// k_rs_area replays the loop body per 8-pixel group, which costs ~60 VALU instructions per pixel in branchy,
// divergent code (the kernel was issue-bound at half the HBM rate).  Here every per-sample expression is
// evaluated once, branch free, and the three values a lane needs from the previous sample arrive through a
// DPP wave shift: lane l of a wave holds sample base + l - 2, lanes 0 and 1 are ghosts that only feed lane 2.
// A workgroup (4 waves x `rounds` x 62 samples) writes its pixels [pix_in(sA), pix_in(sB)) into an LDS tile laid
// out with the output's 16-byte phase, then stores the tile with dwordx4.  Used when 1 <= r <= 8 (the
// reference's geometry always upsamples by ~2); anything else runs k_rs_area.
// ---------------------------------------------------------------------------
#define RSU_LANES 62     // samples per wave and round
#define RSU_BATCH 4     // rounds whose loads are in flight together
#define RSU_TILE 2048    // pixels per workgroup the host may plan for (+8 floats of slack in LDS)
__device__ __forceinline__ int rsu_shr1(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false); }
__device__ __forceinline__ double rsu_shr1(double v)
{
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = rsu_shr1((int)(b & 0xffffffffLL)), hi = rsu_shr1((int)(b >> 32));
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}

// MM: frame tracking (tsdrgpu_resampler_track_frames) — every wave also leaves the min/max of the non-sentinel pixels it
// wrote, split over the (at most two) frames the workgroup's span touches, exactly like k_rs_area (RsBlockMM).
// BAND (tsdrgpu_resample_band): only the pixels of one row band of every frame are wanted.  A band is a CONTIGUOUS range
// [b_lo, b_hi) of each frame's pixels, so a workgroup whose span misses it returns before it loads a sample (the work shrinks
// with the band), and the others store the part of their tile that lies inside — frame j of the call to band + j * Pb.
struct RsBandGeom {
    long long b_lo, b_hi, Pb;  // the band's pixel range inside a frame, and its length
};
template <bool IQ, bool MM, bool BAND = false>
__global__ __launch_bounds__(256) void k_rs_area_up(const RsChunk *__restrict__ chunks, double r, double rinv, const float *__restrict__ in,
                                                    const double *__restrict__ cin, float *__restrict__ out, int rounds, int maxc,
                                                    const RsChunkFrame *__restrict__ cframes, long long P, RsBlockMM *__restrict__ slots,
                                                    RsBandGeom bd)
{
    const RsChunk ch = chunks[blockIdx.y];
    RsGeom g;
    g.r = r;
    g.rinv = rinv;
    g.o = ch.o;
    g.size = ch.size;
    const int size = (int)ch.size, n_out = (int)ch.n_out;
    const int S = 4 * RSU_LANES * rounds;
    const int sA = (int)blockIdx.x * S;
    const int sB = (sA + S < size) ? sA + S : size;
    const int PA = (sA < size) ? (int)rs_pix_in(g, sA) : 0;
    int PE = (sA < size) ? (int)rs_pix_in(g, sB) : 0;  // end of what the samples [sA, sB) store
    PE = PE < n_out ? PE : n_out;
    const int PB = (sB >= size) ? n_out : PE;        // end of what this workgroup writes (zeros behind PE)
    if (sA >= size || n_out <= 0 || PA >= PB) {       // nothing to write
        if (MM && (threadIdx.x & 63) == 63) {
            RsBlockMM o;
            o.f0 = -1;
            o.mn0 = o.mn1 = INFINITY;
            o.mx0 = o.mx1 = -INFINITY;
            slots[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6)] = o;
        }
        return;
    }
    // frame tracking: frame of the workgroup's first pixel (fb) and how many of its pixels still belong to it (up
    // here, so that the record's load is long done when the copy-out needs it)
    float mn0 = INFINITY, mx0 = -INFINITY, mn1 = INFINITY, mx1 = -INFINITY;
    int fb = -1, to_b = 0x3fffffff;
    if (MM) {
        const RsChunkFrame cf = cframes[blockIdx.y];
        long long remb;
        rs_frame_of(P, cf.f, cf.rem, PA, &fb, &remb);
        const long long tb = P - remb;
        to_b = tb > 0x3fffffffLL ? 0x3fffffff : (int)tb;  // pixel p is in frame fb iff p - PA < to_b
    }
    const bool crosses = MM && PB - PA > to_b;  // uniform: the span reaches into frame fb + 1
    float *dst = out + ch.out_off;
    // band form: chunk pixel p of frame fb lives at dstA[p], of frame fb + 1 at dstB[p]; [lo0, hi0) / [lo1, hi1) = the chunk
    // pixels of the two frames that lie inside the band
    float *dstB = nullptr;
    int lo0 = 0, hi0 = 0, lo1 = 0, hi1 = 0;
    if (BAND) {
        const RsChunkFrame cf = cframes[blockIdx.y];
        int fbb;
        long long remb;
        rs_frame_of(P, cf.f, cf.rem, PA, &fbb, &remb);
        const long long n = PB - PA, tb = P - remb;  // tb: pixels of the span that still belong to frame fbb
        long long a0 = bd.b_lo - remb, e0 = bd.b_hi - remb;
        a0 = a0 < 0 ? 0 : a0;
        e0 = e0 > n ? n : e0;
        e0 = e0 > tb ? tb : e0;
        long long a1 = tb + bd.b_lo, e1 = tb + bd.b_hi;
        e1 = e1 > n ? n : e1;
        if (a0 >= e0 && a1 >= e1) {  // nothing of this workgroup's span lies in the band
            if (MM && (threadIdx.x & 63) == 63) {  // (tracked: an empty record, like a workgroup with nothing to write)
                RsBlockMM o;
                o.f0 = -1;
                o.mn0 = o.mn1 = INFINITY;
                o.mx0 = o.mx1 = -INFINITY;
                slots[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6)] = o;
            }
            return;
        }
        lo0 = PA + (int)a0; hi0 = a0 < e0 ? PA + (int)e0 : lo0;
        lo1 = PA + (int)(a1 < e1 ? a1 : 0); hi1 = a1 < e1 ? PA + (int)e1 : lo1;
        dst = out + (long long)fbb * bd.Pb - bd.b_lo + remb - PA;
        dstB = dst + bd.Pb - P;
    }
    const int mis = (int)((((uintptr_t)(dst + PA)) >> 2) & 3);  // phase of pixel PA inside its 16-byte group
    __shared__ float4 tile4[(RSU_TILE + 8) / 4];
    float *tile = reinterpret_cast<float *>(tile4);
    if (PB - PA + mis > RSU_TILE + 8) __builtin_trap();  // the host plans `rounds` so that this cannot happen

    SampleLoad<IQ> ld{in + (IQ ? 2 : 1) * ch.in_off};
    const double c_in = cin[blockIdx.y];
    // (the wave index through readfirstlane: the compiler then keeps everything derived from it — a round's base sample, the
    // tests that only a chunk's first wave needs — on the scalar unit)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int toff = mis - PA;
    // rounds go in batches of RSU_BATCH: all of a batch's samples are requested before the first one is used
    for (int j0 = 0; j0 < rounds; j0 += RSU_BATCH) {
        // (the raw samples: demod1 holds a wave-uniform branch, and a branch between the requests would serialise them)
        float2 raw[RSU_BATCH];
#pragma unroll
        for (int jj = 0; jj < RSU_BATCH; jj++) {
            const int id = sA + (wave * rounds + j0 + jj) * RSU_LANES + lane - 2;
            const unsigned idc = (unsigned)max(0, min(id, size - 1));  // (size >= 1 here; an unsigned index spares the sign extension)
            if (IQ) raw[jj] = ((const float2 *)ld.base)[idc];
            else raw[jj] = make_float2(ld.base[idc], 0.f);
        }
#pragma unroll
        for (int jj = 0; jj < RSU_BATCH; jj++) {
            const int base = sA + (wave * rounds + j0 + jj) * RSU_LANES;
            if (j0 + jj >= rounds || base >= sB) break;  // wave-uniform
            const int id = base + lane - 2;
            const RsUpGeom a = rs_up_geom(g, id);
            const int pnext = (int)a.pnext;
            const float vf = IQ ? demod1(raw[jj].x, raw[jj].y) : raw[jj].x;
            const double val = (double)vf;
            const double tail = rs_up_tail(g, a, val);
            int pin = rsu_shr1(pnext);
            const double tail_prev = rsu_shr1(tail);
            // dsp.c:299-302 leaves contrib = 0.0 + tail; a demodulated sample is a square root, never -0.0, so its tail term is
            // never -0.0 either and the addition of 0.0 changes nothing (magnitude input keeps it)
            double contrib = IQ ? tail_prev : 0.0 + tail_prev;
            if (__builtin_expect(base == 0, 0)) {  // wave-uniform: only a chunk's first wave holds the samples id <= 0 (ghost lanes) and id == 0
                asm volatile("" ::: "memory");     // (a real branch: as selects these tests cost every round eight instructions)
                pin = (id <= 0) ? 0 : pin;
                if (id == 0) contrib = c_in;
            }
            const double pind = (double)pin;
            const bool fired = rs_up_fired(a, pind);
            const int fired_prev = rsu_shr1(fired ? 1 : 0);
            const bool real = lane >= 2 && id < sB;
            if (real && fired && id > 0 && !fired_prev) contrib = rs_contrib_before(g, id, c_in, ld);  // the exception when r >= 1
            const float first = rs_up_first(a, pind, contrib, val);
            // the sample's pixels [pin, pin + cnt) that lie below PE: `lim` of them, at consecutive tile slots.  maxc (uniform:
            // (int)r + 2) bounds cnt; the reference's geometry has r ~ 2, i.e. maxc = 3 or 4, so the first four stores are
            // straight-line — one compare each and an immediate offset — instead of a counted loop with the index arithmetic and
            // both comparisons per pixel (the kernel's time is its VALU count)
            const int cnt = real ? pnext - pin : 0;
            const int lim = min(cnt, PE - pin);
            float *tp = tile + (pin + toff);
            if (__builtin_expect(lim > 0, 1)) tp[0] = fired ? first : vf;
            if (__builtin_expect(lim > 1, 1)) tp[1] = vf;
            if (__builtin_expect(lim > 2, 1)) tp[2] = vf;
            if (lim > 3) tp[3] = vf;
            for (int c = 4; c < maxc; c++)  // r >= 3 only
                if (c < lim) tp[c] = vf;
        }
    }
    __syncthreads();
    const int ngroups = (PB - PA + mis + 3) >> 2;
    const bool zero_tail = PB > PE;  // uniform: only a chunk's last workgroup writes pixels the reference's loop never stores
    for (int grp = threadIdx.x; grp < ngroups; grp += 256) {
        const int p0 = PA - mis + 4 * grp;
        float4 q = tile4[grp];
        float v[4] = {q.x, q.y, q.z, q.w};
        if (zero_tail) {  // (the kernel's time is its VALU count: twelve instructions per group that all but one workgroup in ~80 can skip)
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (p0 + k >= PE) v[k] = 0.0f;
        }
        const bool whole = BAND ? (p0 >= lo0 && p0 + 4 <= hi0) : (p0 >= PA && p0 + 4 <= PB);
        if (whole) {
            *reinterpret_cast<float4 *>(dst + p0) = make_float4(v[0], v[1], v[2], v[3]);
        } else if (BAND) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int p = p0 + k;
                if (p >= lo0 && p < hi0) dst[p] = v[k];
                else if (p >= lo1 && p < hi1) dstB[p] = v[k];
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (p0 + k >= PA && p0 + k < PB) dst[p0 + k] = v[k];
        }
        if (MM) {
            // usual case: a whole group in a span that stays inside one frame, no sentinel among its pixels -> min/max
            // of the four values
            float m, M;
            if (IQ) {
                // Demodulated input: every pixel is the result of arithmetic (a square root or a blend), so it cannot be a
                // SIGNALLING NaN, and then the bare instructions are fminf / fmaxf (a quiet NaN operand is ignored, like the
                // reference's comparisons ignore it, dsp.c:57-60).  From fminf() the compiler also emits a canonicalising
                // v_max_f32 x, x per operand (IEEE mode quiets signalling NaNs first): 14 VALU instructions per group instead
                // of 8, in a kernel whose duration IS its VALU count (SQ counters: 6 waves per SIMD x 16 % each).
                float t;
                asm("v_min3_f32 %0, %1, %2, %3" : "=v"(t) : "v"(v[0]), "v"(v[1]), "v"(v[2]));
                asm("v_min_f32 %0, %1, %2" : "=v"(m) : "v"(t), "v"(v[3]));
                asm("v_max3_f32 %0, %1, %2, %3" : "=v"(t) : "v"(v[0]), "v"(v[1]), "v"(v[2]));
                asm("v_max_f32 %0, %1, %2" : "=v"(M) : "v"(t), "v"(v[3]));
            } else {
                m = fminf(fminf(v[0], v[1]), fminf(v[2], v[3]));
                M = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
            }
            if (whole && !crosses && !(M > 250.0f || m < -250.0f)) {
                if (IQ) {
                    asm("v_min_f32 %0, %1, %2" : "=v"(mn0) : "v"(mn0), "v"(m));
                    asm("v_max_f32 %0, %1, %2" : "=v"(mx0) : "v"(mx0), "v"(M));
                } else {
                    mn0 = fminf(mn0, m);
                    mx0 = fmaxf(mx0, M);
                }
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int p = p0 + k;
                    const float val = v[k];
                    // (band form: only the pixels this rank stores count — the frame's range is the max over the bands')
                    const bool inband = !BAND || (p >= lo0 && p < hi0) || (p >= lo1 && p < hi1);
                    const bool ok = p >= PA && p < PB && inband && !((val > 250.0f) || (val < -250.0f));  // dsp.c:57
                    const bool first = p - PA < to_b;
                    mn0 = fminf(mn0, (ok && first) ? val : INFINITY);
                    mx0 = fmaxf(mx0, (ok && first) ? val : -INFINITY);
                    mn1 = fminf(mn1, (ok && !first) ? val : INFINITY);
                    mx1 = fmaxf(mx1, (ok && !first) ? val : -INFINITY);
                }
            }
        }
    }
    if (MM) {
        mn0 = rs_wave_min(mn0); mx0 = rs_wave_max(mx0);
        if (crosses) { mn1 = rs_wave_min(mn1); mx1 = rs_wave_max(mx1); }
        if ((threadIdx.x & 63) == 63) {
            RsBlockMM o;
            o.f0 = fb;
            o.mn0 = mn0; o.mx0 = mx0; o.mn1 = mn1; o.mx1 = mx1;
            slots[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6)] = o;
        }
    }
}

// Frame tracking, second stage: one workgroup per frame touched by the call folds the records of
// the workgroups whose span overlaps it (chunk range from the host), joins frame 0 with the
// incomplete frame carried from the previous call and leaves the call's last, incomplete frame
// in the other carry slot.  min/max are order independent, so this is exact.
#define RSMM_T 1024  // a frame's ~7000 records are read by one workgroup: 7 dependent loads per lane instead of 29
__global__ __launch_bounds__(RSMM_T) void k_rs_minmax(const RsBlockMM *__restrict__ slots, int gx, const RsFrameRange *__restrict__ ranges,
                                                   int ntouched, int ncomplete, const float *__restrict__ carry_in,
                                                   float *__restrict__ carry_out, float *__restrict__ fmin_, float *__restrict__ fmax_)
{
    const int j = blockIdx.x;
    const RsFrameRange rg = ranges[j];
    float lo = INFINITY, hi = -INFINITY;
    const long long first = (long long)rg.c_lo * gx * 4, last = (long long)rg.c_hi * gx * 4;  // 4 wave records per workgroup
    // four records per lane requested together (the loop is a chain of load latencies otherwise)
    for (long long i0 = first + threadIdx.x; i0 < last; i0 += 4LL * blockDim.x) {
        RsBlockMM b[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const long long i = i0 + (long long)k * blockDim.x;
            b[k] = slots[i < last ? i : last - 1];
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (i0 + (long long)k * blockDim.x >= last) break;
            if (b[k].f0 == j) { lo = fminf(lo, b[k].mn0); hi = fmaxf(hi, b[k].mx0); }
            else if (b[k].f0 >= 0 && b[k].f0 + 1 == j) { lo = fminf(lo, b[k].mn1); hi = fmaxf(hi, b[k].mx1); }
        }
    }
    __shared__ float red[RSMM_T / 64][2];
    lo = rs_wave_min(lo);
    hi = rs_wave_max(hi);
    if ((threadIdx.x & 63) == 63) { red[threadIdx.x >> 6][0] = lo; red[threadIdx.x >> 6][1] = hi; }
    __syncthreads();
    if (threadIdx.x != 0) return;
    lo = red[0][0];
    hi = red[0][1];
    for (int w = 1; w < (int)(blockDim.x >> 6); w++) { lo = fminf(lo, red[w][0]); hi = fmaxf(hi, red[w][1]); }
    if (j == 0) { lo = fminf(lo, carry_in[0]); hi = fmaxf(hi, carry_in[1]); }
    fmin_[j] = lo;
    fmax_[j] = hi;
    if (j == ntouched - 1) {
        const bool incomplete = ntouched > ncomplete;
        carry_out[0] = incomplete ? lo : INFINITY;
        carry_out[1] = incomplete ? hi : -INFINITY;
    }
}

template <bool IQ>
__global__ __launch_bounds__(256) void k_rs_nearest(const RsChunk *__restrict__ chunks, const float *__restrict__ in,
                                                    float *__restrict__ out)
{
    const RsChunk ch = chunks[blockIdx.y];
    SampleLoad<IQ> ld{in + (IQ ? 2 : 1) * ch.in_off};
    float *dst = out + ch.out_off;
    for (unsigned p = blockIdx.x * blockDim.x + threadIdx.x; p < ch.n_out; p += gridDim.x * blockDim.x)
        dst[p] = ld((int)rs_nearest_src(ch.size, ch.n_out, p));
}

extern "C" int tsdrgpu_resampler_create(tsdrgpu_t *g, tsdrgpu_resampler_t **out)
{
    if (!g || !out) return TSDRGPU_EINVAL;
    tsdrgpu_resampler_t *rs = (tsdrgpu_resampler_t *)calloc(1, sizeof(*rs));
    if (!rs) return TSDRGPU_ENOMEM;
    rs->g = g;
    int rc = staging_init(g, &rs->ring);
    if (rc) { free(rs); return rc; }
    if (hipMalloc(&rs->d_contrib, sizeof(double)) != hipSuccess) { staging_free(&rs->ring); free(rs); return TSDRGPU_ENOMEM; }
    rs->last_complete = -1;
    *out = rs;
    return tsdrgpu_resampler_reset(rs);
}

extern "C" void tsdrgpu_resampler_destroy(tsdrgpu_resampler_t *rs)
{
    if (!rs) return;
    hipStreamSynchronize(rs->g->stream);
    staging_free(&rs->ring);
    hipFree(rs->d_contrib);
    hipFree(rs->d_cin);
    hipFree(rs->d_tail);
    hipFree(rs->d_need);
    hipFree(rs->d_slots);
    hipFree(rs->d_fmin);
    hipFree(rs->d_fmax);
    hipFree(rs->d_carry);
    free(rs);
}

extern "C" int tsdrgpu_resampler_setstate(tsdrgpu_resampler_t *rs, double contrib, double offset)
{
    if (!rs) return TSDRGPU_EINVAL;
    tsdrgpu_t *g = rs->g;
    rs->offset = offset;
    HIP_TRY(g, hipMemcpyAsync(rs->d_contrib, &contrib, sizeof(double), hipMemcpyHostToDevice, g->stream));
    HIP_TRY(g, hipStreamSynchronize(g->stream));  // `contrib` is a stack variable
    return TSDRGPU_OK;
}

extern "C" int tsdrgpu_resampler_reset(tsdrgpu_resampler_t *rs) { return tsdrgpu_resampler_setstate(rs, 0.0, 0.0); }

extern "C" int tsdrgpu_resampler_getstate(tsdrgpu_resampler_t *rs, double *contrib, double *offset)
{
    if (!rs) return TSDRGPU_EINVAL;
    tsdrgpu_t *g = rs->g;
    if (offset) *offset = rs->offset;
    if (contrib) {
        HIP_TRY(g, hipMemcpyAsync(contrib, rs->d_contrib, sizeof(double), hipMemcpyDeviceToHost, g->stream));
        HIP_TRY(g, hipStreamSynchronize(g->stream));
    }
    return TSDRGPU_OK;
}

// the host half of dsp_resample_process: counts and phases (dsp.c:258-262,272,306)
static int64_t build_chunks(double *offset_io, uint32_t chunk, int nchunks, double up, double down, RsChunk *tab)
{
    const double r = up / down;
    const double rinv = down / up;
    double offset = *offset_io;
    long long in_off = 0, out_off = 0;
    for (int c = 0; c < nchunks; c++) {
        const uint32_t n_out = (uint32_t)(int)(((double)chunk - offset) * r);
        if (tab) {
            tab[c].in_off = in_off;
            tab[c].out_off = out_off;
            tab[c].size = chunk;
            tab[c].n_out = n_out;
            tab[c].o = -offset * r;
        }
        offset += n_out * rinv - chunk;
        in_off += chunk;
        out_off += n_out;
    }
    *offset_io = offset;
    return out_off;
}

extern "C" int64_t tsdrgpu_resample_count(tsdrgpu_resampler_t *rs, uint32_t chunk, int nchunks, double up, double down)
{
    if (!rs || nchunks < 0 || !(up > 0) || !(down > 0)) return -1;
    double off = rs->offset;
    return build_chunks(&off, chunk, nchunks, up, down, nullptr);
}

extern "C" int tsdrgpu_resampler_track_frames(tsdrgpu_resampler_t *rs, int64_t frame_pixels, int64_t phase)
{
    if (!rs) return TSDRGPU_EINVAL;
    tsdrgpu_t *g = rs->g;
    if (frame_pixels == 0) {
        rs->frame_pixels = 0;
        rs->last_complete = -1;
        return TSDRGPU_OK;
    }
    // a workgroup's 2048-pixel span must not touch more than two frames
    if (frame_pixels < 4096 || phase < 0 || phase >= frame_pixels)
        return tsdr_fail(g, TSDRGPU_EINVAL, "tsdrgpu_resampler_track_frames", "frame_pixels must be >= 4096 and 0 <= phase < frame_pixels");
    if (!rs->d_carry && hipMalloc(&rs->d_carry, 4 * sizeof(float)) != hipSuccess)
        return tsdr_fail(g, TSDRGPU_ENOMEM, "tsdrgpu_resampler_track_frames", "carry");
    const float init[4] = {INFINITY, -INFINITY, INFINITY, -INFINITY};
    HIP_TRY(g, hipMemcpyAsync(rs->d_carry, init, sizeof(init), hipMemcpyHostToDevice, g->stream));
    HIP_TRY(g, hipStreamSynchronize(g->stream));  // `init` is on the stack
    rs->frame_pixels = frame_pixels;
    rs->phase = phase;
    rs->parity = 0;
    rs->last_complete = -1;
    return TSDRGPU_OK;
}

extern "C" int tsdrgpu_resampler_frame_minmax(tsdrgpu_resampler_t *rs, const float **d_min, const float **d_max, int *nframes)
{
    if (!rs) return TSDRGPU_EINVAL;
    if (rs->frame_pixels <= 0 || rs->last_complete < 0)
        return tsdr_fail(rs->g, TSDRGPU_ESTATE, "tsdrgpu_resampler_frame_minmax", "no tracked tsdrgpu_resample call yet");
    if (d_min) *d_min = rs->d_fmin;
    if (d_max) *d_max = rs->d_fmax;
    if (nframes) *nframes = rs->last_complete;
    return TSDRGPU_OK;
}

extern "C" int tsdrgpu_resample(tsdrgpu_resampler_t *rs, const float *d_in, int in_is_iq, uint32_t chunk, int nchunks,
                                double up, double down, int nearest, float *d_out, int64_t out_capacity,
                                int64_t *h_n_out)
{
    if (!rs || !d_in || !d_out || chunk == 0 || nchunks < 0 || nchunks > 65535 || !(up > 0) || !(down > 0))
        return rs ? tsdr_fail(rs->g, TSDRGPU_EINVAL, "tsdrgpu_resample", "bad argument") : TSDRGPU_EINVAL;
    tsdrgpu_t *g = rs->g;
    if (h_n_out) *h_n_out = 0;
    if (nchunks == 0) return TSDRGPU_OK;

    double probe = rs->offset;
    const int64_t total = build_chunks(&probe, chunk, nchunks, up, down, nullptr);
    if (total > out_capacity) return tsdr_fail(g, TSDRGPU_EINVAL, "tsdrgpu_resample", "output buffer too small");

    const bool track = rs->frame_pixels > 0;
    if (track && nearest) return tsdr_fail(g, TSDRGPU_EINVAL, "tsdrgpu_resample", "frame tracking needs the area mode");
    const long long P = rs->frame_pixels;
    const int ntouched = (track && total > 0) ? (int)((rs->phase + total + P - 1) / P) : 0;
    const int ncomplete = track ? (int)((rs->phase + total) / P) : 0;
    // one staging slot: chunk table [+ per-chunk frame positions + per-frame chunk ranges]
    const size_t tab_bytes = sizeof(RsChunk) * (size_t)nchunks;
    const size_t cf_off = (tab_bytes + 15) & ~(size_t)15;
    const size_t rg_off = cf_off + (track ? sizeof(RsChunkFrame) * (size_t)nchunks : 0);
    const size_t bytes = track ? rg_off + sizeof(RsFrameRange) * (size_t)(ntouched ? ntouched : 1) : tab_bytes;
    const int slot = staging_acquire(g, &rs->ring, bytes);
    if (slot < 0) return slot;
    RsChunk *tab = (RsChunk *)rs->ring.h[slot];
    double new_offset = rs->offset;  // committed only once nothing below can fail any more
    build_chunks(&new_offset, chunk, nchunks, up, down, tab);
    unsigned max_out = 0;
    for (int c = 0; c < nchunks; c++) max_out = tab[c].n_out > max_out ? tab[c].n_out : max_out;
    if (track) {
        RsChunkFrame *cf = (RsChunkFrame *)((char *)rs->ring.h[slot] + cf_off);
        for (int c = 0; c < nchunks; c++) {
            const long long pos = rs->phase + tab[c].out_off;
            cf[c].f = (int)(pos / P);
            cf[c].rem = pos % P;
            cf[c].pad = 0;
        }
        RsFrameRange *rg = (RsFrameRange *)((char *)rs->ring.h[slot] + rg_off);
        int c0 = 0;
        for (int j = 0; j < ntouched; j++) {
            const long long lo = (long long)j * P - rs->phase, hi = lo + P;  // call-relative pixel range of frame j
            while (c0 < nchunks && tab[c0].out_off + (long long)tab[c0].n_out <= lo) c0++;
            int c1 = c0;
            while (c1 < nchunks && tab[c1].out_off < hi) c1++;
            rg[j].c_lo = c0;
            rg[j].c_hi = c1;
        }
    }
    int rc = staging_push(g, &rs->ring, slot, bytes);
    if (rc) return rc;
    const RsChunk *d_tab = (const RsChunk *)rs->ring.d[slot];
    const RsChunkFrame *d_cf = (const RsChunkFrame *)((const char *)rs->ring.d[slot] + cf_off);
    const RsFrameRange *d_rg = (const RsFrameRange *)((const char *)rs->ring.d[slot] + rg_off);

    if (rs->cap_chunks < nchunks) {
        hipFree(rs->d_cin); hipFree(rs->d_tail); hipFree(rs->d_need);
        rs->d_cin = rs->d_tail = nullptr; rs->d_need = nullptr; rs->cap_chunks = 0;
        const int cap = nchunks + 64;
        if (hipMalloc(&rs->d_cin, sizeof(double) * cap) != hipSuccess || hipMalloc(&rs->d_tail, sizeof(double) * cap) != hipSuccess ||
            hipMalloc(&rs->d_need, cap) != hipSuccess)
            return tsdr_fail(g, TSDRGPU_ENOMEM, "tsdrgpu_resample", "chunk scratch");
        rs->cap_chunks = cap;
    }

    const double r = up / down;
    const unsigned bx = ceil_div_u(max_out ? max_out : 1, 256);
    dim3 grid(bx, (unsigned)nchunks);                                   // one pixel per thread (nearest)
    dim3 grid4(ceil_div_u((max_out ? max_out : 1) + 2 * RS_NPIX, 256 * RS_NPIX), (unsigned)nchunks);  // one pixel group per thread
    if (nearest) {
        // dsp.c:274-276: contrib is untouched in this mode
        if (max_out) {
            if (in_is_iq) TSDR_LAUNCH(g, PROF_RS_NEAREST, g->stream, (k_rs_nearest<true>), grid, 256, d_tab, d_in, d_out);
            else TSDR_LAUNCH(g, PROF_RS_NEAREST, g->stream, (k_rs_nearest<false>), grid, 256, d_tab, d_in, d_out);
            KERNEL_CHECK(g, "k_rs_nearest");
        }
    } else {
        const unsigned tb = ceil_div_u((unsigned)nchunks, 128);
        {
            if (in_is_iq) {
                TSDR_LAUNCH(g, PROF_RS_CARRY, g->stream, (k_rs_tail<true>), tb, 128, d_tab, nchunks, r, 1.0 / r, d_in, rs->d_tail, rs->d_need);
                TSDR_LAUNCH(g, PROF_RS_CARRY, g->stream, (k_rs_chain<true>), 1, 256, d_tab, nchunks, r, 1.0 / r, d_in, rs->d_tail, rs->d_need, rs->d_cin, rs->d_contrib);
            } else {
                TSDR_LAUNCH(g, PROF_RS_CARRY, g->stream, (k_rs_tail<false>), tb, 128, d_tab, nchunks, r, 1.0 / r, d_in, rs->d_tail, rs->d_need);
                TSDR_LAUNCH(g, PROF_RS_CARRY, g->stream, (k_rs_chain<false>), 1, 256, d_tab, nchunks, r, 1.0 / r, d_in, rs->d_tail, rs->d_need, rs->d_cin, rs->d_contrib);
            }
        }
        if (track) {
            const unsigned gx_up = ceil_div_u(chunk, (unsigned)(4 * RSU_LANES));  // k_rs_area_up's widest grid (one round per wave)
            const size_t nslots = (size_t)(grid4.x > gx_up ? grid4.x : gx_up) * (size_t)nchunks * 4;
            if (rs->cap_slots < nslots) {
                (void)hipStreamSynchronize(g->stream);
                hipFree(rs->d_slots);
                rs->d_slots = nullptr; rs->cap_slots = 0;
                if (hipMalloc(&rs->d_slots, sizeof(RsBlockMM) * (nslots + nslots / 8)) != hipSuccess)
                    return tsdr_fail(g, TSDRGPU_ENOMEM, "tsdrgpu_resample", "frame tracking records");
                rs->cap_slots = nslots + nslots / 8;
            }
            if (rs->cap_frames < ntouched) {
                (void)hipStreamSynchronize(g->stream);
                hipFree(rs->d_fmin); hipFree(rs->d_fmax);
                rs->d_fmin = rs->d_fmax = nullptr; rs->cap_frames = 0;
                if (hipMalloc(&rs->d_fmin, sizeof(float) * (ntouched + 16)) != hipSuccess ||
                    hipMalloc(&rs->d_fmax, sizeof(float) * (ntouched + 16)) != hipSuccess)
                    return tsdr_fail(g, TSDRGPU_ENOMEM, "tsdrgpu_resample", "frame min/max");
                rs->cap_frames = ntouched + 16;
            }
        }
        // sample-parallel kernel whenever the ratio upsamples moderately (the reference's geometry: r ~ 2); `rounds`
        // keeps a workgroup's pixels inside its LDS tile
        int mm_gx = (int)grid4.x;  // workgroups per chunk of whichever kernel writes the tracking records
        static const int force_old = getenv("TSDRGPU_RS_GROUPS") ? 1 : 0;
        const bool up_kernel = !force_old && r >= 1.0 && r <= 8.0;
        if (max_out && up_kernel) {
            int rounds = (int)((RSU_TILE - 4) / (4.0 * RSU_LANES * r));
            if (rounds < 1) rounds = 1;
            const int maxc = (int)r + 2;
            dim3 gridu(ceil_div_u(chunk, (unsigned)(4 * RSU_LANES * rounds)), (unsigned)nchunks);
            mm_gx = (int)gridu.x;
            if (track) {
                if (in_is_iq) TSDR_LAUNCH(g, PROF_RS_AREA, g->stream, (k_rs_area_up<true, true>), gridu, 256, d_tab, r, 1.0 / r, d_in, rs->d_cin, d_out, rounds, maxc, d_cf, P, rs->d_slots, RsBandGeom{0, 0, 0});
                else TSDR_LAUNCH(g, PROF_RS_AREA, g->stream, (k_rs_area_up<false, true>), gridu, 256, d_tab, r, 1.0 / r, d_in, rs->d_cin, d_out, rounds, maxc, d_cf, P, rs->d_slots, RsBandGeom{0, 0, 0});
            } else {
                if (in_is_iq) TSDR_LAUNCH(g, PROF_RS_AREA, g->stream, (k_rs_area_up<true, false>), gridu, 256, d_tab, r, 1.0 / r, d_in, rs->d_cin, d_out, rounds, maxc, (const RsChunkFrame *)nullptr, 0LL, (RsBlockMM *)nullptr, RsBandGeom{0, 0, 0});
                else TSDR_LAUNCH(g, PROF_RS_AREA, g->stream, (k_rs_area_up<false, false>), gridu, 256, d_tab, r, 1.0 / r, d_in, rs->d_cin, d_out, rounds, maxc, (const RsChunkFrame *)nullptr, 0LL, (RsBlockMM *)nullptr, RsBandGeom{0, 0, 0});
            }
        } else if (max_out) {
            if (track) {
                if (in_is_iq) TSDR_LAUNCH(g, PROF_RS_AREA, g->stream, (k_rs_area<true, true>), grid4, 256, d_tab, r, 1.0 / r, d_in, rs->d_cin, d_out, d_cf, P, rs->d_slots);
                else TSDR_LAUNCH(g, PROF_RS_AREA, g->stream, (k_rs_area<false, true>), grid4, 256, d_tab, r, 1.0 / r, d_in, rs->d_cin, d_out, d_cf, P, rs->d_slots);
            } else {
                if (in_is_iq) TSDR_LAUNCH(g, PROF_RS_AREA, g->stream, (k_rs_area<true, false>), grid4, 256, d_tab, r, 1.0 / r, d_in, rs->d_cin, d_out, (const RsChunkFrame *)nullptr, 0LL, (RsBlockMM *)nullptr);
                else TSDR_LAUNCH(g, PROF_RS_AREA, g->stream, (k_rs_area<false, false>), grid4, 256, d_tab, r, 1.0 / r, d_in, rs->d_cin, d_out, (const RsChunkFrame *)nullptr, 0LL, (RsBlockMM *)nullptr);
            }
        }
        KERNEL_CHECK(g, "k_rs_area");
        if (track && ntouched > 0) {
            TSDR_LAUNCH(g, PROF_RS_CARRY, g->stream, k_rs_minmax, (unsigned)ntouched, RSMM_T, rs->d_slots, mm_gx, d_rg, ntouched, ncomplete,
                        rs->d_carry + 2 * rs->parity, rs->d_carry + 2 * (1 - rs->parity), rs->d_fmin, rs->d_fmax);
            KERNEL_CHECK(g, "k_rs_minmax");
            rs->parity = 1 - rs->parity;
        }
    }
    if (track) {
        rs->phase = (rs->phase + total) % P;
        rs->last_complete = ncomplete;
    }
    rc = staging_release(g, &rs->ring, slot);
    if (rc) return rc;
    rs->offset = new_offset;
    if (h_n_out) *h_n_out = total;
    return TSDRGPU_OK;
}

// The row-band form of tsdrgpu_resample (include/tsdrgpu.h).  Chunk table, incoming-contrib chain and the carried state
// are those of the full call; what differs is the pixels computed: per chunk at most two band entries.
extern "C" int tsdrgpu_resample_band(tsdrgpu_resampler_t *rs, const float *d_in, int in_is_iq, uint32_t chunk, int nchunks, double up,
                                     double down, int width, int height, int y0, int rows, int64_t phase, float *d_band,
                                     int64_t band_capacity_frames, int64_t *h_n_out, int *h_frames_touched)
{
    if (!rs || !d_in || !d_band || chunk == 0 || nchunks < 0 || nchunks > 65535 || !(up > 0) || !(down > 0) || width <= 0 || height <= 0 ||
        y0 < 0 || rows <= 0 || y0 + rows > height || phase < 0 || phase >= (int64_t)width * height)
        return rs ? tsdr_fail(rs->g, TSDRGPU_EINVAL, "tsdrgpu_resample_band", "bad argument") : TSDRGPU_EINVAL;
    tsdrgpu_t *g = rs->g;
    // Frame tracking (tsdrgpu_resampler_track_frames) in the band form: every frame's min/max over THIS BAND's pixels — what the
    // fused band run (tsdrgpu_postproc_band_begin_minmax) exchanges before its one trip over the band.
    const bool track = rs->frame_pixels > 0;
    if (track && (rs->frame_pixels != (long long)width * height || rs->phase != phase))
        return tsdr_fail(g, TSDRGPU_EINVAL, "tsdrgpu_resample_band", "frame tracking is on with another frame size or phase");
    if (h_n_out) *h_n_out = 0;
    if (h_frames_touched) *h_frames_touched = 0;
    if (nchunks == 0) return TSDRGPU_OK;
    const long long P = (long long)width * height;
    double probe = rs->offset;
    const int64_t total = build_chunks(&probe, chunk, nchunks, up, down, nullptr);
    const long long touched = total > 0 ? (phase + total + P - 1) / P : 0;
    if (touched > band_capacity_frames) return tsdr_fail(g, TSDRGPU_EINVAL, "tsdrgpu_resample_band", "band buffer too small");
    const int ntouched = (int)touched, ncomplete = (int)((phase + total) / P);
    const size_t tab_bytes = sizeof(RsChunk) * (size_t)nchunks;
    const size_t ent_off = (tab_bytes + 15) & ~(size_t)15;
    // a chunk's output touches n_out / P + 2 frames at most (usually 1 or 2: the library polls 0.1 frame at a time, but
    // nothing forbids small frames with large chunks), each at most one band entry
    const size_t max_ent = (size_t)(2 * (long long)nchunks + total / P + 2);
    const size_t rg_off = (ent_off + sizeof(RsBandEntry) * max_ent + 15) & ~(size_t)15;  // tracked: the per-frame chunk ranges behind the entries
    const size_t bytes = rg_off + (track ? sizeof(RsFrameRange) * (size_t)(ntouched ? ntouched : 1) : 0);
    const int slot = staging_acquire(g, &rs->ring, bytes);
    if (slot < 0) return slot;
    RsChunk *tab = (RsChunk *)rs->ring.h[slot];
    RsBandEntry *ent = (RsBandEntry *)((char *)rs->ring.h[slot] + ent_off);
    double new_offset = rs->offset;
    build_chunks(&new_offset, chunk, nchunks, up, down, tab);
    const long long b_lo = (long long)y0 * width, b_hi = (long long)(y0 + rows) * width, Pb = (long long)rows * width;
    int nent = 0;
    int max_span = 0;
    // The sample-parallel kernel in its band form whenever it applies (1 <= r <= 8, frames of >= 4096 pixels so that a
    // workgroup's span touches at most two frames): 0.56 instead of 0.82 ms for a whole 2962 x 2250 frame stream, and
    // workgroups outside the band do no work.  TSDRGPU_RS_GROUPS keeps the pixel-group kernel (A/B).
    static const int force_groups = getenv("TSDRGPU_RS_GROUPS") ? 1 : 0;
    const bool up_kernel = !force_groups && up / down >= 1.0 && up / down <= 8.0 && P >= 4096;
    if (track && !up_kernel) {
        staging_release(g, &rs->ring, slot);
        return tsdr_fail(g, TSDRGPU_EINVAL, "tsdrgpu_resample_band", "frame tracking in the band form needs the sample-parallel kernel (1 <= up/down <= 8)");
    }
    if (up_kernel) {
        // (the chunk-frame table takes the place of the band entries in the staging slot, which was sized for more)
        RsChunkFrame *cf = (RsChunkFrame *)ent;
        for (int c = 0; c < nchunks; c++) {
            const long long pos = phase + tab[c].out_off;
            cf[c].f = (int)(pos / P);
            cf[c].rem = pos % P;
            cf[c].pad = 0;
        }
    }
    if (track) {  // as in tsdrgpu_resample: the chunks whose output overlaps frame j of the call
        RsFrameRange *rg = (RsFrameRange *)((char *)rs->ring.h[slot] + rg_off);
        int c0 = 0;
        for (int j = 0; j < ntouched; j++) {
            const long long lo = (long long)j * P - phase, hi = lo + P;
            while (c0 < nchunks && tab[c0].out_off + (long long)tab[c0].n_out <= lo) c0++;
            int c1 = c0;
            while (c1 < nchunks && tab[c1].out_off < hi) c1++;
            rg[j].c_lo = c0;
            rg[j].c_hi = c1;
        }
    }
    for (int c = 0; c < nchunks && !up_kernel; c++) {
        const long long g0 = phase + tab[c].out_off, g1 = g0 + tab[c].n_out;  // pixels counted from the first frame's start
        for (long long j = g0 / P; j * P < g1; j++) {
            long long lo = j * P + b_lo, hi = j * P + b_hi;
            if (lo < g0) lo = g0;
            if (hi > g1) hi = g1;
            if (lo >= hi) continue;
            RsBandEntry e;
            e.chunk = c;
            e.p_lo = (int)(lo - g0);
            e.p_hi = (int)(hi - g0);
            e.pad = 0;
            e.dst_off = j * Pb + (lo - j * P - b_lo);
            if ((size_t)nent >= max_ent || nent >= 65535) {
                staging_release(g, &rs->ring, slot);
                return tsdr_fail(g, TSDRGPU_EINVAL, "tsdrgpu_resample_band", "too many band entries in one call (<= 65535)");
            }
            ent[nent++] = e;
            if (e.p_hi - e.p_lo > max_span) max_span = e.p_hi - e.p_lo;
        }
    }
    int rc = staging_push(g, &rs->ring, slot, bytes);
    if (rc) return rc;
    const RsChunk *d_tab = (const RsChunk *)rs->ring.d[slot];
    const RsBandEntry *d_ent = (const RsBandEntry *)((const char *)rs->ring.d[slot] + ent_off);
    if (rs->cap_chunks < nchunks) {
        hipFree(rs->d_cin); hipFree(rs->d_tail); hipFree(rs->d_need);
        rs->d_cin = rs->d_tail = nullptr; rs->d_need = nullptr; rs->cap_chunks = 0;
        const int cap = nchunks + 64;
        if (hipMalloc(&rs->d_cin, sizeof(double) * cap) != hipSuccess || hipMalloc(&rs->d_tail, sizeof(double) * cap) != hipSuccess ||
            hipMalloc(&rs->d_need, cap) != hipSuccess)
            return tsdr_fail(g, TSDRGPU_ENOMEM, "tsdrgpu_resample_band", "chunk scratch");
        rs->cap_chunks = cap;
    }
    const double r = up / down;
    const unsigned tb = ceil_div_u((unsigned)nchunks, 128);
    // the incoming contrib of every chunk (data dependent, a few samples per chunk): every band needs it
    if (in_is_iq) {
        TSDR_LAUNCH(g, PROF_RS_CARRY, g->stream, (k_rs_tail<true>), tb, 128, d_tab, nchunks, r, 1.0 / r, d_in, rs->d_tail, rs->d_need);
        TSDR_LAUNCH(g, PROF_RS_CARRY, g->stream, (k_rs_chain<true>), 1, 256, d_tab, nchunks, r, 1.0 / r, d_in, rs->d_tail, rs->d_need, rs->d_cin, rs->d_contrib);
    } else {
        TSDR_LAUNCH(g, PROF_RS_CARRY, g->stream, (k_rs_tail<false>), tb, 128, d_tab, nchunks, r, 1.0 / r, d_in, rs->d_tail, rs->d_need);
        TSDR_LAUNCH(g, PROF_RS_CARRY, g->stream, (k_rs_chain<false>), 1, 256, d_tab, nchunks, r, 1.0 / r, d_in, rs->d_tail, rs->d_need, rs->d_cin, rs->d_contrib);
    }
    if (up_kernel) {
        int rounds = (int)((RSU_TILE - 4) / (4.0 * RSU_LANES * r));
        if (rounds < 1) rounds = 1;
        const int maxc = (int)r + 2;
        dim3 gridu(ceil_div_u(chunk, (unsigned)(4 * RSU_LANES * rounds)), (unsigned)nchunks);
        RsBandGeom bd;
        bd.b_lo = b_lo; bd.b_hi = b_hi; bd.Pb = Pb;
        const RsChunkFrame *d_cf = (const RsChunkFrame *)d_ent;
        if (track) {
            const size_t nslots = (size_t)gridu.x * (size_t)nchunks * 4;
            if (rs->cap_slots < nslots) {
                (void)hipStreamSynchronize(g->stream);
                hipFree(rs->d_slots);
                rs->d_slots = nullptr; rs->cap_slots = 0;
                if (hipMalloc(&rs->d_slots, sizeof(RsBlockMM) * (nslots + nslots / 8)) != hipSuccess)
                    return tsdr_fail(g, TSDRGPU_ENOMEM, "tsdrgpu_resample_band", "frame tracking records");
                rs->cap_slots = nslots + nslots / 8;
            }
            if (rs->cap_frames < ntouched) {
                (void)hipStreamSynchronize(g->stream);
                hipFree(rs->d_fmin); hipFree(rs->d_fmax);
                rs->d_fmin = rs->d_fmax = nullptr; rs->cap_frames = 0;
                if (hipMalloc(&rs->d_fmin, sizeof(float) * (ntouched + 16)) != hipSuccess ||
                    hipMalloc(&rs->d_fmax, sizeof(float) * (ntouched + 16)) != hipSuccess)
                    return tsdr_fail(g, TSDRGPU_ENOMEM, "tsdrgpu_resample_band", "frame min/max");
                rs->cap_frames = ntouched + 16;
            }
            if (in_is_iq) TSDR_LAUNCH(g, PROF_RS_AREA, g->stream, (k_rs_area_up<true, true, true>), gridu, 256, d_tab, r, 1.0 / r, d_in, rs->d_cin, d_band, rounds, maxc, d_cf, P, rs->d_slots, bd);
            else TSDR_LAUNCH(g, PROF_RS_AREA, g->stream, (k_rs_area_up<false, true, true>), gridu, 256, d_tab, r, 1.0 / r, d_in, rs->d_cin, d_band, rounds, maxc, d_cf, P, rs->d_slots, bd);
            if (ntouched > 0) {
                const RsFrameRange *d_rg = (const RsFrameRange *)((const char *)rs->ring.d[slot] + rg_off);
                TSDR_LAUNCH(g, PROF_RS_CARRY, g->stream, k_rs_minmax, (unsigned)ntouched, RSMM_T, rs->d_slots, (int)gridu.x, d_rg, ntouched, ncomplete,
                            rs->d_carry + 2 * rs->parity, rs->d_carry + 2 * (1 - rs->parity), rs->d_fmin, rs->d_fmax);
                rs->parity = 1 - rs->parity;
            }
        } else if (in_is_iq) TSDR_LAUNCH(g, PROF_RS_AREA, g->stream, (k_rs_area_up<true, false, true>), gridu, 256, d_tab, r, 1.0 / r, d_in, rs->d_cin, d_band, rounds, maxc, d_cf, P, (RsBlockMM *)nullptr, bd);
        else TSDR_LAUNCH(g, PROF_RS_AREA, g->stream, (k_rs_area_up<false, false, true>), gridu, 256, d_tab, r, 1.0 / r, d_in, rs->d_cin, d_band, rounds, maxc, d_cf, P, (RsBlockMM *)nullptr, bd);
    } else if (nent) {
        dim3 grid(ceil_div_u((unsigned)max_span + 2 * RS_NPIX, 256 * RS_NPIX), (unsigned)nent);
        if (in_is_iq) TSDR_LAUNCH(g, PROF_RS_AREA, g->stream, (k_rs_area_band<true>), grid, 256, d_tab, d_ent, r, 1.0 / r, d_in, rs->d_cin, d_band);
        else TSDR_LAUNCH(g, PROF_RS_AREA, g->stream, (k_rs_area_band<false>), grid, 256, d_tab, d_ent, r, 1.0 / r, d_in, rs->d_cin, d_band);
    }
    KERNEL_CHECK(g, "k_rs_area_band");
    rc = staging_release(g, &rs->ring, slot);
    if (rc) return rc;
    rs->offset = new_offset;
    if (track) {
        rs->phase = (rs->phase + total) % P;
        rs->last_complete = ncomplete;
    }
    if (h_n_out) *h_n_out = total;
    if (h_frames_touched) *h_frames_touched = (int)touched;
    return TSDRGPU_OK;
}
