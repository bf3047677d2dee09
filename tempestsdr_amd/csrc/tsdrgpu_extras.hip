// tsdrgpu_extras.hip — the components right next to the hot path:
//   f1  sample-format decode of the RawFile plugin on the device
//       (TSDRPlugin_RawFile/src/TSDRPlugin_RawFile.c:241-261): int8/uint8/int16/uint16 raw IQ cross
//       PCIe at 1/4..1/2 of the float32 volume
//   f2  native mode detection: argmax of the two plots -> frame rate / line count, the
//       "seen 3 times" acceptance rule and the video-mode table lookup that live in the Java GUI
//       (PlotVisualizer.java:200-247, Main.java:82,1233-1277,1301-1303,1346-1350, VideoMode.java:25-190)
//   f3  frame -> packed RGB with the debug colours and inversion of the JNI shim
//       (JavaGUI/jni/TSDRLibraryNDK.c:222-276), so that 4 bytes/pixel of final image leave the GPU
//   f4  plot decimation for display: per pixel column the maximum of the lags drawn in it
//       (PlotVisualizer.java:200-247, gui/scale/ZoomableXScale.java:133-149), so that `nwidth`
//       doubles instead of 0.67 M per plot go to the host
#include "tsdrgpu_internal.h"
#include <math.h>

// ---------------------------------------------------------------------------
// f1  decode.  The plugin divides in double and stores a float; the same here.
// ---------------------------------------------------------------------------
template <int TYPE>
__global__ __launch_bounds__(256) void k_decode(const void *__restrict__ raw, float *__restrict__ out, long long n)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        double v;
        if (TYPE == 1) v = (double)((const signed char *)raw)[i] / 128.0;
        else if (TYPE == 2) v = (double)((const short *)raw)[i] / 32767.0;
        else if (TYPE == 3) v = (double)((int)((const unsigned char *)raw)[i] - 128) / 128.0;
        else v = (double)((int)((const unsigned short *)raw)[i] - 32767) / 32767.0;
        out[i] = (float)v;
    }
}

extern "C" int tsdrgpu_decode_samples(tsdrgpu_t *g, const void *d_raw, int type, float *d_out, int64_t n)
{
    if (!g || !d_raw || !d_out || n < 0 || type < 0 || type > 4) return g ? tsdr_fail(g, TSDRGPU_EINVAL, "tsdrgpu_decode_samples", "bad argument") : TSDRGPU_EINVAL;
    if (n == 0) return TSDRGPU_OK;
    if (type == 0) {
        HIP_TRY(g, hipMemcpyAsync(d_out, d_raw, sizeof(float) * (size_t)n, hipMemcpyDeviceToDevice, g->stream));
        return TSDRGPU_OK;
    }
    long long blocks = (n + 255) / 256;
    const long long cap = (long long)g->prop.multiProcessorCount * 16;
    if (blocks > cap) blocks = cap;
    switch (type) {
        case 1: TSDR_LAUNCH(g, PROF_EXTRAS, g->stream, (k_decode<1>), (unsigned)blocks, 256, d_raw, d_out, n); break;
        case 2: TSDR_LAUNCH(g, PROF_EXTRAS, g->stream, (k_decode<2>), (unsigned)blocks, 256, d_raw, d_out, n); break;
        case 3: TSDR_LAUNCH(g, PROF_EXTRAS, g->stream, (k_decode<3>), (unsigned)blocks, 256, d_raw, d_out, n); break;
        default: TSDR_LAUNCH(g, PROF_EXTRAS, g->stream, (k_decode<4>), (unsigned)blocks, 256, d_raw, d_out, n); break;
    }
    KERNEL_CHECK(g, "k_decode");
    return TSDRGPU_OK;
}

// ---------------------------------------------------------------------------
// f3  frame -> 0x00RRGGBB
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_frame_to_rgb(const float *__restrict__ frame, int *__restrict__ rgb, long long n, int inverted)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float val = frame[i];
        if (val > 0.0f && val <= 1.0f) {
            const int col = inverted ? (255 - (int)(val * 255.0f)) : ((int)(val * 255.0f));
            rgb[i] = col | (col << 8) | (col << 16);
        } else if (val <= 0.0f) {
            rgb[i] = inverted ? 0xFFFFFF : 0;
        } else if (val == 256.0f) {   // PIXEL_SPECIAL_VALUE_R
            rgb[i] = 255 << 16;
        } else if (val == 512.0f) {   // PIXEL_SPECIAL_VALUE_G
            rgb[i] = 255 << 8;
        } else if (val == 1024.0f) {  // PIXEL_SPECIAL_VALUE_B
            rgb[i] = 255;
        } else if (val == 2048.0f) {  // PIXEL_SPECIAL_VALUE_TRANSPARENT: the previous pixel stays
        } else {
            rgb[i] = inverted ? 0 : 0xFFFFFF;
        }
    }
}

// ---------------------------------------------------------------------------
// dsp_autogain_t.snr (dsp.c:69-93): mean / stdev of a frame as dsp_autogain_run leaves it — the sum skips sentinel
// pixels but divides by all of them, the deviations run over every pixel.  No caller of the reference reads it (its
// report is commented out at dsp.c:234), so it is not part of the post-processing launch set: this is the on-demand
// form.  f64 tree sums in a fixed order (deterministic); the reference adds sequentially, so the result agrees to
// ~1e-12 relative, not bit for bit (tests: 1e-9).
// ---------------------------------------------------------------------------
#define SNR_BLOCKS 256
__device__ __forceinline__ double snr_block_sum(double v, double *sh)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    const double r = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return r;  // the same value in every thread
}
// blockIdx.y = frame; part: [F][3][SNR_BLOCKS] (sums, squared deviations, deviations)
__global__ __launch_bounds__(256) void k_snr_sum(const float *__restrict__ frames, long long stride, long long n, double *__restrict__ parts)
{
    __shared__ double sh[4];
    const float *x = frames + (long long)blockIdx.y * stride;
    double s = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)SNR_BLOCKS * 256) {
        const float v = x[i];
        if (!(v > 250.0 || v < -250)) s += (double)v;
    }
    s = snr_block_sum(s, sh);
    if (threadIdx.x == 0) parts[(size_t)blockIdx.y * 3 * SNR_BLOCKS + blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_snr_dev(const float *__restrict__ frames, long long stride, long long n, double *__restrict__ parts)
{
    __shared__ double sh[4];
    const float *x = frames + (long long)blockIdx.y * stride;
    double *part = parts + (size_t)blockIdx.y * 3 * SNR_BLOCKS;
    const double sum = snr_block_sum(part[threadIdx.x], sh);  // SNR_BLOCKS == blockDim.x
    const double mean = sum / (double)n;
    double s2 = 0.0, s1 = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)SNR_BLOCKS * 256) {
        const double d = x[i] - mean;
        s2 += d * d;
        s1 += d;
    }
    s2 = snr_block_sum(s2, sh);
    s1 = snr_block_sum(s1, sh);
    if (threadIdx.x == 0) { part[SNR_BLOCKS + blockIdx.x] = s2; part[2 * SNR_BLOCKS + blockIdx.x] = s1; }
}
__global__ __launch_bounds__(256) void k_snr_final(long long n, const double *__restrict__ parts, float *__restrict__ out)
{
    __shared__ double sh[4];
    const double *part = parts + (size_t)blockIdx.x * 3 * SNR_BLOCKS;
    const double sum = snr_block_sum(part[threadIdx.x], sh);
    const double s2 = snr_block_sum(part[SNR_BLOCKS + threadIdx.x], sh);
    const double s1 = snr_block_sum(part[2 * SNR_BLOCKS + threadIdx.x], sh);
    if (threadIdx.x == 0) {
        const double mean = sum / (double)n;
        const double stdev = sqrt((s2 - s1 * s1 / (double)n) / (double)(n - 1));
        out[blockIdx.x] = (float)(mean / stdev);
    }
}
// F frames at once on `st` (the post-processing run's by-product, tsdrgpu_postproc_set_snr): d_parts holds
// tsdr_snr_part_doubles() doubles per frame
size_t tsdr_snr_part_doubles(void) { return 3 * SNR_BLOCKS; }
int tsdr_snr_batch(tsdrgpu_t *g, hipStream_t st, const float *d_frames, long long stride, long long npixels, int F, double *d_parts, float *d_snr)
{
    TSDR_LAUNCH(g, PROF_EXTRAS, st, k_snr_sum, dim3(SNR_BLOCKS, F), 256, d_frames, stride, npixels, d_parts);
    TSDR_LAUNCH(g, PROF_EXTRAS, st, k_snr_dev, dim3(SNR_BLOCKS, F), 256, d_frames, stride, npixels, d_parts);
    TSDR_LAUNCH(g, PROF_EXTRAS, st, k_snr_final, F, 256, npixels, (const double *)d_parts, d_snr);
    if (hipGetLastError() != hipSuccess) return tsdr_fail(g, TSDRGPU_EHIP, "dsp_autogain_t.snr", "launch");
    return TSDRGPU_OK;
}
extern "C" int tsdrgpu_frame_snr(tsdrgpu_t *g, const float *d_frame, int64_t npixels, float *h_snr)
{
    if (!g || !d_frame || npixels < 1 || !h_snr) return g ? tsdr_fail(g, TSDRGPU_EINVAL, "tsdrgpu_frame_snr", "bad argument") : TSDRGPU_EINVAL;
    double *d_part = nullptr;
    if (hipMalloc(&d_part, sizeof(double) * (3 * SNR_BLOCKS + 1)) != hipSuccess) return tsdr_fail(g, TSDRGPU_ENOMEM, "tsdrgpu_frame_snr", "partials");
    float *d_out = reinterpret_cast<float *>(d_part + 3 * SNR_BLOCKS);
    int rc = tsdr_snr_batch(g, g->stream, d_frame, 0, (long long)npixels, 1, d_part, d_out);
    if (!rc && (hipMemcpyAsync(h_snr, d_out, sizeof(float), hipMemcpyDeviceToHost, g->stream) != hipSuccess ||
                hipStreamSynchronize(g->stream) != hipSuccess))
        rc = tsdr_fail(g, TSDRGPU_EHIP, "tsdrgpu_frame_snr", "copy");
    else if (rc) (void)hipStreamSynchronize(g->stream);
    (void)hipFree(d_part);
    return rc;
}

extern "C" int tsdrgpu_frame_to_rgb(tsdrgpu_t *g, const float *d_frame, int32_t *d_rgb, int64_t npixels, int inverted)
{
    if (!g || !d_frame || !d_rgb || npixels < 0) return g ? tsdr_fail(g, TSDRGPU_EINVAL, "tsdrgpu_frame_to_rgb", "bad argument") : TSDRGPU_EINVAL;
    if (npixels == 0) return TSDRGPU_OK;
    long long blocks = (npixels + 255) / 256;
    const long long cap = (long long)g->prop.multiProcessorCount * 16;
    if (blocks > cap) blocks = cap;
    TSDR_LAUNCH(g, PROF_EXTRAS, g->stream, k_frame_to_rgb, (unsigned)blocks, 256, d_frame, (int *)d_rgb, npixels, inverted);
    KERNEL_CHECK(g, "k_frame_to_rgb");
    return TSDRGPU_OK;
}

// ---------------------------------------------------------------------------
// f2  mode detection (host-side arithmetic on the two argmax indices)
// ---------------------------------------------------------------------------
struct VideoModeRow {
    const char *name;
    int width, height;  // TOTAL pixels per line / lines per frame, blanking included
    double refresh;
};

// the GUI's pre-registered modes, JavaGUI/src/martin/tempest/gui/VideoMode.java:25-106 (data)
static const VideoModeRow kModes[] = {
    {"PAL TV", 576, 625, 25}, {"640x400 @ 85Hz", 832, 445, 85}, {"720x400 @ 85Hz", 936, 446, 85},
    {"640x480 @ 60Hz", 800, 525, 60}, {"640x480 @ 100Hz", 848, 509, 100}, {"640x480 @ 72Hz", 832, 520, 72},
    {"640x480 @ 75Hz", 840, 500, 75}, {"640x480 @ 85Hz", 832, 509, 85}, {"768x576 @ 60 Hz", 976, 597, 60},
    {"768x576 @ 72 Hz", 992, 601, 72}, {"768x576 @ 75 Hz", 1008, 602, 75}, {"768x576 @ 85 Hz", 1008, 605, 85},
    {"768x576 @ 100 Hz", 1024, 611, 100}, {"800x600 @ 56Hz", 1024, 625, 56}, {"800x600 @ 60Hz", 1056, 628, 60},
    {"800x600 @ 72Hz", 1040, 666, 72}, {"800x600 @ 75Hz", 1056, 625, 75}, {"800x600 @ 85Hz", 1048, 631, 85},
    {"800x600 @ 100Hz", 1072, 636, 100}, {"1024x600 @ 60 Hz", 1312, 622, 60}, {"1024x768i @ 43Hz", 1264, 817, 43},
    {"1024x768 @ 60Hz", 1344, 806, 60}, {"1024x768 @ 70Hz", 1328, 806, 70}, {"1024x768 @ 75Hz", 1312, 800, 75},
    {"1024x768 @ 85Hz", 1376, 808, 85}, {"1024x768 @ 100Hz", 1392, 814, 100}, {"1024x768 @ 120Hz", 1408, 823, 120},
    {"1152x864 @ 60Hz", 1520, 895, 60}, {"1152x864 @ 75Hz", 1600, 900, 75}, {"1152x864 @ 85Hz", 1552, 907, 85},
    {"1152x864 @ 100Hz", 1568, 915, 100}, {"1280x768 @ 60 Hz", 1680, 795, 60}, {"1280x800 @ 60 Hz", 1680, 828, 60},
    {"1280x960 @ 60Hz", 1800, 1000, 60}, {"1280x960 @ 75Hz", 1728, 1002, 75}, {"1280x960 @ 85Hz", 1728, 1011, 85},
    {"1280x960 @ 100Hz", 1760, 1017, 100}, {"1280x1024 @ 60Hz", 1688, 1066, 60}, {"1280x1024 @ 75Hz", 1688, 1066, 75},
    {"1280x1024 @ 85Hz", 1728, 1072, 85}, {"1280x1024 @ 100Hz", 1760, 1085, 100}, {"1280x1024 @ 120Hz", 1776, 1097, 120},
    {"1368x768 @ 60 Hz", 1800, 795, 60}, {"1400x1050 @ 60Hz", 1880, 1082, 60}, {"1400x1050 @ 72 Hz", 1896, 1094, 72},
    {"1400x1050 @ 75 Hz", 1896, 1096, 75}, {"1400x1050 @ 85 Hz", 1912, 1103, 85}, {"1400x1050 @ 100 Hz", 1928, 1112, 100},
    {"1440x900 @ 60 Hz", 1904, 932, 60}, {"1440x1050 @ 60 Hz", 1936, 1087, 60}, {"1600x1000 @ 60Hz", 2144, 1035, 60},
    {"1600x1000 @ 75Hz", 2160, 1044, 75}, {"1600x1000 @ 85Hz", 2176, 1050, 85}, {"1600x1000 @ 100Hz", 2192, 1059, 100},
    {"1600x1024 @ 60Hz", 2144, 1060, 60}, {"1600x1024 @ 75Hz", 2176, 1069, 75}, {"1600x1024 @ 76Hz", 2096, 1070, 76},
    {"1600x1024 @ 85Hz", 2176, 1075, 85}, {"1600x1200 @ 60Hz", 2160, 1250, 60}, {"1600x1200 @ 65Hz", 2160, 1250, 65},
    {"1600x1200 @ 70Hz", 2160, 1250, 70}, {"1600x1200 @ 75Hz", 2160, 1250, 75}, {"1600x1200 @ 85Hz", 2160, 1250, 85},
    {"1600x1200 @ 100 Hz", 2208, 1271, 100}, {"1680x1050 @ 60Hz (reduced blanking)", 1840, 1080, 60},
    {"1680x1050 @ 60Hz (non-interlaced)", 2240, 1089, 60}, {"1680x1050 @ 60 Hz", 2256, 1087, 60},
    {"1792x1344 @ 60Hz", 2448, 1394, 60}, {"1792x1344 @ 75Hz", 2456, 1417, 75}, {"1856x1392 @ 60Hz", 2528, 1439, 60},
    {"1856x1392 @ 75Hz", 2560, 1500, 75}, {"1920x1080 @ 60Hz", 2576, 1125, 60}, {"1920x1080 @ 75Hz", 2608, 1126, 75},
    {"1920x1200 @ 60Hz", 2592, 1242, 60}, {"1920x1200 @ 75Hz", 2624, 1253, 75}, {"1920x1440 @ 60Hz", 2600, 1500, 60},
    {"1920x1440 @ 75Hz", 2640, 1500, 75}, {"1920x2400 @ 25Hz", 2048, 2434, 25}, {"1920x2400 @ 30Hz", 2044, 2434, 30},
    {"2048x1536 @ 60Hz", 2800, 1589, 60},
};
static const int kModeCount = (int)(sizeof(kModes) / sizeof(kModes[0]));

// VideoMode.findClosestVideoModeId(framerate, height, modes), VideoMode.java:163-190
static int closest_mode(double framerate, int height)
{
    int mode = -1;
    double diff = 5000.0;
    for (int i = 0; i < kModeCount; i++)
        if (kModes[i].height == height) {
            const double delta = fabs(kModes[i].refresh - framerate);
            if (delta < diff) { diff = delta; mode = i; }
        }
    if (mode == -1) {
        int idiff = 5000;
        for (int i = 0; i < kModeCount; i++) {
            const int delta = abs(kModes[i].height - height);
            if (delta < idiff) { idiff = delta; mode = i; }
        }
    }
    return mode;
}

#define DETECT_SLOTS 256
struct tsdrgpu_modedetect {
    long long key[DETECT_SLOTS];
    int count[DETECT_SLOTS];
    int used;
};

extern "C" int tsdrgpu_modedetect_create(tsdrgpu_modedetect_t **out)
{
    if (!out) return TSDRGPU_EINVAL;
    *out = (tsdrgpu_modedetect_t *)calloc(1, sizeof(tsdrgpu_modedetect_t));
    return *out ? TSDRGPU_OK : TSDRGPU_ENOMEM;
}
extern "C" void tsdrgpu_modedetect_destroy(tsdrgpu_modedetect_t *d) { free(d); }
extern "C" void tsdrgpu_modedetect_reset(tsdrgpu_modedetect_t *d) { if (d) d->used = 0; }

// One (frame plot, line plot) pair, as Main.onIncommingPlot handles it (Main.java:1233-1277).
extern "C" int tsdrgpu_modedetect_feed(tsdrgpu_modedetect_t *d, int frame_offset, int frame_idx, int line_offset, int line_idx,
                                       uint32_t samplerate, tsdrgpu_detection_t *out)
{
    if (!d || !out || frame_offset + frame_idx <= 0 || line_offset + line_idx <= 0) return TSDRGPU_EINVAL;
    memset(out, 0, sizeof(*out));
    const int frame_lag = frame_offset + frame_idx, line_lag = line_offset + line_idx;
    const double fps = samplerate / (double)frame_lag;                         // Main.java:1301-1303
    const double h = (double)frame_lag / (double)line_lag;                     // Main.java:1346-1350
    const int height = (int)floor(h + 0.5);                                    // Math.round
    const long long key = (long long)(fps * height);                           // hashHeightAndFPS, Main.java:1229-1231
    out->frame_lag = frame_lag;
    out->line_lag = line_lag;
    out->framerate = fps;
    out->linerate = samplerate / (double)line_lag;
    out->height = height;
    int slot = -1;
    for (int i = 0; i < d->used; i++)
        if (d->key[i] == key) { slot = i; break; }
    // accepted once the same (fps, height) has already been seen 3 times (Main.java:82,1257-1268)
    if (slot >= 0 && d->count[slot] == 3) {
        out->accepted = 1;
    } else {
        if (slot < 0) {
            slot = d->used < DETECT_SLOTS ? d->used++ : (DETECT_SLOTS - 1);
            d->key[slot] = key;
            d->count[slot] = 0;
        }
        d->count[slot]++;
    }
    out->seen = d->count[slot];
    const int m = closest_mode(fps, height);
    out->mode_id = m;
    if (m >= 0) {
        snprintf(out->mode_name, sizeof(out->mode_name), "%s", kModes[m].name);
        out->mode_width = kModes[m].width;
        out->mode_height = kModes[m].height;
        out->mode_refresh = kModes[m].refresh;
    }
    // what tsdr_setresolution(height, fps) would derive (TSDRLibrary.c:543-546)
    out->pixelrate = (double)((int)(2 * (samplerate / (fps * height)))) * height * fps;
    return TSDRGPU_OK;
}

// ---------------------------------------------------------------------------
// f4  plot decimation (PlotVisualizer.populateData).  value_to_pixel_absolute is monotone in the lag
// index, so the lags of pixel column c form one contiguous range, found by bisection; one wave per
// column takes its maximum.  max_index (first lag holding the largest value of the visible range) is a
// two-stage argmax.  The O(nwidth) tail of populateData — repeating the previous column where no lag
// maps to a column, lowest/highest — runs on the host from the column maxima.
// ---------------------------------------------------------------------------
struct PlotScale {
    double one_val_in_pixels, one_px_in_values, offset_val, min_value;
    int offset_px;
};
__host__ __device__ static inline int plot_px(const PlotScale &s, int id)
{
    return (int)(((double)id - s.min_value) * s.one_val_in_pixels) - s.offset_px;
}
// first id in [lo, hi) whose pixel is >= c
__device__ static int plot_lower_bound(const PlotScale &s, int lo, int hi, int c)
{
    while (lo < hi) {
        const int mid = lo + (hi - lo) / 2;
        if (plot_px(s, mid) >= c) hi = mid; else lo = mid + 1;
    }
    return lo;
}

#define PLOT_ARG_BLOCKS 256
__global__ __launch_bounds__(64) void k_plot_columns(const double *__restrict__ data, int first_id, int last_id, int nwidth,
                                                     PlotScale s, double *__restrict__ colmax, int *__restrict__ colcount)
{
    const int c = blockIdx.x, lane = threadIdx.x;
    const int a = plot_lower_bound(s, first_id, last_id, c), b = plot_lower_bound(s, a, last_id, c + 1);
    double m = -INFINITY;
    for (int id = a + lane; id < b; id += 64) m = fmax(m, data[id]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_down(m, o, 64));
    // populateData starts a column from its first lag and moves on a LARGER value only (PlotVisualizer.java:222-231): a NaN there
    // stays the column's value, a NaN anywhere else is never taken (fmax's semantics)
    if (lane == 0) {
        if (b > a && data[a] != data[a]) m = data[a];
        colmax[c] = m;
        colcount[c] = b - a;
    }
}

__global__ __launch_bounds__(256) void k_plot_argmax(const double *__restrict__ data, int first_id, int last_id,
                                                     double *__restrict__ pval, int *__restrict__ pidx, int *__restrict__ done,
                                                     int *__restrict__ out)
{
    double best = -INFINITY;
    int at = 0x7fffffff;
    for (int i = first_id + blockIdx.x * blockDim.x + threadIdx.x; i < last_id; i += gridDim.x * blockDim.x) {
        const double v = data[i];
        if (v > best) { best = v; at = i; }
    }
    __shared__ double sb[4];
    __shared__ int si[4];
    __shared__ int last_block;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double ob = __shfl_down(best, o, 64);
        const int oi = __shfl_down(at, o, 64);
        if (ob > best || (ob == best && oi < at)) { best = ob; at = oi; }
    }
    if ((threadIdx.x & 63) == 0) { sb[threadIdx.x >> 6] = best; si[threadIdx.x >> 6] = at; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++)
            if (sb[w] > best || (sb[w] == best && si[w] < at)) { best = sb[w]; at = si[w]; }
        pval[blockIdx.x] = best;
        pidx[blockIdx.x] = at;
        __threadfence();
        last_block = atomicAdd(done, 1) == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (!last_block) return;
    // the last workgroup to finish folds the partials (any order: the comparison is a total order)
    __threadfence();
    best = -INFINITY;
    at = 0x7fffffff;
    for (int b = threadIdx.x; b < (int)gridDim.x; b += blockDim.x) {
        const double ob = ((volatile double *)pval)[b];
        const int oi = ((volatile int *)pidx)[b];
        if (ob > best || (ob == best && oi < at)) { best = ob; at = oi; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double ob = __shfl_down(best, o, 64);
        const int oi = __shfl_down(at, o, 64);
        if (ob > best || (ob == best && oi < at)) { best = ob; at = oi; }
    }
    if ((threadIdx.x & 63) == 0) { sb[threadIdx.x >> 6] = best; si[threadIdx.x >> 6] = at; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++)
            if (sb[w] > best || (sb[w] == best && si[w] < at)) { best = sb[w]; at = si[w]; }
        out[0] = at;
        *done = 0;
    }
}

extern "C" void tsdrgpu_plotscale_default(int size, int nwidth, tsdrgpu_plotscale_t *s)
{
    // ZoomableXScale after reset(), setMinMaxValue(0, size), setMaxPixels(nwidth): zoom 1, max_zoom_val 10
    // (PlotVisualizer.java:64,259-264,296; ZoomableXScale.java:177-188)
    if (!s) return;
    const double span = (double)size;
    double scale = 1.0;
    s->one_val_in_pixels = nwidth / (span * scale);
    s->one_px_in_values = (span * scale) / nwidth;
    if (nwidth * s->one_px_in_values < 10.0) {
        scale = 10.0 / span;
        s->one_val_in_pixels = nwidth / (span * scale);
        s->one_px_in_values = (span * scale) / nwidth;
    }
    s->offset_val = 0.0;
    s->min_value = 0.0;
    s->offset_px = 0;
}

extern "C" int tsdrgpu_plot_columns(tsdrgpu_t *g, const double *d_data, int size, int nwidth, const tsdrgpu_plotscale_t *scale,
                                    double *h_visdata, double *h_lowest, double *h_highest, int *h_max_index)
{
    if (!g || !d_data || size <= 0 || nwidth <= 0 || nwidth > 65535 || !h_visdata)
        return g ? tsdr_fail(g, TSDRGPU_EINVAL, "tsdrgpu_plot_columns", "bad argument") : TSDRGPU_EINVAL;
    tsdrgpu_plotscale_t def;
    if (!scale) { tsdrgpu_plotscale_default(size, nwidth, &def); scale = &def; }
    PlotScale s = {scale->one_val_in_pixels, scale->one_px_in_values, scale->offset_val, scale->min_value, scale->offset_px};
    if (!(s.one_val_in_pixels > 0.0)) return tsdr_fail(g, TSDRGPU_EINVAL, "tsdrgpu_plot_columns", "scale must be positive");
    // visible range, PlotVisualizer.java:211-212
    double t = 0 * s.one_px_in_values + s.offset_val + s.min_value;
    t = t > 0 ? t : 0;
    const int first_id = (int)(t < size ? t : size);
    t = nwidth * s.one_px_in_values + s.offset_val + s.min_value + 1;
    t = t > 0 ? t : 0;
    const int last_id = (int)(t < size ? t : size);

    // scratch: [nwidth] column maxima + [PLOT_ARG_BLOCKS] partial maxima (doubles), then the ints
    const size_t nd = (size_t)nwidth + PLOT_ARG_BLOCKS + 2, ni = (size_t)nwidth + PLOT_ARG_BLOCKS + 2;
    void *d_scratch = nullptr;
    if (hipMalloc(&d_scratch, nd * sizeof(double) + ni * sizeof(int)) != hipSuccess)
        return tsdr_fail(g, TSDRGPU_ENOMEM, "tsdrgpu_plot_columns", "scratch");
    double *d_col = (double *)d_scratch, *d_pval = d_col + nwidth;
    int *d_cnt = (int *)(d_col + nd), *d_pidx = d_cnt + nwidth, *d_done = d_pidx + PLOT_ARG_BLOCKS, *d_out = d_done + 1;
    int rc = TSDRGPU_OK;
    double *hcol = (double *)malloc(sizeof(double) * (nwidth + 2));
    int *hcnt = (int *)malloc(sizeof(int) * (nwidth + 2));
    do {
        if (!hcol || !hcnt) { rc = tsdr_fail(g, TSDRGPU_ENOMEM, "tsdrgpu_plot_columns", "host scratch"); break; }
        if (hipMemsetAsync(d_done, 0, 2 * sizeof(int), g->stream) != hipSuccess) { rc = tsdr_fail(g, TSDRGPU_EHIP, "tsdrgpu_plot_columns", "memset"); break; }
        TSDR_LAUNCH(g, PROF_EXTRAS, g->stream, k_plot_columns, nwidth, 64, d_data, first_id, last_id, nwidth, s, d_col, d_cnt);
        if (last_id > first_id)
            TSDR_LAUNCH(g, PROF_EXTRAS, g->stream, k_plot_argmax, PLOT_ARG_BLOCKS, 256, d_data, first_id, last_id, d_pval, d_pidx, d_done, d_out);
        if (hipGetLastError() != hipSuccess) { rc = tsdr_fail(g, TSDRGPU_EHIP, "tsdrgpu_plot_columns", "launch"); break; }
        // data[0] (initial lowest/highest/max) and data[first_id] (initial localmax)
        const int fid = first_id < size ? first_id : size - 1;
        double edge[2];
        int hmax = 0;
        if (hipMemcpyAsync(hcol, d_col, sizeof(double) * nwidth, hipMemcpyDeviceToHost, g->stream) != hipSuccess ||
            hipMemcpyAsync(hcnt, d_cnt, sizeof(int) * nwidth, hipMemcpyDeviceToHost, g->stream) != hipSuccess ||
            hipMemcpyAsync(&edge[0], d_data, sizeof(double), hipMemcpyDeviceToHost, g->stream) != hipSuccess ||
            hipMemcpyAsync(&edge[1], d_data + fid, sizeof(double), hipMemcpyDeviceToHost, g->stream) != hipSuccess ||
            (last_id > first_id && hipMemcpyAsync(&hmax, d_out, sizeof(int), hipMemcpyDeviceToHost, g->stream) != hipSuccess) ||
            hipStreamSynchronize(g->stream) != hipSuccess) {
            rc = tsdr_fail(g, TSDRGPU_EHIP, "tsdrgpu_plot_columns", "copy back");
            break;
        }
        // the sequential tail of populateData over the non-empty columns
        double highest = edge[0], lowest = edge[0], localmax = edge[1];
        int prev_px = 0;
        for (int px = 0; px < nwidth; px++) {
            if (hcnt[px] <= 0) continue;
            if (prev_px != px) {
                if (localmax > highest) highest = localmax; else if (localmax < lowest) lowest = localmax;
                for (int i = prev_px; i < px; i++) h_visdata[i] = localmax;
                localmax = hcol[px];
                prev_px = px;
            } else if (hcol[px] > localmax) {
                localmax = hcol[px];  // column 0 joins the initial data[first_id]
            }
        }
        for (int i = prev_px; i < nwidth; i++) h_visdata[i] = localmax;
        // max over the visible range starts from data[0] (PlotVisualizer.java:203-205,226-229)
        int maxi = 0;
        if (last_id > first_id && hmax != 0x7fffffff) {  // (0x7fffffff: no lag of the range compares larger than -inf — all NaN: index 0 stays)
            double dv;
            if (hipMemcpy(&dv, d_data + hmax, sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) { rc = tsdr_fail(g, TSDRGPU_EHIP, "tsdrgpu_plot_columns", "copy back"); break; }
            if (dv > edge[0]) maxi = hmax;
        }
        if (h_lowest) *h_lowest = lowest;
        if (h_highest) *h_highest = highest;
        if (h_max_index) *h_max_index = maxi;
    } while (0);
    free(hcol);
    free(hcnt);
    (void)hipFree(d_scratch);
    return rc;
}
