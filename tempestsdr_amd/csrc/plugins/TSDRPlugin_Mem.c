/* TSDRPlugin_Mem.c — an in-memory replay source implementing the tsdrplugin_* ABI
 * (include/TSDRPlugin.h; the reference's counterpart is TSDRPlugin_RawFile, which re-reads its file
 * through fread() for every block).  The whole float32 IQ recording is loaded once into page-aligned
 * memory and handed to the library block by block without a copy, free-running, so that what a run
 * measures is the library (DMA straight out of these pages, kernels, DMA back) and not the source.
 *
 * params: "<file> <samplerate> [values_per_block = 524288] [loops = 0: forever] [sleep_us = 0] [type = float]"
 * type: float | int8 | uint8 | int16 | uint16, the RawFile plugin's formats.  Besides the ten mandatory entry points
 * the plugin exports tsdrplugin_readasync_raw (include/TSDRLibraryExt.h): a library that knows it receives the
 * blocks in their native format (and decodes them on the device); through the plain tsdrplugin_readasync the
 * values are converted to float on the host exactly like TSDRPlugin_RawFile.c:241-261 does.
 * After `loops` passes over the recording it idles like a live source until tsdrplugin_stop. */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "TSDRCodes.h"
#include "TSDRPlugin.h"
#include "TSDRLibraryExt.h"

static char g_file[1024];
static uint32_t g_rate;
static long g_block = 524288, g_loops = 0, g_sleep = 0;
static volatile int g_working;
static void *g_data;
static float *g_conv;   /* one block converted to float for the plain callback (narrow formats only) */
static size_t g_floats; /* values (two per IQ sample) */
static int g_type = TSDRX_SAMPLE_FLOAT32;
static size_t g_elem = 4;
static char g_err[256];
static int g_errcode;

void tsdrplugin_getName(char *name) { strcpy(name, "TSDR in-memory IQ replay"); }

static void unload(void)
{
    free(g_data);
    free(g_conv);
    g_data = NULL;
    g_conv = NULL;
    g_floats = 0;
}

int tsdrplugin_init(const char *params)
{
    unload();
    g_block = 524288; g_loops = 0; g_sleep = 0;
    char type[32] = "float";
    const int n = sscanf(params, "%1023s %u %ld %ld %ld %31s", g_file, &g_rate, &g_block, &g_loops, &g_sleep, type);
    if (!strcmp(type, "float")) { g_type = TSDRX_SAMPLE_FLOAT32; g_elem = 4; }
    else if (!strcmp(type, "int8")) { g_type = TSDRX_SAMPLE_INT8; g_elem = 1; }
    else if (!strcmp(type, "uint8")) { g_type = TSDRX_SAMPLE_UINT8; g_elem = 1; }
    else if (!strcmp(type, "int16")) { g_type = TSDRX_SAMPLE_INT16; g_elem = 2; }
    else if (!strcmp(type, "uint16")) { g_type = TSDRX_SAMPLE_UINT16; g_elem = 2; }
    else g_type = -1;
    if (n < 2 || g_rate == 0 || g_block <= 0 || (g_block & 1) || g_type < 0) {
        snprintf(g_err, sizeof(g_err), "usage: file samplerate [values_per_block] [loops] [sleep_us] [float|int8|uint8|int16|uint16]");
        return g_errcode = TSDR_PLUGIN_PARAMETERS_WRONG;
    }
    FILE *f = fopen(g_file, "rb");
    if (!f) {
        snprintf(g_err, sizeof(g_err), "cannot open %.200s", g_file);
        return g_errcode = TSDR_PLUGIN_PARAMETERS_WRONG;
    }
    fseek(f, 0, SEEK_END);
    const long bytes = ftell(f);
    fseek(f, 0, SEEK_SET);
    const size_t blocks = (size_t)bytes / g_elem / (size_t)g_block;
    if (blocks == 0) {
        fclose(f);
        snprintf(g_err, sizeof(g_err), "%.160s holds less than one block of %ld floats", g_file, g_block);
        return g_errcode = TSDR_PLUGIN_PARAMETERS_WRONG;
    }
    g_floats = blocks * (size_t)g_block;
    /* whole 2 MiB pages, so that every block the library page-locks for DMA has pages of its own */
    if (posix_memalign(&g_data, 2u << 20, ((g_floats * g_elem + (2u << 20) - 1) >> 21) << 21) != 0) g_data = NULL;
    if (!g_data || fread(g_data, g_elem, g_floats, f) != g_floats) {
        fclose(f);
        unload();
        snprintf(g_err, sizeof(g_err), "cannot load %.200s into memory", g_file);
        return g_errcode = TSDR_PLUGIN_PARAMETERS_WRONG;
    }
    fclose(f);
    /* everything the callbacks will ever point at exists from here until tsdrplugin_cleanup (tsdrplugin_memory_stable) */
    if (g_type != TSDRX_SAMPLE_FLOAT32 && !(g_conv = (float *)malloc(sizeof(float) * (size_t)g_block))) {
        unload();
        snprintf(g_err, sizeof(g_err), "out of memory");
        return g_errcode = TSDR_ERR_PLUGIN;
    }
    return g_errcode = TSDR_OK;
}

/* include/TSDRLibraryExt.h: the recording (and the conversion block) stay where they are from tsdrplugin_init to
 * tsdrplugin_cleanup, so the library may page-lock them and DMA straight out of them */
/* ... and the recording's contents never change: blocks handed over in their native format always point into it, blocks of
 * the plain callback do when the recording is float32 (narrow formats are converted into one reused block) */
TSDRPLUGIN_API int tsdrplugin_memory_stable(void)
{
    return TSDRX_MEMORY_MAPPED | TSDRX_MEMORY_IMMUTABLE_RAW | (g_type == TSDRX_SAMPLE_FLOAT32 ? TSDRX_MEMORY_IMMUTABLE : 0);
}

uint32_t tsdrplugin_setsamplerate(uint32_t rate) { (void)rate; return g_rate; }
uint32_t tsdrplugin_getsamplerate(void) { return g_rate; }
int tsdrplugin_setbasefreq(uint32_t freq) { (void)freq; return TSDR_OK; }
int tsdrplugin_setgain(float gain) { (void)gain; return TSDR_OK; }
char *tsdrplugin_getlasterrortext(void) { return g_errcode == TSDR_OK ? NULL : g_err; }
int tsdrplugin_stop(void) { g_working = 0; return TSDR_OK; }
void tsdrplugin_cleanup(void) { unload(); }

/* the blocks in their native format (TSDRLibraryExt.h); zero copy */
TSDRPLUGIN_API int tsdrplugin_readasync_raw(tsdrplugin_readasync_raw_function cb, void *ctx)
{
    if (!g_data) {
        snprintf(g_err, sizeof(g_err), "no recording loaded");
        return g_errcode = TSDR_PLUGIN_PARAMETERS_WRONG;
    }
    g_working = 1;
    const size_t blocks = g_floats / (size_t)g_block;
    for (long pass = 0; g_working && (g_loops == 0 || pass < g_loops); pass++)
        for (size_t b = 0; b < blocks && g_working; b++) {
            cb((const char *)g_data + b * (size_t)g_block * g_elem, (uint64_t)g_block, g_type, ctx, 0);
            if (g_sleep > 0) usleep((useconds_t)g_sleep);
        }
    while (g_working) usleep(2000); /* idle like a live source until stopped */
    return g_errcode = TSDR_OK;
}

int tsdrplugin_readasync(tsdrplugin_readasync_function cb, void *ctx)
{
    if (!g_data) {
        snprintf(g_err, sizeof(g_err), "no recording loaded");
        return g_errcode = TSDR_PLUGIN_PARAMETERS_WRONG;
    }
    g_working = 1;
    const size_t blocks = g_floats / (size_t)g_block;
    float *conv = g_conv; /* narrow formats: converted per block like TSDRPlugin_RawFile.c:241-261 */
    for (long pass = 0; g_working && (g_loops == 0 || pass < g_loops); pass++)
        for (size_t b = 0; b < blocks && g_working; b++) {
            const size_t at = b * (size_t)g_block;
            float *out = (float *)g_data + at;
            if (conv) {
                out = conv;
                switch (g_type) {
                    case TSDRX_SAMPLE_INT8: for (long i = 0; i < g_block; i++) conv[i] = ((const int8_t *)g_data)[at + i] / 128.0; break;
                    case TSDRX_SAMPLE_UINT8: for (long i = 0; i < g_block; i++) conv[i] = (((const uint8_t *)g_data)[at + i] - 128) / 128.0; break;
                    case TSDRX_SAMPLE_INT16: for (long i = 0; i < g_block; i++) conv[i] = ((const int16_t *)g_data)[at + i] / 32767.0; break;
                    default: for (long i = 0; i < g_block; i++) conv[i] = (((const uint16_t *)g_data)[at + i] - 32767) / 32767.0; break;
                }
            }
            cb(out, (uint64_t)g_block, ctx, 0);
            if (g_sleep > 0) usleep((useconds_t)g_sleep);
        }
    while (g_working) usleep(2000); /* idle like a live source until stopped */
    return g_errcode = TSDR_OK;
}
