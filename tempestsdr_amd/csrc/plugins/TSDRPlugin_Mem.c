/* TSDRPlugin_Mem.c — an in-memory replay source implementing the tsdrplugin_* ABI
 * (include/TSDRPlugin.h; the reference's counterpart is TSDRPlugin_RawFile, which re-reads its file
 * through fread() for every block).  The whole float32 IQ recording is loaded once into page-aligned
 * memory and handed to the library block by block without a copy, free-running, so that what a run
 * measures is the library (DMA straight out of these pages, kernels, DMA back) and not the source.
 *
 * params: "<file> <samplerate> [floats_per_block = 524288] [loops = 0: forever] [sleep_us = 0]"
 * After `loops` passes over the recording it idles like a live source until tsdrplugin_stop. */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "TSDRCodes.h"
#include "TSDRPlugin.h"

static char g_file[1024];
static uint32_t g_rate;
static long g_block = 524288, g_loops = 0, g_sleep = 0;
static volatile int g_working;
static float *g_data;
static size_t g_floats;
static char g_err[256];
static int g_errcode;

void tsdrplugin_getName(char *name) { strcpy(name, "TSDR in-memory IQ replay"); }

static void unload(void)
{
    free(g_data);
    g_data = NULL;
    g_floats = 0;
}

int tsdrplugin_init(const char *params)
{
    unload();
    g_block = 524288; g_loops = 0; g_sleep = 0;
    const int n = sscanf(params, "%1023s %u %ld %ld %ld", g_file, &g_rate, &g_block, &g_loops, &g_sleep);
    if (n < 2 || g_rate == 0 || g_block <= 0 || (g_block & 1)) {
        snprintf(g_err, sizeof(g_err), "usage: file samplerate [floats_per_block] [loops] [sleep_us]");
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
    const size_t blocks = (size_t)bytes / sizeof(float) / (size_t)g_block;
    if (blocks == 0) {
        fclose(f);
        snprintf(g_err, sizeof(g_err), "%.160s holds less than one block of %ld floats", g_file, g_block);
        return g_errcode = TSDR_PLUGIN_PARAMETERS_WRONG;
    }
    g_floats = blocks * (size_t)g_block;
    /* whole 2 MiB pages, so that every block the library page-locks for DMA has pages of its own */
    if (posix_memalign((void **)&g_data, 2u << 20, ((g_floats * sizeof(float) + (2u << 20) - 1) >> 21) << 21) != 0) g_data = NULL;
    if (!g_data || fread(g_data, sizeof(float), g_floats, f) != g_floats) {
        fclose(f);
        unload();
        snprintf(g_err, sizeof(g_err), "cannot load %.200s into memory", g_file);
        return g_errcode = TSDR_PLUGIN_PARAMETERS_WRONG;
    }
    fclose(f);
    return g_errcode = TSDR_OK;
}

uint32_t tsdrplugin_setsamplerate(uint32_t rate) { (void)rate; return g_rate; }
uint32_t tsdrplugin_getsamplerate(void) { return g_rate; }
int tsdrplugin_setbasefreq(uint32_t freq) { (void)freq; return TSDR_OK; }
int tsdrplugin_setgain(float gain) { (void)gain; return TSDR_OK; }
char *tsdrplugin_getlasterrortext(void) { return g_errcode == TSDR_OK ? NULL : g_err; }
int tsdrplugin_stop(void) { g_working = 0; return TSDR_OK; }
void tsdrplugin_cleanup(void) { unload(); }

int tsdrplugin_readasync(tsdrplugin_readasync_function cb, void *ctx)
{
    if (!g_data) {
        snprintf(g_err, sizeof(g_err), "no recording loaded");
        return g_errcode = TSDR_PLUGIN_PARAMETERS_WRONG;
    }
    g_working = 1;
    const size_t blocks = g_floats / (size_t)g_block;
    for (long pass = 0; g_working && (g_loops == 0 || pass < g_loops); pass++)
        for (size_t b = 0; b < blocks && g_working; b++) {
            cb(g_data + b * (size_t)g_block, (uint64_t)g_block, ctx, 0);
            if (g_sleep > 0) usleep((useconds_t)g_sleep);
        }
    while (g_working) usleep(2000); /* idle like a live source until stopped */
    return g_errcode = TSDR_OK;
}
