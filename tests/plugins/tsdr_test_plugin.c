/* tsdr_test_plugin.c — a deterministic file source implementing the
 * tsdrplugin_* ABI (include/TSDRPlugin.h), for the host-library tests.
 * Unlike the reference's RawFile plugin it does not loop and paces by a fixed
 * sleep per block, and it can inject a "samples dropped" event.
 *
 * params: "<file> <samplerate> <floats_per_block> <sleep_us> [<drop_before_block> <drop_samples> [<as_empty_block>]]"
 * as_empty_block = 1: the drop is reported the way the reference's UHD plugin reports an overflow it could not keep up with,
 * cb(buf, 0, ctx, dropped) (TSDRPlugin_UHD/src/TSDRPlugin_UHD.cpp:294) — an EMPTY block carrying the count — and the next
 * block reports 0.
 * After the file is exhausted it idles (like a live source) until tsdrplugin_stop. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "TSDRCodes.h"
#include "TSDRPlugin.h"

static char g_file[1024];
static uint32_t g_rate;
static long g_block = 524288, g_sleep = 2000, g_drop_at = -1, g_drop_n = 0, g_drop_empty = 0;
static volatile int g_working;
static volatile long g_blocks_sent;
static char g_err[256];
static int g_errcode;

void tsdrplugin_getName(char *name) { strcpy(name, "TSDR deterministic test source"); }

int tsdrplugin_init(const char *params)
{
    g_drop_at = -1;
    g_drop_n = 0;
    g_drop_empty = 0;
    const int n = sscanf(params, "%1023s %u %ld %ld %ld %ld %ld", g_file, &g_rate, &g_block, &g_sleep, &g_drop_at, &g_drop_n, &g_drop_empty);
    if (n < 4 || g_rate == 0 || g_block <= 0 || (g_block & 1)) {
        snprintf(g_err, sizeof(g_err), "usage: file samplerate floats_per_block sleep_us [drop_before_block drop_samples]");
        return g_errcode = TSDR_PLUGIN_PARAMETERS_WRONG;
    }
    return g_errcode = TSDR_OK;
}

uint32_t tsdrplugin_setsamplerate(uint32_t rate) { (void)rate; return g_rate; }
uint32_t tsdrplugin_getsamplerate(void) { return g_rate; }
int tsdrplugin_setbasefreq(uint32_t freq) { (void)freq; return TSDR_OK; }
int tsdrplugin_setgain(float gain) { (void)gain; return TSDR_OK; }
char *tsdrplugin_getlasterrortext(void) { return g_errcode == TSDR_OK ? NULL : g_err; }
int tsdrplugin_stop(void) { g_working = 0; return TSDR_OK; }
void tsdrplugin_cleanup(void) {}

/* test hook: how many blocks have been handed over */
TSDRPLUGIN_API long tsdr_test_plugin_blocks_sent(void) { return g_blocks_sent; }

int tsdrplugin_readasync(tsdrplugin_readasync_function cb, void *ctx)
{
    FILE *f = fopen(g_file, "rb");
    if (!f) {
        snprintf(g_err, sizeof(g_err), "cannot open %s", g_file);
        return g_errcode = TSDR_PLUGIN_PARAMETERS_WRONG;
    }
    float *buf = (float *)malloc(sizeof(float) * (size_t)g_block);
    g_working = 1;
    g_blocks_sent = 0;
    long blk = 0;
    while (g_working) {
        const size_t got = fread(buf, sizeof(float), (size_t)g_block, f);
        if (got < (size_t)g_block) break;
        if (blk == g_drop_at && g_drop_n > 0) fseek(f, g_drop_n * 2 * (long)sizeof(float), SEEK_CUR);
        const int after_drop = blk == g_drop_at + 1 && g_drop_at >= 0;
        if (after_drop && g_drop_empty) cb(buf, 0, ctx, g_drop_n); /* TSDRPlugin_UHD.cpp:294 */
        cb(buf, (uint64_t)g_block, ctx, (after_drop && !g_drop_empty) ? g_drop_n : 0);
        blk++;
        g_blocks_sent = blk;
        if (g_sleep > 0) usleep((useconds_t)g_sleep);
    }
    while (g_working) usleep(2000); /* idle like a live source until stopped */
    free(buf);
    fclose(f);
    return g_errcode = TSDR_OK;
}
