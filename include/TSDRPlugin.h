/*
 * TSDRPlugin.h — the source-plugin ABI this library loads (lower drop-in
 * boundary).  Ten C symbols resolved by name with dlopen/dlsym, exactly the set
 * martinmarinov/TempestSDR plugins export (TempestSDR/src/include/TSDRPlugin.h:49-60,
 * resolved in TSDRPluginLoader.c:57-68), so existing plugin binaries
 * (RawFile, UHD, Mirics, SDRplay) load unchanged.
 *
 * Contract reminders:
 *  - tsdrplugin_readasync blocks and calls `cb` on its own thread with
 *    interleaved float32 I,Q; items_count = number of floats (even);
 *    samples_dropped = samples lost before this block.  The buffer belongs to
 *    the plugin and is only valid during the call.
 *  - tsdrplugin_init may write into its parameter string (RawFile does,
 *    TSDRPlugin_RawFile.c:145): the host passes a private writable copy.
 */
#ifndef TSDR_PLUGIN_H_
#define TSDR_PLUGIN_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(_WIN32) || defined(__CYGWIN__)
#define TSDRPLUGIN_API __declspec(dllexport)
#else
#define TSDRPLUGIN_API __attribute__((visibility("default")))
#ifndef __stdcall
#define __stdcall
#endif
#endif

typedef void (*tsdrplugin_readasync_function)(float *buf, uint64_t items_count, void *ctx, int64_t samples_dropped);

TSDRPLUGIN_API void __stdcall tsdrplugin_getName(char *name /* >= 200 bytes */);
TSDRPLUGIN_API int __stdcall tsdrplugin_init(const char *params);
TSDRPLUGIN_API uint32_t __stdcall tsdrplugin_setsamplerate(uint32_t rate);
TSDRPLUGIN_API uint32_t __stdcall tsdrplugin_getsamplerate(void);
TSDRPLUGIN_API int __stdcall tsdrplugin_setbasefreq(uint32_t freq);
TSDRPLUGIN_API int __stdcall tsdrplugin_stop(void);
TSDRPLUGIN_API int __stdcall tsdrplugin_setgain(float gain /* 0..1 */);
TSDRPLUGIN_API char *__stdcall tsdrplugin_getlasterrortext(void);
TSDRPLUGIN_API int __stdcall tsdrplugin_readasync(tsdrplugin_readasync_function cb, void *ctx);
TSDRPLUGIN_API void __stdcall tsdrplugin_cleanup(void);

#ifdef __cplusplus
}
#endif
#endif
