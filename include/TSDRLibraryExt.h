/*
 * TSDRLibraryExt.h — extensions of the tsdr_* API that only this (MI355X) implementation has.  They live
 * behind their own symbols (tsdrx_*): the eighteen tsdr_* entry points of include/TSDRLibrary.h stay exactly the
 * reference's, and a host that never calls anything here sees the reference's behaviour.
 *
 * What they are for: the two conversions that sit right next to the hot path in the reference —
 * the sample-format decode inside the RawFile plugin (TSDRPlugin_RawFile/src/TSDRPlugin_RawFile.c:241-261) and the
 * float -> packed RGB pixel loop inside the Java GUI's JNI shim (JavaGUI/jni/TSDRLibraryNDK.c:222-276) — run on the
 * device here, so narrow samples cross PCIe as they are (2-4x fewer bytes in) and the per-pixel branch chain on the
 * host disappears.
 */
#ifndef TSDR_LIBRARY_EXT_H_
#define TSDR_LIBRARY_EXT_H_

#include <stdint.h>

#include "TSDRLibrary.h"

#ifdef __cplusplus
extern "C" {
#endif

/* One reconstructed frame as width*height packed 0x00RRGGBB pixels, converted exactly like the JNI shim does
 * (gray = (int)(v*255) for 0 < v <= 1, black/white outside, the PIXEL_SPECIAL_VALUE_* debug colours, pixels equal
 * to PIXEL_SPECIAL_VALUE_TRANSPARENT keep the colour the previous frame of the same size left there).  The buffer
 * is the library's and valid during the call. */
typedef void (*tsdrx_readasync_rgb_function)(int32_t *pixels, int width, int height, void *ctx);

/* tsdr_readasync with frames delivered through `cb` as packed RGB; `inverted` as the GUI's invert-colours option
 * (Java_martin_tempest_core_TSDRLibrary_setInvertedColors).  Same blocking / error behaviour as tsdr_readasync. */
int tsdrx_readasync_rgb(tsdr_lib_t *tsdr, tsdrx_readasync_rgb_function cb, void *ctx, int inverted);

/* Counters of the tsdr_readasync session that is running, or of the last one that ended: what went in, what came
 * out, what the (lossy by design, like the reference's rings) queues dropped, and how the detector's certified mode
 * fared.  A host that must not lose frames checks frames_lost_to_viewer == 0. */
typedef struct tsdrx_stats {
    int64_t blocks_in, blocks_lost;              /* plugin callbacks taken / dropped because the input queue was full */
    int64_t frames_made, frames_lost_to_viewer;  /* frames post-processed / not delivered because the viewer was slower */
    int64_t windows;                             /* capture windows correlated */
    int64_t plots_held, epochs_replayed;         /* certified detector: plots held back / epochs replayed exactly */
    int64_t frames_fused;                        /* frames that went through the fused run (backlogs of >= 8 frames) */
} tsdrx_stats_t;
int tsdrx_get_stats(tsdr_lib_t *tsdr, tsdrx_stats_t *out);

/* Sample formats of the optional raw plugin entry point, numbered like tsdrgpu_decode_samples:
 *     int tsdrplugin_readasync_raw(tsdrplugin_readasync_raw_function cb, void *ctx);
 * A source plugin that exports it (besides the ten mandatory tsdrplugin_* symbols) hands its blocks over in their
 * native format; the library decodes them on the device bit-exactly like the RawFile plugin does on the host
 * (double division, float store).  items_count counts values (two per IQ sample), like the float callback's.
 * TSDR_GPU_RAW=0 makes the library ignore the entry point. */
#define TSDRX_SAMPLE_FLOAT32 0
#define TSDRX_SAMPLE_INT8 1
#define TSDRX_SAMPLE_INT16 2
#define TSDRX_SAMPLE_UINT8 3
#define TSDRX_SAMPLE_UINT16 4
typedef void (*tsdrplugin_readasync_raw_function)(const void *buf, uint64_t items_count, int sample_type, void *ctx, int64_t samples_dropped);

/* Optional plugin entry point:
 *     int tsdrplugin_memory_stable(void);
 * A source plugin that exports it and returns non-zero PROMISES that every buffer it hands to the callbacks lies in
 * memory that stays allocated, mapped and at the same address from the moment it first appears until
 * tsdrplugin_cleanup (or the next tsdrplugin_init) — a recording held in memory, one ring allocated at init.  Only
 * then does the library page-lock the plugin's memory and DMA straight out of it (the ranges are unlocked after
 * tsdrplugin_readasync has returned, before anything can call cleanup/init).  Without the promise every block is
 * copied through the library's own pinned buffers, which is what the plugin ABI guarantees to be safe: the
 * reference's plugins (RawFile, Mirics, SDRplay) free their buffer INSIDE tsdrplugin_readasync, before it returns.
 * TSDR_GPU_ZEROCOPY=0 disables the direct path, =1 forces it for a plugin the user knows to be stable.
 * The return value is a set of bits: TSDRX_MEMORY_MAPPED is the promise above (any non-zero value of older plugins is
 * read as this bit alone... so they keep returning 1).  TSDRX_MEMORY_IMMUTABLE adds that the CONTENTS of a block
 * handed to the tsdrplugin_readasync callback do not change for as long as tsdrplugin_readasync runs (a recording, not a
 * ring that hardware refills), TSDRX_MEMORY_IMMUTABLE_RAW the same for tsdrplugin_readasync_raw: only then MAY the library
 * return from the callback while its DMA out of the block is still in flight, so that consecutive blocks' transfers
 * overlap — which it does only when the host opts in with TSDR_GPU_ASYNC_UPLOAD=1: by default every DMA is waited for
 * (measured faster on MI355X: 35-41 against 20-25 GB/s through the upload lane), and the promise changes nothing. */
#define TSDRX_MEMORY_MAPPED 1
#define TSDRX_MEMORY_IMMUTABLE 2
#define TSDRX_MEMORY_IMMUTABLE_RAW 4

#ifdef __cplusplus
}
#endif
#endif
