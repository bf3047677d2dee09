/*
 * TSDRLibrary.h — the tsdr_* API (upper drop-in boundary).
 *
 * Same symbols, argument meaning, return codes and threading behaviour as
 * martinmarinov/TempestSDR's library (TempestSDR/src/include/TSDRLibrary.h:55-76,
 * implementation TempestSDR/src/TSDRLibrary.c), so the JNI shim
 * (JavaGUI/jni/TSDRLibraryNDK.c) and any other host links against this
 * library unchanged.  Behind it the array work runs on an MI355X through
 * include/tsdrgpu.h; there is no CPU path.
 *
 * Threading: tsdr_readasync blocks its caller until tsdr_stop is called from
 * another thread.  The frame, plot and value callbacks fire on threads the
 * library creates (never on the caller's threads); buffers handed to them are
 * owned by the library and valid only during the call.  Setters may be called
 * from any thread at any time.
 */
#ifndef TSDR_LIBRARY_H_
#define TSDR_LIBRARY_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* debug colours a frame may contain besides grey levels 0..1 (TSDRLibrary.h:20-24) */
#define PIXEL_SPECIAL_VALUE_R (256.0f)
#define PIXEL_SPECIAL_VALUE_G (512.0f)
#define PIXEL_SPECIAL_VALUE_B (1024.0f)
#define PIXEL_SPECIAL_VALUE_TRANSPARENT (2048.0f)
#define PIXEL_SPECIAL_COLOURS_ENABLED (1)

/* tsdr_sync directions (TSDRLibrary.h:26-30) */
#define DIRECTION_CUSTOM (0)
#define DIRECTION_UP (1)
#define DIRECTION_DOWN (2)
#define DIRECTION_LEFT (3)
#define DIRECTION_RIGHT (4)

/* tsdr_setparameter_int ids; the Java enum relies on this order
 * (TSDRLibrary.h:32-41, core/TSDRLibrary.java:49) */
#define PARAM_INT_AUTOSHIFT (0)
#define PARAM_INT_FRAMERATE_PLL (1)
#define PARAM_AUTOCORR_PLOTS_RESET (2)
#define PARAM_AUTOCORR_PLOTS_OFF (3)
#define PARAM_AUTOCORR_SUPERRESOLUTION (4)
#define PARAM_NEAREST_NEIGHBOUR_RESAMPLING (5)
#define PARAM_LOW_PASS_BEFORE_SYNC (6)
#define PARAM_AUTOGAIN_AFTER_PROCESSING (7)
#define PARAM_AUTOCORR_DUMP (8)
#define COUNT_PARAM_INT (9)
#define COUNT_PARAM_DOUBLE (2)

/* tsdr_value_changed_callback ids (TSDRLibrary.h:45-50) */
#define VALUE_ID_PLL_FRAMERATE (0)
#define VALUE_ID_AUTOCORRECT_RESET (1)
#define VALUE_ID_AUTOCORRECT_FRAMES_COUNT (2)
#define VALUE_ID_AUTOGAIN_VALUES (3)
#define VALUE_ID_SNR (4)
#define VALUE_ID_AUTOCORRECT_DUMPED (5)

/* tsdr_on_plot_ready_callback ids (TSDRLibrary.h:52-53) */
#define PLOT_ID_FRAME (0)
#define PLOT_ID_LINE (1)

typedef struct tsdr_lib tsdr_lib_t; /* opaque */

/* one reconstructed frame: width*height floats, raster order */
typedef void (*tsdr_readasync_function)(float *buf, int width, int height, void *ctx);
typedef void (*tsdr_value_changed_callback)(int value_id, double arg0, double arg1, void *ctx);
/* averaged |autocorrelation| over lags [offset, offset+size) */
typedef void (*tsdr_on_plot_ready_callback)(int plot_id, int offset, double *values, int size, uint32_t samplerate, void *ctx);

void tsdr_init(tsdr_lib_t **tsdr, tsdr_value_changed_callback callback, tsdr_on_plot_ready_callback plotready_callback, void *ctx);
void tsdr_free(tsdr_lib_t **tsdr);
void *tsdr_getctx(tsdr_lib_t *tsdr);
char *tsdr_getlasterrortext(tsdr_lib_t *tsdr); /* NULL when the last call succeeded */

int tsdr_loadplugin(tsdr_lib_t *tsdr, const char *pluginfilepath, const char *params);
int tsdr_unloadplugin(tsdr_lib_t *tsdr);

int tsdr_setresolution(tsdr_lib_t *tsdr, int height, double refreshrate);
int tsdr_setbasefreq(tsdr_lib_t *tsdr, uint32_t freq);
int tsdr_setgain(tsdr_lib_t *tsdr, float gain);
int tsdr_motionblur(tsdr_lib_t *tsdr, float coeff);
int tsdr_sync(tsdr_lib_t *tsdr, int pixels, int direction);
int tsdr_setparameter_int(tsdr_lib_t *tsdr, int parameter, uint32_t value);
int tsdr_setparameter_double(tsdr_lib_t *tsdr, int parameter, double value);

int tsdr_readasync(tsdr_lib_t *tsdr, tsdr_readasync_function cb, void *ctx); /* blocks */
int tsdr_stop(tsdr_lib_t *tsdr);
int tsdr_isrunning(tsdr_lib_t *tsdr);

/* exported by the reference as well, though not declared in its header
 * (TSDRLibrary.c:118,181) */
int tsdr_getsamplerate(tsdr_lib_t *tsdr);
void tsdr_reset(tsdr_lib_t *tsdr);

#ifdef __cplusplus
}
#endif
#endif
