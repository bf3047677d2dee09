/*
 * TSDRCodes.h — status codes shared by the tsdr_* API and the tsdrplugin_* ABI.
 * Values are the wire contract of martinmarinov/TempestSDR
 * (TempestSDR/src/include/TSDRCodes.h:16-27): the JNI shim maps them to Java
 * exception classes (JavaGUI/jni/TSDRLibraryNDK.c:47-88) and source plugins
 * return them, so they must not change.
 */
#ifndef TSDR_CODES_H_
#define TSDR_CODES_H_

enum {
    TSDR_OK = 0,
    TSDR_ERR_PLUGIN = 1,              /* plugin missing / failed */
    TSDR_WRONG_VIDEOPARAMS = 2,       /* impossible height / refresh rate / shift */
    TSDR_ALREADY_RUNNING = 3,         /* call not allowed while tsdr_readasync runs */
    TSDR_PLUGIN_PARAMETERS_WRONG = 4, /* plugin rejected its parameter string */
    TSDR_SAMPLE_RATE_WRONG = 5,
    TSDR_CANNOT_OPEN_DEVICE = 6,
    TSDR_INCOMPATIBLE_PLUGIN = 7,     /* dlopen failed (missing dependencies, wrong arch) */
    TSDR_INVALID_PARAMETER = 8,
    TSDR_INVALID_PARAMETER_VALUE = 9,
    TSDR_NOT_RUNNING = 10,
    TSDR_NOT_IMPLEMENTED = 404
};

#endif
