"""ctypes front-end to oracle/liboracle.so (the CPU restatement, tsdr_oracle.c)
and, when it has been built, oracle/_ref/libtsdr_ref.so (the real reference).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product (tempestsdr_amd/) never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")
_REF = os.path.join(_HERE, "_ref", "libtsdr_ref.so")

f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")


def build():
    """(Re)build liboracle.so and, if /root/reference exists, oracle/_ref."""
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)


def _load():
    if not os.path.exists(_LIB):
        build()
    return C.CDLL(_LIB)


lib = _load()


class Resample(C.Structure):
    _fields_ = [("contrib", C.c_double), ("offset", C.c_double)]


class Autogain(C.Structure):
    _fields_ = [("lastmax", C.c_float), ("lastmin", C.c_float), ("snr", C.c_float)]


class Sweetspot(C.Structure):
    _fields_ = [("dx", C.c_int), ("vx", C.c_int), ("absvx", C.c_int), ("curr_stripsize", C.c_int)]


class Geometry(C.Structure):
    _fields_ = [("samplerate", C.c_uint32), ("width", C.c_int), ("height", C.c_int),
                ("refreshrate", C.c_double), ("pixelrate", C.c_double),
                ("pixeltimeoversampletime", C.c_double)]


class SyncDetector(C.Structure):
    _fields_ = [("db_x", Sweetspot), ("db_y", Sweetspot), ("last_frame_diff", C.c_double),
                ("state", C.c_int), ("avg_speed", C.c_double)]


def _sig(fn, res, *args):
    fn.restype = res
    fn.argtypes = list(args)
    return fn


_sig(lib.orc_am_demod, None, f32p, f32p, C.c_int64)
_sig(lib.orc_resample_init, None, C.POINTER(Resample))
_sig(lib.orc_resample_count, C.c_uint32, C.POINTER(Resample), C.c_uint32, C.c_double, C.c_double)
_sig(lib.orc_resample_process, C.c_uint32, C.POINTER(Resample), f32p, C.c_uint32, f32p,
     C.c_double, C.c_double, C.c_int, C.POINTER(C.c_uint32))
_sig(lib.orc_timelowpass_run, None, C.c_float, C.c_int, f32p, f32p)
_sig(lib.orc_autogain_init, None, C.POINTER(Autogain))
_sig(lib.orc_autogain_run, None, C.POINTER(Autogain), C.c_int, f32p, f32p, C.c_float)
_sig(lib.orc_average_v_h, None, C.c_int, C.c_int, f32p, f32p, f32p)
_sig(lib.orc_gaussian_taps, None, f32p)
_sig(lib.orc_gaussianblur, None, f32p, C.c_int)
_sig(lib.orc_findthesweetspot, None, C.POINTER(Sweetspot), f32p, C.c_int, C.c_int, C.c_double)
_sig(lib.orc_set_internal_samplerate, None, C.POINTER(Geometry), C.c_uint32)
_sig(lib.orc_syncdetector_init, None, C.POINTER(SyncDetector))
_sig(lib.orc_frameratepll, C.c_int, C.POINTER(SyncDetector), C.POINTER(Geometry), C.c_int)
_sig(lib.orc_postprocess_new, C.c_void_p)
_sig(lib.orc_postprocess_free, None, C.c_void_p)
_sig(lib.orc_post_process, C.POINTER(C.c_float), C.c_void_p, C.POINTER(Geometry), f32p, C.c_int,
     C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int)
_sig(lib.orc_postprocess_state, None, C.c_void_p, i32p, f64p)
_sig(lib.orc_postprocess_colsum, C.POINTER(C.c_float), C.c_void_p)
_sig(lib.orc_postprocess_rowsum, C.POINTER(C.c_float), C.c_void_p)
_sig(lib.orc_fft_getrealsize, C.c_uint32, C.c_uint32)
_sig(lib.orc_fft_perform, None, f32p, C.c_uint32, C.c_int)
_sig(lib.orc_fft_autocorrelation, None, f32p, f32p, C.c_uint32)
_sig(lib.orc_accumulate, None, f64p, f32p, C.c_int, C.c_int, C.c_uint64)
_sig(lib.orc_lag_windows, None, C.c_uint32, i32p)
_sig(lib.orc_capture_size, C.c_uint32, C.c_uint32)
_sig(lib.orc_fft_crosscorrelation, None, f32p, f32p, C.c_uint32)
_sig(lib.orc_complex_to_abs_diff, None, f32p, C.c_int)
_sig(lib.orc_superb_bestfit, C.c_int, f32p, f32p, C.c_int, C.c_int)
_sig(lib.orc_superb_stitch, C.c_uint32, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, f32p,
     i32p)
_sig(lib.orc_decode_samples, None, C.c_void_p, C.c_int, f32p, C.c_int64)
_sig(lib.orc_frame_to_rgb, None, f32p, i32p, C.c_int64, C.c_int)


class PlotScale(C.Structure):
    _fields_ = [("one_val_in_pixels", C.c_double), ("one_px_in_values", C.c_double), ("offset_val", C.c_double),
                ("min_value", C.c_double), ("offset_px", C.c_int)]


_sig(lib.orc_plotscale_default, None, C.c_int, C.c_int, C.POINTER(PlotScale))
_sig(lib.orc_plot_populate, None, f64p, C.c_int, C.c_int, C.POINTER(PlotScale), f64p, C.POINTER(C.c_double),
     C.POINTER(C.c_double), C.POINTER(C.c_int))
_sig(lib.orc_dropped_shift_with, C.c_int64, C.c_int64, C.c_uint32, C.c_int64)
_sig(lib.orc_dropped_add, C.c_int64, C.c_int64, C.c_uint32, C.c_uint32, C.c_int,
     C.POINTER(C.c_uint32), C.POINTER(C.c_uint32))


# --------------------------------------------------------------------------
# numpy-level helpers (what the tests call)
# --------------------------------------------------------------------------
def am_demod(iq):
    iq = np.ascontiguousarray(iq, np.float32)
    out = np.empty(iq.size // 2, np.float32)
    lib.orc_am_demod(iq, out, out.size)
    return out


def geometry(samplerate, height, refreshrate):
    g = Geometry()
    g.height = int(height)
    g.refreshrate = float(refreshrate)
    lib.orc_set_internal_samplerate(C.byref(g), int(samplerate))
    return g


class Resampler:
    """Stateful dsp_resample_process restatement (dsp.c:250-307)."""

    def __init__(self):
        self.st = Resample()
        lib.orc_resample_init(C.byref(self.st))

    def process(self, x, up, down, nearest=False):
        x = np.ascontiguousarray(x, np.float32)
        n = lib.orc_resample_count(C.byref(self.st), x.size, up, down)
        out = np.zeros(n + 2, np.float32)
        emitted = C.c_uint32(0)
        lib.orc_resample_process(C.byref(self.st), x, x.size, out, up, down, int(nearest),
                                 C.byref(emitted))
        self.last_emitted = emitted.value
        return out[:n]


def chunk_size(samplerate, refreshrate):
    """decimatingthread chunk: (int)(0.1*fs/refresh), TSDRLibrary.c:335."""
    return int(0.1 * samplerate / refreshrate)


def demod_resample_stream(iq, g, nearest=False, resampler=None):
    """Deterministic single-threaded driver of SURVEY §8(c): am_demod, then
    dsp_resample_process per 0.1-frame chunk.  Returns the pixel stream and the
    per-chunk (offset, contrib, count) trace."""
    mag = am_demod(iq)
    rs = resampler or Resampler()
    chunk = chunk_size(g.samplerate, g.refreshrate)
    up = g.width * g.height * g.refreshrate
    down = float(g.samplerate)
    outs, trace = [], []
    for s in range(0, mag.size - chunk + 1, chunk):
        off, con = rs.st.offset, rs.st.contrib
        o = rs.process(mag[s:s + chunk], up, down, nearest)
        trace.append((off, con, o.size))
        outs.append(o)
    return (np.concatenate(outs) if outs else np.zeros(0, np.float32)), trace


class PostProcess:
    """dsp_post_process restatement with its own state (dsp.c:134-239)."""

    def __init__(self, geom):
        self.h = lib.orc_postprocess_new()
        self.g = geom

    def __del__(self):
        if getattr(self, "h", None):
            lib.orc_postprocess_free(self.h)
            self.h = None

    def run(self, frame, motionblur=0.0, lowpasscoeff=0.1, lowpass_before_sync=0,
            autogain_after_proc=0, autoshift=0, pll=0, superres=0):
        g = self.g
        w, h = g.width, g.height
        frame = np.ascontiguousarray(frame, np.float32)
        assert frame.size == w * h
        self._keep = frame  # may be painted in place, like the reference
        p = lib.orc_post_process(self.h, C.byref(g), frame, w, h, motionblur, lowpasscoeff,
                                 lowpass_before_sync, autogain_after_proc, autoshift, pll, superres)
        return np.ctypeslib.as_array(p, shape=(w * h,)).copy()

    def state(self):
        oi = np.zeros(10, np.int32)
        od = np.zeros(4, np.float64)
        lib.orc_postprocess_state(self.h, oi, od)
        return oi, od

    def strips(self):
        w, h = self.g.width, self.g.height
        return (np.ctypeslib.as_array(lib.orc_postprocess_colsum(self.h), shape=(w,)).copy(),
                np.ctypeslib.as_array(lib.orc_postprocess_rowsum(self.h), shape=(h,)).copy())


def fft_perform(z, inverse):
    z = np.ascontiguousarray(z, np.float32).copy()
    lib.orc_fft_perform(z, z.size // 2, int(inverse))
    return z


def fft_autocorrelation(x):
    x = np.ascontiguousarray(x, np.float32)
    ans = np.zeros(2 * x.size, np.float32)
    lib.orc_fft_autocorrelation(ans, x, x.size)
    return ans


def lag_windows(samplerate):
    w = np.zeros(4, np.int32)
    lib.orc_lag_windows(int(samplerate), w)
    return tuple(int(v) for v in w)  # frame_lo, frame_len, line_lo, line_len


def capture_size(samplerate):
    return int(lib.orc_capture_size(int(samplerate)))


class Autocorr:
    """frameratedetector_runontodata restatement (frameratedetector.c:87-126):
    running mean of |R| over the frame- and line-lag windows."""

    def __init__(self, samplerate):
        self.fs = int(samplerate)
        self.flo, self.flen, self.llo, self.llen = lag_windows(samplerate)
        self.reset()

    def reset(self):
        self.calls = 0
        self.frame = np.zeros(self.flen, np.float64)
        self.line = np.zeros(self.llen, np.float64)

    def run(self, window):
        corr = fft_autocorrelation(window)
        self.calls += 1
        lib.orc_accumulate(self.frame, corr, self.flo, self.flen, self.calls)
        lib.orc_accumulate(self.line, corr, self.llo, self.llen, self.calls)
        return corr


def superb_stitch(hops, samples_in_frame):
    """hops: list of complex64-as-float32 arrays (2*gathered floats each)."""
    hops = [np.ascontiguousarray(h, np.float32).copy() for h in hops]
    gathered = hops[0].size // 2
    per = lib.orc_fft_getrealsize(gathered)
    out = np.zeros(len(hops) * per * 2, np.float32)
    ptrs = (C.c_void_p * len(hops))(*[h.ctypes.data for h in hops])
    offs = np.zeros(len(hops), np.int32)
    n = lib.orc_superb_stitch(ptrs, len(hops), gathered, samples_in_frame, out, offs)
    return out[:2 * n], offs


# --------------------------------------------------------------------------
# the real reference (present only where oracle/_ref was built)
# --------------------------------------------------------------------------
def have_ref():
    return os.path.exists(_REF)


_ref = None


def ref():
    """ctypes handle of the compiled reference + shim (oracle/_ref/libtsdr_ref.so)."""
    global _ref
    if _ref is not None:
        return _ref
    r = C.CDLL(_REF)
    _sig(r.ref_new, C.c_void_p, C.c_int, C.c_double, C.c_uint32, C.c_float, C.c_void_p)
    _sig(r.ref_free, None, C.c_void_p)
    _sig(r.ref_width, C.c_int, C.c_void_p)
    _sig(r.ref_height, C.c_int, C.c_void_p)
    _sig(r.ref_refreshrate, C.c_double, C.c_void_p)
    _sig(r.ref_pixelrate, C.c_double, C.c_void_p)
    _sig(r.ref_pixeltimeoversampletime, C.c_double, C.c_void_p)
    _sig(r.ref_setparam, None, C.c_void_p, C.c_int, C.c_uint32)
    _sig(r.ref_post_process, C.POINTER(C.c_float), C.c_void_p, f32p, C.c_float, C.c_float, C.c_int,
         C.c_int)
    _sig(r.ref_postprocess_state, None, C.c_void_p, i32p, f64p)
    _sig(r.ref_colsum, C.POINTER(C.c_float), C.c_void_p)
    _sig(r.ref_rowsum, C.POINTER(C.c_float), C.c_void_p)
    _sig(r.ref_resampler_new, C.c_void_p)
    _sig(r.ref_resampler_free, None, C.c_void_p)
    _sig(r.ref_resampler_state, None, C.c_void_p, f64p)
    _sig(r.ref_resampler_setstate, None, C.c_void_p, C.c_double, C.c_double)
    _sig(r.ref_resampler_process, C.c_uint32, C.c_void_p, f32p, C.c_uint32, f32p, C.c_double,
         C.c_double, C.c_int)
    _sig(r.ref_accumulate, None, f64p, f32p, C.c_int, C.c_int, C.c_uint64)
    _sig(r.ref_dump_autocorr, C.c_int, f32p, C.c_int, C.c_double)
    _sig(r.ref_superb_stitch, C.c_uint32, C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int,
         C.c_int, C.c_uint32, f32p)
    # reference stage symbols that take plain pointers are called directly
    _sig(r.complex_to_real, None, f32p, C.c_int)
    _sig(r.fft_perform, None, f32p, C.c_uint32, C.c_int)
    _sig(r.fft_autocorrelation, None, f32p, f32p, C.c_uint32)
    _sig(r.fft_crosscorrelation, None, f32p, f32p, C.c_uint32)
    _sig(r.fft_getrealsize, C.c_uint32, C.c_uint32)
    _sig(r.complex_to_abs_diff, None, f32p, C.c_int)
    _sig(r.gaussianblur, None, f32p, C.c_int)
    _sig(r.findthesweetspot, None, C.POINTER(Sweetspot), f32p, C.c_int, C.c_int, C.c_double)
    _sig(r.dsp_autogain_run, None, C.POINTER(Autogain), C.c_int, f32p, f32p, C.c_float)
    _sig(r.dsp_average_v_h, None, C.c_int, C.c_int, f32p, f32p, f32p)
    _sig(r.dsp_timelowpass_run, None, C.c_float, C.c_int, f32p, f32p)
    _sig(r.dsp_dropped_compensation_shift_with, None, C.POINTER(C.c_int64), C.c_uint32, C.c_int64)
    _ref = r
    return r


_REF_JNI = os.path.join(_HERE, "_ref", "libtsdr_ref_jni.so")
_ref_jni = None


def have_ref_jni():
    return os.path.exists(_REF_JNI)


def ref_jni():
    """ctypes handle of the reference's JNI shim compiled behind oracle/jni_stub (its frame -> RGB pixel loop)."""
    global _ref_jni
    if _ref_jni is None:
        r = C.CDLL(_REF_JNI)
        _sig(r.ref_jni_frame_to_rgb, None, f32p, C.c_int, C.c_int, C.c_int, i32p)
        _sig(r.ref_jni_reset, None)
        _ref_jni = r
    return _ref_jni


def plot_populate(data, nwidth, scale=None):
    """PlotVisualizer.populateData: (visdata, lowest, highest, max_index); scale=None = unzoomed."""
    data = np.ascontiguousarray(data, np.float64)
    if scale is None:
        scale = PlotScale()
        lib.orc_plotscale_default(data.size, nwidth, C.byref(scale))
    vis = np.empty(nwidth, np.float64)
    lo, hi, mi = C.c_double(), C.c_double(), C.c_int()
    lib.orc_plot_populate(data, data.size, nwidth, C.byref(scale), vis, C.byref(lo), C.byref(hi), C.byref(mi))
    return vis, lo.value, hi.value, mi.value
