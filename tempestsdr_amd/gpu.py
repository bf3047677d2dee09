"""ctypes binding of tempestsdr_amd/libtsdrgpu.so (C ABI: include/tsdrgpu.h).

This is the host-side mirror used by the tests and bench.py; the method names
follow the reference's stage functions (am_demod, dsp_resample_process,
dsp_post_process, fft_perform, fft_autocorrelation/accummulate,
superb_ondataready) so the parity tests read like calls into the reference.

There is NO fallback: if the HIP library is missing or no GPU is present the
constructor raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# TSDRGPU_LIB: another build of the SAME library (same-box A/B of two kernel versions in one gpurun call); never a fallback
LIB_PATH = os.environ.get("TSDRGPU_LIB") or os.path.join(_HERE, "libtsdrgpu.so")

vp = C.c_void_p


class PPParams(C.Structure):
    _fields_ = [("lowpass_before_sync", C.c_int), ("autogain_after_proc", C.c_int),
                ("autoshift", C.c_int), ("pll", C.c_int), ("superresolution", C.c_int),
                ("motionblur", C.c_float), ("lowpasscoeff", C.c_float)]


class BandExchange(C.Structure):
    """tsdrgpu_band_exchange_t: the collective a general band run asks its caller for"""
    _fields_ = [("kind", C.c_int), ("d_buf", C.c_void_p), ("count", C.c_int64)]


BAND_DONE, BAND_SUM_F64, BAND_MAX_F32, BAND_ALLGATHER_F32 = 0, 1, 2, 3


class ProfileEntry(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("total_ms", C.c_double), ("launches", C.c_int)]


class Detection(C.Structure):
    _fields_ = [("frame_lag", C.c_int), ("line_lag", C.c_int), ("framerate", C.c_double), ("linerate", C.c_double),
                ("height", C.c_int), ("pixelrate", C.c_double), ("seen", C.c_int), ("accepted", C.c_int),
                ("mode_id", C.c_int), ("mode_name", C.c_char * 48), ("mode_width", C.c_int),
                ("mode_height", C.c_int), ("mode_refresh", C.c_double)]


class PlotScale(C.Structure):
    """State of the GUI's ZoomableXScale (gui/scale/ZoomableXScale.java:24-31)."""
    _fields_ = [("one_val_in_pixels", C.c_double), ("one_px_in_values", C.c_double), ("offset_val", C.c_double),
                ("min_value", C.c_double), ("offset_px", C.c_int)]


class AcCertificate(C.Structure):
    """tsdrgpu_ac_certificate_t: the argmax certificate of the certified autocorrelation mode."""
    _fields_ = [("frame_certified", C.c_int), ("line_certified", C.c_int), ("exact_epoch", C.c_int), ("promotions", C.c_int),
                ("frame_best", C.c_double), ("frame_runner_up", C.c_double), ("line_best", C.c_double),
                ("line_runner_up", C.c_double), ("r0", C.c_double), ("margin", C.c_double),
                ("premise_checked", C.c_int), ("premise_ok", C.c_int), ("premise_err", C.c_double), ("premise_r0", C.c_double),
                ("premise_checks", C.c_long), ("premise_failures", C.c_long)]


class PPFrameInfo(C.Structure):
    _fields_ = [("lastmin", C.c_float), ("lastmax", C.c_float),
                ("dx", C.c_int), ("vx", C.c_int), ("stripx", C.c_int),
                ("dy", C.c_int), ("vy", C.c_int), ("stripy", C.c_int),
                ("locked", C.c_int), ("pll_fired", C.c_int),
                ("avg_speed", C.c_double), ("frameratediff", C.c_double)]


_SIGS = {
    "tsdrgpu_create": (C.c_int, [C.POINTER(vp), C.c_int]),
    "tsdrgpu_destroy": (None, [vp]),
    "tsdrgpu_last_error": (C.c_char_p, [vp]),
    "tsdrgpu_sync": (C.c_int, [vp]),
    "tsdrgpu_stream": (vp, [vp]),
    "tsdrgpu_device_name": (C.c_int, [vp, C.c_char_p, C.c_size_t]),
    "tsdrgpu_alloc": (C.c_int, [vp, C.POINTER(vp), C.c_size_t]),
    "tsdrgpu_free": (C.c_int, [vp, vp]),
    "tsdrgpu_alloc_host": (C.c_int, [vp, C.POINTER(vp), C.c_size_t]),
    "tsdrgpu_free_host": (C.c_int, [vp, vp]),
    "tsdrgpu_upload": (C.c_int, [vp, vp, vp, C.c_size_t]),
    "tsdrgpu_download": (C.c_int, [vp, vp, vp, C.c_size_t]),
    "tsdrgpu_copy": (C.c_int, [vp, vp, vp, C.c_size_t]),
    "tsdrgpu_frame_snr": (C.c_int, [vp, vp, C.c_int64, C.POINTER(C.c_float)]),
    "tsdrgpu_postproc_set_snr": (C.c_int, [vp, C.c_int]),
    "tsdrgpu_postproc_snr": (C.c_int, [vp, C.POINTER(C.c_float), C.c_int]),
    "tsdrgpu_copy2": (C.c_int, [vp, vp, vp, vp, C.c_size_t]),
    "tsdrgpu_gather2": (C.c_int, [vp, vp, vp, C.POINTER(vp), C.POINTER(C.c_size_t), C.c_int]),
    "tsdrgpu_event_create": (C.c_int, [vp, C.POINTER(vp)]),
    "tsdrgpu_event_destroy": (None, [vp, vp]),
    "tsdrgpu_event_record": (C.c_int, [vp, vp, C.c_int]),
    "tsdrgpu_lane_wait": (C.c_int, [vp, C.c_int, vp]),
    "tsdrgpu_event_sync": (C.c_int, [vp, vp]),
    "tsdrgpu_event_done": (C.c_int, [vp, vp]),
    "tsdrgpu_lane_sync": (C.c_int, [vp, C.c_int]),
    "tsdrgpu_upload_lane": (C.c_int, [vp, vp, vp, C.c_size_t]),
    "tsdrgpu_download_lane": (C.c_int, [vp, vp, vp, C.c_size_t]),
    "tsdrgpu_host_register": (C.c_int, [vp, vp, C.c_size_t]),
    "tsdrgpu_host_unregister": (C.c_int, [vp, vp]),
    "tsdrgpu_bind_thread": (C.c_int, [vp]),
    "tsdrgpu_postproc_info_pack": (C.c_int, [vp, vp, C.c_int]),
    "tsdrgpu_postproc_band_begin": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.POINTER(vp), C.POINTER(C.c_int64),
                                              C.POINTER(vp), C.POINTER(C.c_int64)]),
    "tsdrgpu_postproc_band_finish": (C.c_int, [vp, vp, vp]),
    "tsdrgpu_postproc_band_open": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int, vp]),
    "tsdrgpu_postproc_band_step": (C.c_int, [vp, vp, vp, vp]),
    "tsdrgpu_postproc_band_advance": (C.c_int, [vp, vp, C.c_int, C.c_int, C.POINTER(vp), C.POINTER(C.c_int64), C.POINTER(C.c_int), vp]),
    "tsdrgpu_postproc_band_begin_minmax": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, C.POINTER(vp), C.POINTER(C.c_int64)]),
    "tsdrgpu_postproc_band_fused": (C.c_int, [vp, vp, C.POINTER(vp), C.POINTER(C.c_int64)]),
    "tsdrgpu_postproc_band_spec_stats": (C.c_int, [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "tsdrgpu_resample_band": (C.c_int, [vp, vp, C.c_int, C.c_uint32, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_int64, vp, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int)]),
    "tsdrgpu_comm_allreduce_f32max": (C.c_int, [vp, vp, C.c_int64, C.c_int]),
    "tsdrgpu_autocorr_plots_async": (C.c_int, [vp, vp, vp, C.POINTER(C.c_uint64)]),
    "tsdrgpu_autocorr_plots_snapshot": (C.c_int, [vp, C.POINTER(vp), C.POINTER(C.c_uint64)]),
    "tsdrgpu_autocorr_lane": (C.c_int, [vp]),
    "tsdrgpu_rccl_unique_id": (C.c_int, [vp]),
    "tsdrgpu_comm_create": (C.c_int, [vp, C.POINTER(vp), C.c_int, C.c_int, vp]),
    "tsdrgpu_comm_destroy": (None, [vp]),
    "tsdrgpu_comm_count": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "tsdrgpu_comm_allreduce_f64": (C.c_int, [vp, vp, C.c_int64, C.c_int]),
    "tsdrgpu_autocorr_allreduce": (C.c_int, [vp, vp, C.c_uint64]),
    "tsdrgpu_comm_broadcast_f32": (C.c_int, [vp, vp, C.c_int64, C.c_int, C.c_int]),
    "tsdrgpu_comm_allgather_f32": (C.c_int, [vp, vp, C.c_int64, C.c_int]),
    "tsdrgpu_superb_shard_create": (C.c_int, [vp, C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int]),
    "tsdrgpu_superb_shard_destroy": (None, [vp]),
    "tsdrgpu_superb_shard_reference": (C.c_int, [vp, vp, C.POINTER(vp), C.POINTER(C.c_int64)]),
    "tsdrgpu_superb_shard_spectrum": (C.c_int, [vp, vp, C.POINTER(vp), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "tsdrgpu_superb_shard_finish": (C.c_int, [vp, vp, C.POINTER(C.c_uint32)]),
    "tsdrgpu_zero": (C.c_int, [vp, vp, C.c_size_t]),
    "tsdrgpu_timer_start": (C.c_int, [vp]),
    "tsdrgpu_timer_stop_ms": (C.c_int, [vp, C.POINTER(C.c_float)]),
    "tsdrgpu_profile_begin": (C.c_int, [vp]),
    "tsdrgpu_profile_end": (C.c_int, [vp, C.POINTER(ProfileEntry), C.c_int, C.POINTER(C.c_int)]),
    "tsdrgpu_am_demod": (C.c_int, [vp, vp, vp, C.c_int64]),
    "tsdrgpu_resampler_create": (C.c_int, [vp, C.POINTER(vp)]),
    "tsdrgpu_resampler_destroy": (None, [vp]),
    "tsdrgpu_resampler_reset": (C.c_int, [vp]),
    "tsdrgpu_resampler_setstate": (C.c_int, [vp, C.c_double, C.c_double]),
    "tsdrgpu_resampler_getstate": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "tsdrgpu_resample": (C.c_int, [vp, vp, C.c_int, C.c_uint32, C.c_int, C.c_double, C.c_double,
                                   C.c_int, vp, C.c_int64, C.POINTER(C.c_int64)]),
    "tsdrgpu_resampler_track_frames": (C.c_int, [vp, C.c_int64, C.c_int64]),
    "tsdrgpu_resampler_frame_minmax": (C.c_int, [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_int)]),
    "tsdrgpu_resample_count": (C.c_int64, [vp, C.c_uint32, C.c_int, C.c_double, C.c_double]),
    "tsdrgpu_postproc_create": (C.c_int, [vp, C.POINTER(vp)]),
    "tsdrgpu_postproc_destroy": (None, [vp]),
    "tsdrgpu_postproc_reset": (C.c_int, [vp]),
    "tsdrgpu_postproc_set_exact_ties": (C.c_int, [vp, C.c_int]),
    "tsdrgpu_postproc_redo_stats": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "tsdrgpu_postproc_redo_raw": (C.c_int, [vp, vp, C.c_int, C.POINTER(C.c_int)]),
    "tsdrgpu_postproc_run": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.POINTER(PPParams), vp,
                                       C.POINTER(PPFrameInfo)]),
    "tsdrgpu_postproc_begin": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.POINTER(PPParams)]),
    "tsdrgpu_postproc_begin_minmax": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.POINTER(PPParams), vp, vp, vp]),
    "tsdrgpu_postproc_finish": (C.c_int, [vp, vp, C.POINTER(PPFrameInfo)]),
    "tsdrgpu_postproc_strips": (C.c_int, [vp, vp, vp]),
    "tsdrgpu_fft": (C.c_int, [vp, vp, C.c_uint32, C.c_int]),
    "tsdrgpu_fft_exact": (C.c_int, [vp, vp, C.c_uint32, C.c_int]),
    "tsdrgpu_autocorr_create": (C.c_int, [vp, C.POINTER(vp), C.c_uint32]),
    "tsdrgpu_autocorr_destroy": (None, [vp]),
    "tsdrgpu_autocorr_reset": (C.c_int, [vp]),
    "tsdrgpu_autocorr_geometry": (C.c_int, [vp] + [C.POINTER(C.c_int32)] * 4 + [C.POINTER(C.c_uint32)] * 2),
    "tsdrgpu_autocorr_run": (C.c_int, [vp, vp, C.c_int, C.c_int64, C.c_int, C.c_int]),
    "tsdrgpu_autocorr_set_async": (C.c_int, [vp, C.c_int]),
    "tsdrgpu_autocorr_plots": (C.c_int, [vp, vp, vp, C.POINTER(C.c_uint64)]),
    "tsdrgpu_autocorr_device_plots": (C.c_int, [vp, C.POINTER(vp), C.POINTER(C.c_int64)]),
    "tsdrgpu_autocorr_promote_step": (C.c_int, [vp, C.c_int, C.POINTER(C.c_int)]),
    "tsdrgpu_autocorr_retention_reserve": (C.c_int, [vp, C.c_int, C.c_int]),
    "tsdrgpu_autocorr_retention": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "tsdrgpu_autocorr_device_sums": (C.c_int, [vp, C.POINTER(vp), C.POINTER(C.c_int64)]),
    "tsdrgpu_autocorr_finalize_sums": (C.c_int, [vp, C.c_uint64]),
    "tsdrgpu_autocorr_set_exact": (C.c_int, [vp, C.c_int]),
    "tsdrgpu_autocorr_set_plan": (C.c_int, [vp, C.c_int]),
    "tsdrgpu_autocorr_argmax_async": (C.c_int, [vp]),
    "tsdrgpu_autocorr_argmax_result": (C.c_int, [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "tsdrgpu_autocorr_argmax": (C.c_int, [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "tsdrgpu_autocorr_last_corr": (C.c_int, [vp, C.POINTER(vp), C.POINTER(C.c_uint32)]),
    "tsdrgpu_autocorr_set_certify": (C.c_int, [vp, C.c_int, C.c_size_t]),
    "tsdrgpu_autocorr_promote": (C.c_int, [vp]),
    "tsdrgpu_autocorr_certificate": (C.c_int, [vp, vp]),
    "tsdrgpu_autocorr_argmax_certified": (C.c_int, [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int)]),
    "tsdrgpu_superb_stitch": (C.c_int, [vp, C.POINTER(vp), C.c_int, C.c_int, C.c_int, vp,
                                        C.POINTER(C.c_int32), C.POINTER(C.c_uint32)]),
    "tsdrgpu_superb_set_plan": (C.c_int, [vp, C.c_int]),
    "tsdrgpu_superb_stitch_exact": (C.c_int, [vp, C.POINTER(vp), C.c_int, C.c_int, C.c_int, vp,
                                        C.POINTER(C.c_int32), C.POINTER(C.c_uint32)]),
    "tsdrgpu_decode_samples": (C.c_int, [vp, vp, C.c_int, vp, C.c_int64]),
    "tsdrgpu_frame_to_rgb": (C.c_int, [vp, vp, vp, C.c_int64, C.c_int]),
    "tsdrgpu_modedetect_create": (C.c_int, [C.POINTER(vp)]),
    "tsdrgpu_modedetect_destroy": (None, [vp]),
    "tsdrgpu_modedetect_reset": (None, [vp]),
    "tsdrgpu_plotscale_default": (None, [C.c_int, C.c_int, C.POINTER(PlotScale)]),
    "tsdrgpu_plot_columns": (C.c_int, [vp, vp, C.c_int, C.c_int, C.POINTER(PlotScale), vp, C.POINTER(C.c_double),
                                       C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "tsdrgpu_modedetect_feed": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32, C.POINTER(Detection)]),
}

_lib = None


def load_library():
    """dlopen the in-tree HIP library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `python -m tempestsdr_amd.build` "
                               "(there is no CPU fallback)")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(lib, name)  # AttributeError = the ABI lost a symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def exported_symbols():
    return sorted(_SIGS)


class TsdrGpuError(RuntimeError):
    pass


class DeviceArray:
    """A typed device allocation owned by a TsdrGpu context."""

    def __init__(self, ctx, count, dtype):
        self.ctx = ctx
        self.dtype = np.dtype(dtype)
        self.count = int(count)
        p = vp()
        ctx._ck(ctx.lib.tsdrgpu_alloc(ctx.h, C.byref(p), max(1, self.count) * self.dtype.itemsize))
        self.ptr = p.value
        ctx._adopt(self)

    def destroy(self):
        self.free()

    @property
    def nbytes(self):
        return self.count * self.dtype.itemsize

    def at(self, elem_offset):
        return self.ptr + int(elem_offset) * self.dtype.itemsize

    def upload(self, host, elem_offset=0):
        host = np.ascontiguousarray(host, self.dtype)
        assert elem_offset + host.size <= self.count
        self.ctx._ck(self.ctx.lib.tsdrgpu_upload(self.ctx.h, self.at(elem_offset), host.ctypes.data, host.nbytes))
        self.ctx.sync()  # pageable source: keep it alive until the copy is done
        return self

    def download(self, count=None, elem_offset=0):
        count = self.count - elem_offset if count is None else int(count)
        out = np.empty(count, self.dtype)
        self.ctx._ck(self.ctx.lib.tsdrgpu_download(self.ctx.h, out.ctypes.data, self.at(elem_offset), out.nbytes))
        self.ctx.sync()
        return out

    def zero(self):
        self.ctx._ck(self.ctx.lib.tsdrgpu_zero(self.ctx.h, self.ptr, self.nbytes))

    def free(self):
        if self.ptr and self.ctx.h:
            self.ctx.lib.tsdrgpu_free(self.ctx.h, self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class TsdrGpu:
    def __init__(self, device=0):
        self.lib = load_library()
        h = vp()
        rc = self.lib.tsdrgpu_create(C.byref(h), device)
        if rc != 0:
            raise TsdrGpuError(f"tsdrgpu_create({device}) failed with {rc}: no usable HIP device "
                               "(this package has no CPU path)")
        self.h = h
        self._children = []  # weakrefs to objects that hold this context inside the C library

    def _adopt(self, child):
        import weakref
        if len(self._children) > 4096:
            self._children = [r for r in self._children if r() is not None]
        self._children.append(weakref.ref(child))

    def profile_begin(self):
        self._ck(self.lib.tsdrgpu_profile_begin(self.h))

    def profile_end(self):
        """{kernel name: (total_ms, launches)} for everything queued since profile_begin()."""
        ent = (ProfileEntry * 32)()
        n = C.c_int()
        self._ck(self.lib.tsdrgpu_profile_end(self.h, ent, 32, C.byref(n)))
        return {ent[i].name.decode(): (ent[i].total_ms, ent[i].launches) for i in range(n.value)}

    def close(self):
        """Destroys dependants first: the C objects keep a pointer to the context."""
        if getattr(self, "h", None):
            for ref in self._children:
                child = ref()
                if child is not None:
                    child.destroy()
            self._children = []
            self.lib.tsdrgpu_destroy(self.h)
            self.h = None

    def _ck(self, rc):
        if rc != 0:
            raise TsdrGpuError(f"tsdrgpu error {rc}: {self.lib.tsdrgpu_last_error(self.h).decode()}")

    def sync(self):
        self._ck(self.lib.tsdrgpu_sync(self.h))

    def device_name(self):
        buf = C.create_string_buffer(256)
        self._ck(self.lib.tsdrgpu_device_name(self.h, buf, 256))
        return buf.value.decode()

    def stream(self):
        return self.lib.tsdrgpu_stream(self.h)

    def empty(self, count, dtype=np.float32):
        return DeviceArray(self, count, dtype)

    def to_device(self, host, dtype=None):
        host = np.ascontiguousarray(host, dtype or host.dtype)
        return DeviceArray(self, host.size, host.dtype).upload(host)

    def timer_start(self):
        self._ck(self.lib.tsdrgpu_timer_start(self.h))

    def timer_stop_ms(self):
        ms = C.c_float()
        self._ck(self.lib.tsdrgpu_timer_stop_ms(self.h, C.byref(ms)))
        return ms.value

    # ---- a1 ----------------------------------------------------------------
    def am_demod(self, d_iq, d_out, nsamples, iq_offset=0, out_offset=0):
        self._ck(self.lib.tsdrgpu_am_demod(self.h, d_iq.at(iq_offset), d_out.at(out_offset), nsamples))

    # ---- §8(f) ---------------------------------------------------------------
    SAMPLE_TYPES = {"float": 0, "int8": 1, "int16": 2, "uint8": 3, "uint16": 4}

    def decode_samples(self, d_raw, sample_type, d_out, n):
        self._ck(self.lib.tsdrgpu_decode_samples(self.h, d_raw.ptr, self.SAMPLE_TYPES[sample_type], d_out.ptr, n))

    def plot_columns(self, d_ptr, size, nwidth, scale=None):
        """PlotVisualizer.populateData on a device plot: (visdata[nwidth], lowest, highest, max_index)."""
        vis = np.empty(nwidth, np.float64)
        lo, hi, mi = C.c_double(), C.c_double(), C.c_int()
        self._ck(self.lib.tsdrgpu_plot_columns(self.h, d_ptr, size, nwidth, C.byref(scale) if scale is not None else None,
                                               vis.ctypes.data, C.byref(lo), C.byref(hi), C.byref(mi)))
        return vis, lo.value, hi.value, mi.value

    def frame_to_rgb(self, d_frame, d_rgb, npixels, inverted=False, frame_offset=0):
        self._ck(self.lib.tsdrgpu_frame_to_rgb(self.h, d_frame.at(frame_offset), d_rgb.ptr, npixels, int(inverted)))

    # ---- a13/a14 -------------------------------------------------------------
    def fft_perform(self, d_iq, n, inverse, offset=0, exact=False):
        fn = self.lib.tsdrgpu_fft_exact if exact else self.lib.tsdrgpu_fft
        self._ck(fn(self.h, d_iq.at(offset), n, int(inverse)))

    def superb_set_plan(self, trips):
        """3: the three-trip plan where it applies (default); 0: the pass-per-radix plan always"""
        self._ck(self.lib.tsdrgpu_superb_set_plan(self.h, int(trips)))

    def superb_stitch(self, d_hops, gathered, samples_in_frame, d_out, exact=False):
        ptrs = (vp * len(d_hops))(*[h.ptr for h in d_hops])
        offs = (C.c_int32 * len(d_hops))()
        total = C.c_uint32()
        fn = self.lib.tsdrgpu_superb_stitch_exact if exact else self.lib.tsdrgpu_superb_stitch
        self._ck(fn(self.h, ptrs, len(d_hops), gathered, samples_in_frame,
                                                d_out.ptr, offs, C.byref(total)))
        return np.array(list(offs), np.int32), total.value


class Resampler:
    """dsp_resample_t + dsp_resample_process on the device (dsp.c:250-307)."""

    def __init__(self, ctx):
        self.ctx = ctx
        h = vp()
        ctx._ck(ctx.lib.tsdrgpu_resampler_create(ctx.h, C.byref(h)))
        self.h = h
        ctx._adopt(self)

    def destroy(self):
        if getattr(self, "h", None) and self.ctx.h:
            self.ctx.lib.tsdrgpu_resampler_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass

    def reset(self):
        self.ctx._ck(self.ctx.lib.tsdrgpu_resampler_reset(self.h))

    def set_state(self, contrib, offset):
        self.ctx._ck(self.ctx.lib.tsdrgpu_resampler_setstate(self.h, contrib, offset))

    def state(self):
        c, o = C.c_double(), C.c_double()
        self.ctx._ck(self.ctx.lib.tsdrgpu_resampler_getstate(self.h, C.byref(c), C.byref(o)))
        return c.value, o.value

    def track_frames(self, frame_pixels, phase=0):
        """Per-frame min/max of the emitted pixels from now on (0 = off)."""
        self.ctx._ck(self.ctx.lib.tsdrgpu_resampler_track_frames(self.h, int(frame_pixels), int(phase)))

    def frame_minmax(self, download=True):
        """(min[], max[]) of the frames the last process() completed; device pointers with download=False."""
        a, b, n = vp(), vp(), C.c_int()
        self.ctx._ck(self.ctx.lib.tsdrgpu_resampler_frame_minmax(self.h, C.byref(a), C.byref(b), C.byref(n)))
        if not download:
            return a.value, b.value, n.value
        mn = np.empty(n.value, np.float32)
        mx = np.empty(n.value, np.float32)
        if n.value:
            self.ctx._ck(self.ctx.lib.tsdrgpu_download(self.ctx.h, mn.ctypes.data, a.value, mn.nbytes))
            self.ctx._ck(self.ctx.lib.tsdrgpu_download(self.ctx.h, mx.ctypes.data, b.value, mx.nbytes))
            self.ctx.sync()
        return mn, mx

    def count(self, chunk, nchunks, up, down):
        return self.ctx.lib.tsdrgpu_resample_count(self.h, chunk, nchunks, up, down)

    def process_band(self, d_in, in_is_iq, chunk, nchunks, up, down, width, height, y0, rows, phase, d_band, capacity_frames, in_offset=0):
        """tsdrgpu_resample_band: only rows [y0, y0+rows) of every frame; returns (pixels of the full call, frames touched)."""
        n, ft = C.c_int64(), C.c_int()
        self.ctx._ck(self.ctx.lib.tsdrgpu_resample_band(self.h, d_in.at(in_offset), int(in_is_iq), chunk, nchunks, up, down, width, height, y0, rows,
                                                        phase, d_band.at(0), capacity_frames, C.byref(n), C.byref(ft)))
        return n.value, ft.value

    def process(self, d_in, in_is_iq, chunk, nchunks, up, down, nearest, d_out, in_offset=0, out_offset=0):
        n = C.c_int64()
        self.ctx._ck(self.ctx.lib.tsdrgpu_resample(self.h, d_in.at(in_offset), int(in_is_iq), chunk, nchunks, up,
                                                   down, int(nearest), d_out.at(out_offset),
                                                   d_out.count - out_offset, C.byref(n)))
        return n.value


class PostProcess:
    """dsp_postprocess_t + dsp_post_process on the device (dsp.c:112-239)."""

    def __init__(self, ctx):
        self.ctx = ctx
        h = vp()
        ctx._ck(ctx.lib.tsdrgpu_postproc_create(ctx.h, C.byref(h)))
        self.h = h
        ctx._adopt(self)

    def destroy(self):
        if getattr(self, "h", None) and self.ctx.h:
            self.ctx.lib.tsdrgpu_postproc_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass

    def reset(self):
        self.ctx._ck(self.ctx.lib.tsdrgpu_postproc_reset(self.h))

    def redo_stats(self):
        """(toss-up decisions, strips re-collapsed because of them, strips flagged up front) of the last run"""
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        self.ctx._ck(self.ctx.lib.tsdrgpu_postproc_redo_stats(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def redo_raw(self, max_frames=4096):
        """(flagged, tossup, recollapsed) arrays of shape (F, 2) for the last run"""
        h = np.zeros(6 * max_frames, np.int32)
        f = C.c_int()
        self.ctx._ck(self.ctx.lib.tsdrgpu_postproc_redo_raw(self.h, h.ctypes.data, h.size, C.byref(f)))
        F = f.value
        return tuple(h[k * 2 * F:(k + 1) * 2 * F].reshape(F, 2) for k in range(3))

    def set_snr(self, on=True):
        """dsp_autogain_t.snr per frame as a by-product of every run from now on (dsp.c:69-93)."""
        self.ctx._ck(self.ctx.lib.tsdrgpu_postproc_set_snr(self.h, int(on)))

    def snr(self, nframes):
        out = (C.c_float * max(nframes, 1))()
        self.ctx._ck(self.ctx.lib.tsdrgpu_postproc_snr(self.h, out, nframes))
        return np.array(out[:nframes], np.float32)

    def set_exact_ties(self, on=True):
        """Detect sync-detector decisions that are toss-ups at the precision of the strips and redo them exactly."""
        self.ctx._ck(self.ctx.lib.tsdrgpu_postproc_set_exact_ties(self.h, int(on)))

    def run(self, d_frames, nframes, width, height, d_out, motionblur=0.0, lowpasscoeff=0.1,
            lowpass_before_sync=0, autogain_after_proc=0, autoshift=0, pll=0, superres=0,
            want_info=True, frames_offset=0, out_offset=0):
        prm = PPParams(lowpass_before_sync, autogain_after_proc, autoshift, pll, superres, motionblur, lowpasscoeff)
        info = (PPFrameInfo * nframes)() if want_info else None
        self.ctx._ck(self.ctx.lib.tsdrgpu_postproc_run(self.h, d_frames.at(frames_offset), nframes, width, height,
                                                       C.byref(prm), d_out.at(out_offset), info))
        return list(info) if want_info else None

    def begin(self, d_frames, nframes, width, height, motionblur=0.0, lowpasscoeff=0.1, lowpass_before_sync=0,
              autogain_after_proc=0, autoshift=0, pll=0, superres=0, frames_offset=0):
        """First half of run(): statistics + the frame-to-frame chain (on the side stream)."""
        prm = PPParams(lowpass_before_sync, autogain_after_proc, autoshift, pll, superres, motionblur, lowpasscoeff)
        self._nframes = nframes
        self.ctx._ck(self.ctx.lib.tsdrgpu_postproc_begin(self.h, d_frames.at(frames_offset), nframes, width, height,
                                                         C.byref(prm)))

    def begin_minmax(self, d_frames, nframes, width, height, fmin_ptr, fmax_ptr, d_out, motionblur=0.0, lowpasscoeff=0.1,
                     lowpass_before_sync=0, autogain_after_proc=0, autoshift=0, pll=0, superres=0, frames_offset=0,
                     out_offset=0):
        """Fused first half: per-frame min/max supplied (device pointers), one trip over the raw frames."""
        prm = PPParams(lowpass_before_sync, autogain_after_proc, autoshift, pll, superres, motionblur, lowpasscoeff)
        self._nframes = nframes
        self.ctx._ck(self.ctx.lib.tsdrgpu_postproc_begin_minmax(self.h, d_frames.at(frames_offset), nframes, width, height,
                                                                C.byref(prm), fmin_ptr, fmax_ptr, d_out.at(out_offset)))

    def finish(self, d_out, want_info=True, out_offset=0):
        """Second half of run(): the normalise / low-pass pass once the chain is done."""
        info = (PPFrameInfo * self._nframes)() if want_info else None
        self.ctx._ck(self.ctx.lib.tsdrgpu_postproc_finish(self.h, d_out.at(out_offset), info))
        return list(info) if want_info else None

    def band_begin(self, d_band, nframes, width, height, y0, rows, motionblur=0.0, lowpasscoeff=0.1, superres=0, frames_offset=0):
        """Row-band sharding, first half: statistics of rows [y0, y0+rows); returns (xsum_ptr, n_doubles, xmax_ptr,
        n_floats): device buffers the caller all-reduces in place (sum / max) across the ranks."""
        prm = PPParams(0, 0, 0, 0, superres, motionblur, lowpasscoeff)
        self._nframes = nframes
        ps, ns, pm, nm = vp(), C.c_int64(), vp(), C.c_int64()
        self.ctx._ck(self.ctx.lib.tsdrgpu_postproc_band_begin(self.h, d_band.at(frames_offset), nframes, width, height, y0, rows, C.byref(prm),
                                                              C.byref(ps), C.byref(ns), C.byref(pm), C.byref(nm)))
        return ps.value, ns.value, pm.value, nm.value

    def band_begin_minmax(self, d_band, nframes, width, height, y0, rows, fmin_ptr, fmax_ptr, motionblur=0.0, lowpasscoeff=0.1, superres=0,
                          frames_offset=0):
        """Fused band run, first step (tsdrgpu_postproc_band_begin_minmax): the band's share of every frame's range (device pointers from
        Resampler.frame_minmax(download=False)) -> (xmax_ptr, n_floats), which the caller max-all-reduces; then band_fused()."""
        prm = PPParams(0, 0, 0, 0, superres, motionblur, lowpasscoeff)
        self._nframes = nframes
        pm, nm = vp(), C.c_int64()
        self.ctx._ck(self.ctx.lib.tsdrgpu_postproc_band_begin_minmax(self.h, d_band.at(frames_offset), nframes, width, height, y0, rows, C.byref(prm),
                                                                     fmin_ptr, fmax_ptr, C.byref(pm), C.byref(nm)))
        return pm.value, nm.value

    def band_fused(self, d_out_band, out_offset=0):
        """Fused band run, the trip (tsdrgpu_postproc_band_fused): returns (xsum_ptr, n_doubles) for the sum all-reduce; then band_advance()."""
        ps, ns = vp(), C.c_int64()
        self.ctx._ck(self.ctx.lib.tsdrgpu_postproc_band_fused(self.h, d_out_band.at(out_offset), C.byref(ps), C.byref(ns)))
        return ps.value, ns.value

    def band_finish(self, d_out_band, want_info=True, out_offset=0):
        info = (PPFrameInfo * self._nframes)() if want_info else None
        self.ctx._ck(self.ctx.lib.tsdrgpu_postproc_band_finish(self.h, d_out_band.at(out_offset), info))
        return list(info) if want_info else None

    def band_advance(self, d_out_band, band_index, nbands, want_info=True, out_offset=0):
        """Contract-exact second half of a band run, one step: returns (more, buf_ptr, n_doubles, infos).  While `more`,
        the caller sum-all-reduces buf in place over the ranks and calls again (tsdrgpu_postproc_band_advance)."""
        info = (PPFrameInfo * self._nframes)() if want_info else None
        buf, n, more = vp(), C.c_int64(), C.c_int()
        self.ctx._ck(self.ctx.lib.tsdrgpu_postproc_band_advance(self.h, d_out_band.at(out_offset), band_index, nbands, C.byref(buf), C.byref(n),
                                                                C.byref(more), info))
        return more.value, buf.value, n.value, (list(info) if (want_info and not more.value) else None)

    def band_spec_stats(self):
        """(band runs speculated, of those replayed literally) since the object was created"""
        runs, replays = C.c_uint64(), C.c_uint64()
        self.ctx._ck(self.ctx.lib.tsdrgpu_postproc_band_spec_stats(self.h, C.byref(runs), C.byref(replays)))
        return runs.value, replays.value

    def band_open(self, d_band, nframes, width, height, edges, band_index, motionblur=0.0, lowpasscoeff=0.1, lowpass_before_sync=0,
                  autogain_after_proc=0, autoshift=0, pll=0, superres=0, frames_offset=0):
        """General band run (every stage order, autoshift, PLL): tsdrgpu_postproc_band_open; then band_step() until BAND_DONE."""
        prm = PPParams(lowpass_before_sync, autogain_after_proc, autoshift, pll, superres, motionblur, lowpasscoeff)
        self._nframes = nframes
        e = (C.c_int * len(edges))(*[int(v) for v in edges])
        self.ctx._ck(self.ctx.lib.tsdrgpu_postproc_band_open(self.h, d_band.at(frames_offset), nframes, width, height, e, len(edges) - 1, band_index,
                                                             C.byref(prm)))

    def band_step(self, d_out_band, want_info=True, out_offset=0):
        """one step: returns (kind, buf_ptr, count, infos) — kind says which collective the caller makes on buf (gpu.BAND_*)"""
        info = (PPFrameInfo * self._nframes)() if want_info else None
        x = BandExchange()
        self.ctx._ck(self.ctx.lib.tsdrgpu_postproc_band_step(self.h, d_out_band.at(out_offset), C.byref(x), info))
        return x.kind, x.d_buf, x.count, (list(info) if (want_info and x.kind == BAND_DONE) else None)

    def strips(self, width, height):
        c = np.empty(width, np.float32)
        r = np.empty(height, np.float32)
        self.ctx._ck(self.ctx.lib.tsdrgpu_postproc_strips(self.h, c.ctypes.data, r.ctypes.data))
        return c, r


class Comm:
    """RCCL communicator for the sharded sweep (tsdrgpu_comm_*): one per rank, on the context's device."""

    @staticmethod
    def unique_id(ctx):
        buf = (C.c_char * 128)()
        ctx._ck(ctx.lib.tsdrgpu_rccl_unique_id(buf))
        return bytes(buf)

    def __init__(self, ctx, world, rank, id128):
        self.ctx = ctx
        h = vp()
        ctx._ck(ctx.lib.tsdrgpu_comm_create(ctx.h, C.byref(h), int(world), int(rank), C.c_char_p(id128)))
        self.h = h
        self.world, self.rank = world, rank

    def count(self):
        """(ranks, my rank) as RCCL itself reports them: ncclCommCount / ncclCommUserRank"""
        n, me = C.c_int(), C.c_int()
        self.ctx._ck(self.ctx.lib.tsdrgpu_comm_count(self.h, C.byref(n), C.byref(me)))
        return n.value, me.value

    def allreduce_f32max(self, d_ptr, count, lane=0):
        self.ctx._ck(self.ctx.lib.tsdrgpu_comm_allreduce_f32max(self.h, d_ptr, int(count), int(lane)))

    def broadcast_f32(self, d_ptr, count, root, lane=0):
        self.ctx._ck(self.ctx.lib.tsdrgpu_comm_broadcast_f32(self.h, d_ptr, int(count), int(root), int(lane)))

    def allgather_f32(self, d_ptr, count_per_rank, lane=0):
        self.ctx._ck(self.ctx.lib.tsdrgpu_comm_allgather_f32(self.h, d_ptr, int(count_per_rank), int(lane)))

    def allreduce_f64(self, d_ptr, count, lane=0):
        self.ctx._ck(self.ctx.lib.tsdrgpu_comm_allreduce_f64(self.h, d_ptr, int(count), int(lane)))

    def destroy(self):
        if getattr(self, "h", None):
            self.ctx.lib.tsdrgpu_comm_destroy(self.h)
        self.h = None


class SuperbShard:
    """One hop of the super-bandwidth stitch on this GPU (tsdrgpu_superb_shard_*): reference() -> broadcast ->
    spectrum() -> all-gather -> finish()."""

    def __init__(self, ctx, nhops, my_hop, gathered, samples_in_frame):
        self.ctx = ctx
        h = vp()
        ctx._ck(ctx.lib.tsdrgpu_superb_shard_create(ctx.h, C.byref(h), nhops, my_hop, gathered, samples_in_frame))
        self.h = h

    def reference(self, d_hop):
        p, n = vp(), C.c_int64()
        self.ctx._ck(self.ctx.lib.tsdrgpu_superb_shard_reference(self.h, d_hop.at(0), C.byref(p), C.byref(n)))
        return p.value, n.value

    def spectrum(self, d_hop):
        p, n, off = vp(), C.c_int64(), C.c_int32()
        self.ctx._ck(self.ctx.lib.tsdrgpu_superb_shard_spectrum(self.h, d_hop.at(0), C.byref(p), C.byref(n), C.byref(off)))
        return p.value, n.value, off.value

    def finish(self, d_out):
        t = C.c_uint32()
        self.ctx._ck(self.ctx.lib.tsdrgpu_superb_shard_finish(self.h, d_out.at(0), C.byref(t)))
        return t.value

    def destroy(self):
        if getattr(self, "h", None) and self.ctx.h:
            self.ctx.lib.tsdrgpu_superb_shard_destroy(self.h)
        self.h = None


class Autocorr:
    """frameratedetector numerics on the device (frameratedetector.c:26-126)."""

    def __init__(self, ctx, samplerate):
        self.ctx = ctx
        h = vp()
        ctx._ck(ctx.lib.tsdrgpu_autocorr_create(ctx.h, C.byref(h), int(samplerate)))
        self.h = h
        ctx._adopt(self)
        a = [C.c_int32() for _ in range(4)]
        b = [C.c_uint32() for _ in range(2)]
        ctx._ck(ctx.lib.tsdrgpu_autocorr_geometry(h, *[C.byref(x) for x in a + b]))
        self.flo, self.flen, self.llo, self.llen = [x.value for x in a]
        self.capture, self.n = [x.value for x in b]

    def destroy(self):
        if getattr(self, "h", None) and self.ctx.h:
            self.ctx.lib.tsdrgpu_autocorr_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass

    def reset(self):
        self.ctx._ck(self.ctx.lib.tsdrgpu_autocorr_reset(self.h))

    def set_async(self, on=True):
        """queue this object's work on the context's side stream (overlaps the frame path)"""
        self.ctx._ck(self.ctx.lib.tsdrgpu_autocorr_set_async(self.h, int(on)))

    def run(self, d_in, in_is_iq, stride, nwindows, mode=0, in_offset=0):
        self.ctx._ck(self.ctx.lib.tsdrgpu_autocorr_run(self.h, d_in.at(in_offset), int(in_is_iq), stride, nwindows, mode))

    def plots(self):
        f = np.empty(self.flen, np.float64)
        l = np.empty(self.llen, np.float64)
        calls = C.c_uint64()
        self.ctx._ck(self.ctx.lib.tsdrgpu_autocorr_plots(self.h, f.ctypes.data, l.ctypes.data, C.byref(calls)))
        return f, l, calls.value

    def device_plots(self):
        p, n = vp(), C.c_int64()
        self.ctx._ck(self.ctx.lib.tsdrgpu_autocorr_device_plots(self.h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def device_sums(self):
        """what a caller's own collective must sum over the ranks: the lags + the lag-0 scale of the certificate"""
        p, n = vp(), C.c_int64()
        self.ctx._ck(self.ctx.lib.tsdrgpu_autocorr_device_sums(self.h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def allreduce(self, comm, total_windows):
        """per-lag sums of every rank -> global plots on every rank (RCCL from C, on this object's lane)"""
        self.ctx._ck(self.ctx.lib.tsdrgpu_autocorr_allreduce(self.h, comm.h, int(total_windows)))

    def finalize_sums(self, total_windows):
        self.ctx._ck(self.ctx.lib.tsdrgpu_autocorr_finalize_sums(self.h, total_windows))

    def argmax(self):
        a, b = C.c_int32(), C.c_int32()
        self.ctx._ck(self.ctx.lib.tsdrgpu_autocorr_argmax(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def set_exact(self, on=True):
        """The reference's own FFT arithmetic: plots / argmax / last_corr bit-identical (slower)."""
        self.ctx._ck(self.ctx.lib.tsdrgpu_autocorr_set_exact(self.h, int(on)))

    def set_certify(self, mode=1, retain_bytes=0):
        """Certified mode (tsdrgpu_autocorr_set_certify): 1 = the library retains the windows, 2 = the caller does."""
        self.ctx._ck(self.ctx.lib.tsdrgpu_autocorr_set_certify(self.h, int(mode), int(retain_bytes)))

    def promote(self):
        self.ctx._ck(self.ctx.lib.tsdrgpu_autocorr_promote(self.h))

    def promote_step(self, max_windows):
        """incremental promotion: replays up to max_windows windows; returns the windows still to go"""
        r = C.c_int()
        self.ctx._ck(self.ctx.lib.tsdrgpu_autocorr_promote_step(self.h, int(max_windows), C.byref(r)))
        return r.value

    def retention_reserve(self, windows, wait_ms=0):
        self.ctx._ck(self.ctx.lib.tsdrgpu_autocorr_retention_reserve(self.h, int(windows), int(wait_ms)))

    def retention(self):
        """(ring capacity in windows, the part of it allocated so far, position of the epoch's next window, epoch runs in the exact form)"""
        a, r, b, c = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        self.ctx._ck(self.ctx.lib.tsdrgpu_autocorr_retention(self.h, C.byref(a), C.byref(r), C.byref(b), C.byref(c)))
        return a.value, r.value, b.value, bool(c.value)

    def certificate(self):
        c = AcCertificate()
        self.ctx._ck(self.ctx.lib.tsdrgpu_autocorr_certificate(self.h, C.byref(c)))
        return c

    def argmax_certified(self):
        """(frame_idx, line_idx, promoted): the argmax pair that is provably the reference's."""
        a, b, p = C.c_int32(), C.c_int32(), C.c_int()
        self.ctx._ck(self.ctx.lib.tsdrgpu_autocorr_argmax_certified(self.h, C.byref(a), C.byref(b), C.byref(p)))
        return a.value, b.value, p.value

    def set_plan(self, trips):
        """3: three-trip (four-step) transform plan where it applies (default); 5: the round-1 Stockham plan."""
        self.ctx._ck(self.ctx.lib.tsdrgpu_autocorr_set_plan(self.h, int(trips)))

    def argmax_async(self):
        self.ctx._ck(self.ctx.lib.tsdrgpu_autocorr_argmax_async(self.h))

    def argmax_result(self):
        a, b = C.c_int32(), C.c_int32()
        self.ctx._ck(self.ctx.lib.tsdrgpu_autocorr_argmax_result(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def last_corr(self):
        p, n = vp(), C.c_uint32()
        self.ctx._ck(self.ctx.lib.tsdrgpu_autocorr_last_corr(self.h, C.byref(p), C.byref(n)))
        out = np.empty(2 * n.value, np.float32)
        self.ctx._ck(self.ctx.lib.tsdrgpu_download(self.ctx.h, out.ctypes.data, p.value, out.nbytes))
        self.ctx.sync()
        return out


class ModeDetect:
    """The GUI's auto-resolution logic (Main.onIncommingPlot) as a library object; CPU only."""

    def __init__(self):
        self.lib = load_library()
        h = vp()
        if self.lib.tsdrgpu_modedetect_create(C.byref(h)) != 0:
            raise TsdrGpuError("tsdrgpu_modedetect_create failed")
        self.h = h

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.tsdrgpu_modedetect_destroy(self.h)
            self.h = None

    def reset(self):
        self.lib.tsdrgpu_modedetect_reset(self.h)

    def feed(self, frame_offset, frame_idx, line_offset, line_idx, samplerate):
        d = Detection()
        rc = self.lib.tsdrgpu_modedetect_feed(self.h, frame_offset, frame_idx, line_offset, line_idx, samplerate, C.byref(d))
        if rc != 0:
            raise TsdrGpuError(f"tsdrgpu_modedetect_feed failed with {rc}")
        return d
