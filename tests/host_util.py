"""ctypes driver of libTSDRLibrary.so (the tsdr_* drop-in) for the host tests.
Replays the call sequence the Java GUI issues through its JNI shim
(SURVEY.md §3.6)."""
import ctypes as C
import os
import subprocess
import threading
import time

import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "tempestsdr_amd", "libTSDRLibrary.so")
PLUGIN_SRC = os.path.join(ROOT, "tests", "plugins", "tsdr_test_plugin.c")
PLUGIN = os.path.join(ROOT, "tests", "plugins", "libtsdr_test_plugin.so")
MEM_PLUGIN = os.path.join(ROOT, "tempestsdr_amd", "libTSDRPlugin_Mem.so")

from tempestsdr_amd.tsdrlib import FRAME_CB, VALUE_CB, PLOT_CB, TSDR_SYMBOLS, load as _load  # noqa: E402,F401

RGB_CB = C.CFUNCTYPE(None, C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_void_p)


class Stats(C.Structure):
    """tsdrx_stats_t (include/TSDRLibraryExt.h)"""
    _fields_ = [(n, C.c_int64) for n in ("blocks_in", "blocks_lost", "frames_made", "frames_lost_to_viewer", "windows",
                                          "plots_held", "epochs_replayed", "frames_fused")]


def build_test_plugin():
    if not os.path.exists(PLUGIN) or os.path.getmtime(PLUGIN) < os.path.getmtime(PLUGIN_SRC):
        subprocess.run(["gcc", "-O2", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "include"), "-o", PLUGIN, PLUGIN_SRC],
                       check=True)
    return PLUGIN


def load():
    return _load(LIB)


class Session:
    """One tsdr_lib_t plus the three callbacks, collecting what they deliver."""

    def __init__(self):
        self.lib = load()
        self.frames, self.plots, self.values = [], [], []
        self.lock = threading.Lock()

        def on_frame(buf, w, h, ctx):
            a = np.ctypeslib.as_array(buf, shape=(w * h,)).copy()
            with self.lock:
                self.frames.append((w, h, a))

        def on_value(vid, a0, a1, ctx):
            with self.lock:
                self.values.append((vid, a0, a1))

        def on_plot(pid, offset, vals, size, rate, ctx):
            a = np.ctypeslib.as_array(vals, shape=(size,)).copy()
            with self.lock:
                self.plots.append((pid, offset, a, rate))

        def on_rgb(buf, w, h, ctx):
            a = np.ctypeslib.as_array(buf, shape=(w * h,)).copy()
            with self.lock:
                self.frames.append((w, h, a))

        self._rgb_cb = RGB_CB(on_rgb)
        self._cbs = (FRAME_CB(on_frame), VALUE_CB(on_value), PLOT_CB(on_plot))
        self.h = C.c_void_p()
        self.lib.tsdr_init(C.byref(self.h), self._cbs[1], self._cbs[2], None)
        self.thread = None
        self.status = None

    def err(self):
        e = self.lib.tsdr_getlasterrortext(self.h)
        return e.decode() if e else None

    def start(self, rgb=None):
        """rgb: None = float frames through tsdr_readasync; 0 / 1 = packed RGB through tsdrx_readasync_rgb (inverted = rgb)"""
        def run():
            if rgb is None:
                self.status = self.lib.tsdr_readasync(self.h, self._cbs[0], None)
            else:
                self.lib.tsdrx_readasync_rgb.argtypes = [C.c_void_p, RGB_CB, C.c_void_p, C.c_int]
                self.status = self.lib.tsdrx_readasync_rgb(self.h, self._rgb_cb, None, int(rgb))
        self.thread = threading.Thread(target=run)
        self.thread.start()

    def wait_frames(self, n, timeout=30.0):
        t0 = time.time()
        while time.time() - t0 < timeout:
            with self.lock:
                if len(self.frames) >= n:
                    return True
            if self.thread is not None and not self.thread.is_alive():
                return False
            time.sleep(0.01)
        return False

    def stats(self):
        st = Stats()
        self.lib.tsdrx_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
        assert self.lib.tsdrx_get_stats(self.h, C.byref(st)) == 0
        return st

    def stop(self):
        rc = self.lib.tsdr_stop(self.h)
        if self.thread is not None:
            self.thread.join(30)
        return rc

    def close(self):
        self.lib.tsdr_free(C.byref(self.h))
