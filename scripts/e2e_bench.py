#!/usr/bin/env python3
"""Whole-library, PCIe-inclusive measurement through the tsdr_* API.

The reference's RawFile plugin rebuilt with PERFORMANCE_BENCHMARK=1 (oracle/_ref, free-running, looping)
feeds (a) this repository's libTSDRLibrary.so (MI355X) and (b) the reference's own library compiled from its
sources (oracle/_ref/libtsdr_ref.so exports the same tsdr_* API) on the host cores of the same box.  Counts
frames handed to the frame callback per wall second; effective MS/s = frames x samples-per-frame (both
pipelines are lossy by design: they drop whole frames/blocks when they cannot keep up).

usage: e2e_bench.py [--fs 100000000 --height 1125 --seconds 4]
"""
import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import host_util as hu  # noqa: E402
from tempestsdr_amd import synth  # noqa: E402

MODES = {525: "640x480", 806: "1024x768", 1125: "1920x1080", 2250: "3840x2160"}


def run(libpath, plugin, params, height, fv, seconds):
    lib = C.CDLL(libpath)
    vp = C.c_void_p
    lib.tsdr_init.argtypes = [C.POINTER(vp), hu.VALUE_CB, hu.PLOT_CB, vp]
    lib.tsdr_init.restype = None
    lib.tsdr_loadplugin.argtypes = [vp, C.c_char_p, C.c_char_p]
    lib.tsdr_setresolution.argtypes = [vp, C.c_int, C.c_double]
    lib.tsdr_readasync.argtypes = [vp, hu.FRAME_CB, vp]
    lib.tsdr_stop.argtypes = [vp]
    lib.tsdr_motionblur.argtypes = [vp, C.c_float]
    lib.tsdr_setgain.argtypes = [vp, C.c_float]
    lib.tsdr_getlasterrortext.argtypes = [vp]
    lib.tsdr_getlasterrortext.restype = C.c_char_p
    cnt = {"frames": 0, "plots": 0, "w": 0, "h": 0}

    def on_frame(buf, w, h, ctx):
        cnt["frames"] += 1
        cnt["w"], cnt["h"] = w, h

    def on_plot(pid, off, vals, size, rate, ctx):
        cnt["plots"] += 1

    cbs = (hu.FRAME_CB(on_frame), hu.VALUE_CB(lambda *a: None), hu.PLOT_CB(on_plot))
    h = vp()
    lib.tsdr_init(C.byref(h), cbs[1], cbs[2], None)
    pbuf = C.create_string_buffer(params.encode())
    rc = lib.tsdr_loadplugin(h, plugin.encode(), pbuf)
    assert rc == 0, (rc, lib.tsdr_getlasterrortext(h))
    lib.tsdr_setgain(h, 0.5)
    lib.tsdr_motionblur(h, 0.0)
    assert lib.tsdr_setresolution(h, height, fv) == 0
    status = {}
    th = threading.Thread(target=lambda: status.setdefault("rc", lib.tsdr_readasync(h, cbs[0], None)))
    th.start()
    time.sleep(1.0)  # warm-up (device context, buffers)
    f0, p0, t0 = cnt["frames"], cnt["plots"], time.time()
    time.sleep(seconds)
    f1, p1, t1 = cnt["frames"], cnt["plots"], time.time()
    lib.tsdr_stop(h)
    th.join(30)
    return {"frames_per_s": (f1 - f0) / (t1 - t0), "plots_per_s": (p1 - p0) / (t1 - t0), "width": cnt["w"], "height": cnt["h"],
            "status": status.get("rc")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fs", type=int, default=100_000_000)
    ap.add_argument("--height", type=int, default=1125)
    ap.add_argument("--fv", type=float, default=60.0)
    ap.add_argument("--seconds", type=float, default=4.0)
    ap.add_argument("--file-seconds", type=float, default=0.5)
    args = ap.parse_args()
    plugin = os.path.join(ROOT, "oracle", "_ref", "libTSDRPlugin_RawFile_bench.so")
    reflib = os.path.join(ROOT, "oracle", "_ref", "libtsdr_ref.so")
    path = "/tmp/e2e_iq.f32"
    n = int(args.file_seconds * args.fs)
    with open(path, "wb") as f:
        step = 1 << 22
        for s in range(0, n, step):
            synth.synth_iq(args.fs, MODES[args.height], args.fv, min(step, n - s), start=s).tofile(f)
    params = f"{path} {args.fs} float"
    S = args.fs / args.fv
    out = {"config": f"{args.fs/1e6:g} MS/s float32 IQ file, h={args.height}, fv={args.fv}, RawFile plugin free-running (PERFORMANCE_BENCHMARK=1)",
           "host_cores": os.cpu_count()}
    ours = run(hu.LIB, plugin, params, args.height, args.fv, args.seconds)
    ours["effective_Msps"] = ours["frames_per_s"] * S / 1e6
    out["mi355x_libTSDRLibrary"] = ours
    if os.path.exists(reflib):
        ref = run(reflib, plugin, params, args.height, args.fv, args.seconds)
        ref["effective_Msps"] = ref["frames_per_s"] * S / 1e6
        out["reference_cpu_libTSDRLibrary"] = ref
    print(json.dumps(out))


if __name__ == "__main__":
    main()
