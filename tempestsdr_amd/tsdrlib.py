"""ctypes view of the tsdr_* host API (include/TSDRLibrary.h) as libTSDRLibrary.so exports it — the calls the
Java GUI issues through its JNI shim — plus a whole-library throughput run used by bench.py
and scripts/e2e_bench.py.  Host-side plumbing only: every sample goes through the C library."""
import ctypes as C
import os
import threading
import time

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libTSDRLibrary.so")
MEM_PLUGIN = os.path.join(HERE, "libTSDRPlugin_Mem.so")

FRAME_CB = C.CFUNCTYPE(None, C.POINTER(C.c_float), C.c_int, C.c_int, C.c_void_p)
VALUE_CB = C.CFUNCTYPE(None, C.c_int, C.c_double, C.c_double, C.c_void_p)
PLOT_CB = C.CFUNCTYPE(None, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_int, C.c_uint32, C.c_void_p)

TSDR_SYMBOLS = ["tsdr_free", "tsdr_getctx", "tsdr_getlasterrortext", "tsdr_getsamplerate", "tsdr_init",
                "tsdr_isrunning", "tsdr_loadplugin", "tsdr_motionblur", "tsdr_readasync", "tsdr_reset",
                "tsdr_setbasefreq", "tsdr_setgain", "tsdr_setparameter_double", "tsdr_setparameter_int",
                "tsdr_setresolution", "tsdr_stop", "tsdr_sync", "tsdr_unloadplugin"]


def load(path=LIB):
    lib = C.CDLL(path)
    vp = C.c_void_p
    lib.tsdr_init.argtypes = [C.POINTER(vp), VALUE_CB, PLOT_CB, vp]
    lib.tsdr_init.restype = None
    lib.tsdr_free.argtypes = [C.POINTER(vp)]
    lib.tsdr_free.restype = None
    lib.tsdr_getctx.argtypes = [vp]
    lib.tsdr_getctx.restype = vp
    lib.tsdr_getlasterrortext.argtypes = [vp]
    lib.tsdr_getlasterrortext.restype = C.c_char_p
    lib.tsdr_loadplugin.argtypes = [vp, C.c_char_p, C.c_char_p]
    lib.tsdr_unloadplugin.argtypes = [vp]
    lib.tsdr_setresolution.argtypes = [vp, C.c_int, C.c_double]
    lib.tsdr_setbasefreq.argtypes = [vp, C.c_uint32]
    lib.tsdr_setgain.argtypes = [vp, C.c_float]
    lib.tsdr_motionblur.argtypes = [vp, C.c_float]
    lib.tsdr_sync.argtypes = [vp, C.c_int, C.c_int]
    lib.tsdr_setparameter_int.argtypes = [vp, C.c_int, C.c_uint32]
    lib.tsdr_setparameter_double.argtypes = [vp, C.c_int, C.c_double]
    lib.tsdr_readasync.argtypes = [vp, FRAME_CB, vp]
    lib.tsdr_stop.argtypes = [vp]
    lib.tsdr_isrunning.argtypes = [vp]
    lib.tsdr_getsamplerate.argtypes = [vp]
    lib.tsdr_reset.argtypes = [vp]
    lib.tsdr_reset.restype = None
    return lib


RGB_CB = C.CFUNCTYPE(None, C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_void_p)


def throughput_run(libpath, plugin, params, height, fv, seconds, warmup=1.0, setup=None, free=True, rgb=False, dump=None, dump_frames=0, dump_skip=0):
    """Loads `plugin` into the library at `libpath`, lets it stream for `seconds` and counts what reaches the
    callbacks.  Both the reference's pipeline and ours are lossy by design (whole blocks / frames are dropped when a
    stage cannot keep up), so frames delivered per wall second x samples per frame is the effective rate."""
    lib = load(libpath)
    vp = C.c_void_p
    cnt = {"frames": 0, "plots": 0, "w": 0, "h": 0}
    kept = []  # dump: dump_frames float frames as they reach the callback, after the first dump_skip (bench.py's configs[0] leg)

    def on_frame(buf, w, h, ctx):
        if dump and not rgb and cnt["frames"] >= dump_skip and len(kept) < dump_frames:
            import numpy as np
            kept.append(np.ctypeslib.as_array(buf, shape=(w * h,)).copy())
        cnt["frames"] += 1
        cnt["w"], cnt["h"] = w, h

    def on_plot(pid, off, vals, size, rate, ctx):
        cnt["plots"] += 1

    cbs = (FRAME_CB(on_frame), VALUE_CB(lambda *a: None), PLOT_CB(on_plot))
    h = vp()
    lib.tsdr_init(C.byref(h), cbs[1], cbs[2], None)
    pbuf = C.create_string_buffer(params.encode())
    rc = lib.tsdr_loadplugin(h, plugin.encode(), pbuf)
    if rc != 0:
        raise RuntimeError(f"tsdr_loadplugin: {rc} {lib.tsdr_getlasterrortext(h)}")
    lib.tsdr_setgain(h, 0.5)
    lib.tsdr_motionblur(h, float(os.environ.get("TSDR_BENCH_MOTIONBLUR", "0")))  # (bench.py's configs[4] leg: 15/16)
    if lib.tsdr_setresolution(h, height, fv) != 0:
        raise RuntimeError("tsdr_setresolution")
    if setup:
        setup(lib, h)
    status = {}
    if rgb:  # frames as packed RGB converted on the device (include/TSDRLibraryExt.h)
        rgb_cb = RGB_CB(lambda buf, w, hh, ctx: on_frame(buf, w, hh, ctx))
        lib.tsdrx_readasync_rgb.argtypes = [vp, RGB_CB, vp, C.c_int]
        th = threading.Thread(target=lambda: status.setdefault("rc", lib.tsdrx_readasync_rgb(h, rgb_cb, None, 0)))
    else:
        th = threading.Thread(target=lambda: status.setdefault("rc", lib.tsdr_readasync(h, cbs[0], None)))
    th.start()
    time.sleep(warmup)  # device context, buffers, page-locking of the source's memory
    f0, p0, t0 = cnt["frames"], cnt["plots"], time.time()
    time.sleep(seconds)
    f1, p1, t1 = cnt["frames"], cnt["plots"], time.time()
    lib.tsdr_stop(h)
    th.join(30)
    if free:  # the reference's own tsdr_free() frees an uninitialised pointer: callers skip it there
        lib.tsdr_free(C.byref(h))
    if dump and kept:
        import numpy as np
        sizes = {k.size for k in kept}
        np.save(dump, np.stack([k for k in kept if k.size == max(sizes)]))
    return {"frames_per_s": (f1 - f0) / (t1 - t0), "plots_per_s": (p1 - p0) / (t1 - t0), "width": cnt["w"], "height": cnt["h"],
            "status": status.get("rc")}


def throughput_subprocess(libpath, plugin, params, height, fv, seconds, env=None, timeout=120, set_int=(), free=True, rgb=False, dump=None, dump_frames=0,
                          dump_skip=0):
    """throughput_run in a process of its own — what a host application is — started with GPU_MAX_HW_QUEUES=2 like a
    launcher script would (tsdrgpu_core.hip says why) instead of inheriting the caller's runtime state."""
    import json
    import subprocess
    import sys
    e = dict(os.environ)
    # the launcher's job (the library does not touch its host's environment): two hardware queues are the streaming
    # optimum on MI355X (tsdrgpu_core.hip); a caller's env overrides
    e["GPU_MAX_HW_QUEUES"] = "2"
    e.update(env or {})
    extra = [("free" if free else "nofree") + ("+rgb" if rgb else "") + (f"+dump={dump}:{int(dump_frames)}:{int(dump_skip)}" if dump else "")]
    extra += [str(v) for pair in set_int for v in pair]  # tsdr_setparameter_int(id, value) pairs
    out = subprocess.run([sys.executable, "-m", "tempestsdr_amd.tsdrlib", libpath, plugin, params, str(height), str(fv), str(seconds)] + extra,
                         env=e, cwd=os.path.dirname(HERE), capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    if out.returncode != 0 or not lines:
        raise RuntimeError(f"throughput run failed ({out.returncode}): {out.stderr[-500:]}")
    r = json.loads(lines[-1])
    r["stderr_tail"] = out.stderr[-2000:]
    return r


if __name__ == "__main__":
    import json
    import sys
    a = sys.argv[1:]
    pairs = [(int(a[i]), int(a[i + 1])) for i in range(7, len(a) - 1, 2)]
    flags = a[6].split("+") if len(a) >= 7 else ["free"]
    dump_arg = [f for f in flags if f.startswith("dump=")]
    dump_path, dump_n, dump_sk = dump_arg[0][5:].rsplit(":", 2) if dump_arg else (None, "0", "0")
    print(json.dumps(throughput_run(a[0], a[1], a[2], int(a[3]), float(a[4]), float(a[5]), free=(flags[0] == "free"),
                                    rgb=("rgb" in flags), dump=dump_path, dump_frames=int(dump_n), dump_skip=int(dump_sk),
                                    setup=(lambda lib, h: [lib.tsdr_setparameter_int(h, i, v) for i, v in pairs]) if pairs else None)))
