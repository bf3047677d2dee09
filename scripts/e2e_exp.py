#!/usr/bin/env python3
"""A/B of engine knobs through the tsdr_* API (experiment helper)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tempestsdr_amd import synth, tsdrlib
import numpy as np
fs, h, fv = 100_000_000, 1125, 60.0
path = "/tmp/e2e_iq.f32"; block = 524288
n = (int(0.3 * fs) // (block // 2)) * (block // 2)
with open(path, "wb") as f:
    for s in range(0, n, 1 << 22):
        synth.synth_iq(fs, "1920x1080", fv, min(1 << 22, n - s), start=s).tofile(f)
path16 = "/tmp/e2e_iq.s16"
a = np.fromfile(path, np.float32)
np.clip(np.round(a * 20000.0), -32768, 32767).astype(np.int16).tofile(path16)
S = fs / fv
def leg(name, params, env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    os.environ["TSDR_GPU_STATS"] = "1"
    try:
        setup = (lambda lib, hh: lib.tsdr_setparameter_int(hh, int(os.environ["PARAM_ID"]), 1)) if os.environ.get("PARAM_ID") else None
        r = tsdrlib.throughput_run(tsdrlib.LIB, tsdrlib.MEM_PLUGIN, params, h, fv, 3.0, setup=setup)
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
    print(name, round(r["frames_per_s"] * S / 1e6, 1), "MS/s", flush=True)
if os.environ.get("DUMMY"):
    from tempestsdr_amd import gpu
    g0 = gpu.TsdrGpu(0)
    g0.sync()
    if os.environ["DUMMY"] == "close":
        g0.close()
    print("dummy context", os.environ["DUMMY"], flush=True)
if os.environ.get("BURN"):
    import time
    from tempestsdr_amd import gpu
    g = gpu.TsdrGpu(0)
    x = g.to_device(np.zeros(1 << 26, np.float32)); y = g.to_device(np.zeros(1 << 25, np.float32))
    t0 = time.time(); n = 0
    while time.time() - t0 < 3.0:
        for _ in range(50): g._ck(g.lib.tsdrgpu_am_demod(g.h, x.ptr, y.ptr, 1 << 25))
        g.sync(); n += 50
    print("burn: ", n, "demod launches of 32M samples", flush=True)
    g.close()
f32 = f"{path} {fs} {block} 0 0"; i16 = f"{path16} {fs} {block} 0 0 int16"
for name, params, env in json.loads(sys.argv[1]):
    leg(name, f32 if params == "f32" else i16, env)
