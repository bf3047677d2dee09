#!/usr/bin/env python3
"""A/B of engine knobs through the tsdr_* API: usage e2e_exp.py '[["name","f32"|"i16",{"ENV":"value",...}],...]' — the legs
run one after the other in THIS process (so that first / later engines of a process can be compared); PARAM_ID=n in a
leg's environment sets tsdr_setparameter_int(n, 1) for it."""
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
f32 = f"{path} {fs} {block} 0 0"; i16 = f"{path16} {fs} {block} 0 0 int16"
for name, params, env in json.loads(sys.argv[1]):
    leg(name, f32 if params == "f32" else i16, env)
