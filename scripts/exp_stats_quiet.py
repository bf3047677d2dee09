#!/usr/bin/env python3
"""What does k_frame_stats see when it does NOT follow an 800 MB producer?  The frame path of the headline config run
(a) back to back, as bench.py runs it (the statistics kernel starts while the dirty lines of the previous kernel's
800 MB are still draining from the Infinity Cache), and (b) with the device idle for 20 ms before every run.  Per-kernel
times from the library's own event pairs."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tempestsdr_amd import build, gpu  # noqa: E402

build.build(verbose=False)
g = gpu.TsdrGpu(0)
F, W, H = 60, 2962, 1125
P = W * H
rng = np.random.default_rng(1)
fr = rng.random((4, P), dtype=np.float32)
d_frames = g.empty(F * P)
for f in range(F):
    g._ck(g.lib.tsdrgpu_upload(g.h, d_frames.ptr + 4 * f * P, fr[f % 4].ctypes.data, 4 * P))
d_out = g.empty(F * P)
pp = gpu.PostProcess(g)
g.sync()


def run(label, quiet, n=12):
    for _ in range(3):
        pp.run(d_frames, F, W, H, d_out)
    g.sync()
    g.profile_begin()
    for _ in range(n):
        if quiet:
            g.sync()
            time.sleep(0.02)
        pp.run(d_frames, F, W, H, d_out)
    g.sync()
    prof = g.profile_end()
    print(label, {k: round(v[0] / n, 4) for k, v in prof.items()}, flush=True)
    st = prof.get("k_frame_stats")
    if st:
        print("   k_frame_stats %.4f ms -> %.2f TB/s (4P = %.1f MB)" % (st[0] / n, 4e-9 * F * P / (st[0] / n), 4e-6 * F * P), flush=True)


run("back to back      ", False)
run("idle 20 ms before ", True)
run("back to back      ", False)
