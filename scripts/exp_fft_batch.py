"""Experiment: autocorrelation time vs windows per launch batch (Infinity-Cache residency)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tempestsdr_amd import gpu
g = gpu.TsdrGpu(0)
fs = 100_000_000
ac = gpu.Autocorr(g, fs)
nwin = 16
x = g.empty(2 * nwin * ac.capture)
x.upload(np.random.default_rng(0).random(2 * nwin * ac.capture).astype(np.float32))
for per in (16, 8, 4, 2, 1):
    for rep in range(2):
        g.sync(); g.timer_start()
        for it in range(5):
            ac.reset()
            for s in range(0, nwin, per):
                ac.run(x, True, ac.capture, per, in_offset=2 * s * ac.capture)
        ms = g.timer_stop_ms() / 5
    print(f"windows per batch {per:2d}: {ms:.3f} ms for {nwin} windows ({ms / nwin * 1000:.1f} us/window)")
