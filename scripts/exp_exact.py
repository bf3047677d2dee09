"""Experiment: the exact autocorrelation (tsdrgpu_autocorr_set_exact) per window, for A/B runs of the trip kernels
(TSDRGPU_FFTX_GENERIC, TSDRGPU_FFTX_U, TSDRGPU_XBATCH are read by the library).  usage: exp_exact.py [fs] [nwin] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from tempestsdr_amd import gpu  # noqa: E402

fs = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
nwin = int(sys.argv[2]) if len(sys.argv) > 2 else 16
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
g = gpu.TsdrGpu(0)
ac = gpu.Autocorr(g, fs)
ac.set_exact(True)
x = g.empty(2 * nwin * ac.capture)
x.upload(np.random.default_rng(0).random(2 * nwin * ac.capture).astype(np.float32))
ac.run(x, True, ac.capture, nwin)
g.sync()
g.timer_start()
for _ in range(reps):
    ac.reset()
    ac.run(x, True, ac.capture, nwin)
ms = g.timer_stop_ms() / reps
env = {k: os.environ[k] for k in ("TSDRGPU_FFTX_GENERIC", "TSDRGPU_FFTX_U", "TSDRGPU_XBATCH") if k in os.environ}
print(f"exact autocorrelation fs={fs} N=2^{int(np.log2(ac.n))} {env}: {ms / nwin * 1000:.1f} us/window ({nwin} windows, {reps} reps)", flush=True)
