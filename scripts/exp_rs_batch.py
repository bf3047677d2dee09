"""Experiment: the sample-parallel resampler (k_rs_area_up, frame tracking on) alone, per second of 100 MS/s signal, against the
batch length — bench.py's 4 s batches showed it 7 % slower per sample than its 1 s batches.  usage: exp_rs_batch.py [seconds ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from tempestsdr_amd import gpu  # noqa: E402

fs, h, fv = 100_000_000, 1125, 60.0
W = 2962  # the library's geometry for this mode (bench.py geometry())
P = W * h
g = gpu.TsdrGpu(0)
chunk = int(0.1 * fs / fv)
rng = np.random.default_rng(0)
for seconds in [float(a) for a in sys.argv[1:]] or [1.0, 2.0, 4.0]:
    nchunks = int(seconds * fs) // chunk
    n = nchunks * chunk
    d_iq = g.empty(2 * n)
    blk = rng.random(2 * chunk * 10).astype(np.float32)
    for off in range(0, 2 * n, blk.size):
        k = min(blk.size, 2 * n - off)
        d_iq.upload(blk[:k], off)
    d_pix = g.empty(int(n * (W * h * fv / fs)) + 2 * P)
    for track in (0, 1):
        rs = gpu.Resampler(g)
        if track:
            rs.track_frames(P, 0)
        for _ in range(int(300 / seconds)):  # ~0.1 s of work first: the clocks of an idle device take that long to come up
            rs.process(d_iq, 1, chunk, nchunks, W * h * fv, float(fs), 0, d_pix)
        g.sync()
        reps = int(200 / seconds)
        g.timer_start()
        for _ in range(reps):
            rs.process(d_iq, 1, chunk, nchunks, W * h * fv, float(fs), 0, d_pix)
        ms = g.timer_stop_ms() / reps
        print(f"batch {seconds:g} s, tracking {track}: {ms:.4f} ms per call = {ms / seconds:.4f} ms per second of signal "
              f"({(8 * n + 4 * n * W * h * fv / fs) / ms / 1e6:.0f} GB/s)", flush=True)
        rs.destroy() if hasattr(rs, "destroy") else None
    del d_iq, d_pix
