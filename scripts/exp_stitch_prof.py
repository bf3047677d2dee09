#!/usr/bin/env python3
"""The super-bandwidth stitch of bench.py's side leg (4 hops x 10 frames of the 100 MS/s stream) a few times in a row, for
a kernel trace:  cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d out -o t -- python scripts/exp_stitch_prof.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tempestsdr_amd import build, gpu  # noqa: E402

build.build(verbose=False)
g = gpu.TsdrGpu(0)
fs, fv = 100_000_000, 60.0
sif = int(fs / fv)
gathered = 10 * sif
per = 1 << (gathered.bit_length() - 1)
rng = np.random.default_rng(1)
base = (rng.standard_normal(2 * gathered) * 0.1).astype(np.float32)
hops = []
for k in range(4):
    h = np.roll(base, 2 * 1000 * k) + (rng.standard_normal(2 * gathered) * 0.01).astype(np.float32)
    hops.append(h.astype(np.float32))
d_hops = [g.to_device(h) for h in hops]
d_st = g.empty(2 * 4 * per)
if "--pmc-calibrate" in sys.argv:
    # a launch whose traffic is known exactly (8 B read + 4 B written per sample): scripts/pmc_summarize.py calibrates the
    # FETCH_SIZE / WRITE_SIZE counters of the same rocprofv3 run on it
    ncal = 99_999_600
    d_cal_in = g.to_device(np.zeros(2 * ncal, np.float32))
    d_cal_out = g.empty(ncal)
    for _ in range(3):
        g.am_demod(d_cal_in, d_cal_out, ncal)
    g.sync()
for rep in range(6):
    g.sync()
    t = time.perf_counter()
    offs, total = g.superb_stitch(d_hops, gathered, sif, d_st)
    dt = time.perf_counter() - t
    print(f"stitch {rep}: {dt * 1e3:.3f} ms, offsets {list(offs)}", flush=True)
