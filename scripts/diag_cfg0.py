#!/usr/bin/env python3
"""Which frames does the reference's threaded library deliver, and which does ours, measured against the deterministic driver (the
oracle: the reference's functions called in order on the same samples)?  8 MS/s recording behind the reference's RawFile plugin
(BASELINE configs[0]); frames 10..70 of the first pass over the recording."""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tempestsdr_amd import tsdrlib, synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402

fs, h, fv = 8_000_000, 525, 60.0
path = "/tmp/diag_cfg0.f32"
iq = synth.synth_iq(fs, "640x480", fv, 2 * fs, seed=0x5EED0000)
iq.tofile(path)
reflib = os.path.join(ROOT, "oracle", "_ref", "libtsdr_ref.so")
rawfile = os.path.join(ROOT, "oracle", "_ref", "libTSDRPlugin_RawFile.so")
geo = orc.geometry(fs, h, fv)
P = geo.width * h
pix, _ = orc.demod_resample_stream(iq, geo)
pp = orc.PostProcess(geo)
want = []
for k in range(min(110, pix.size // P)):
    want.append(pp.run(pix[k * P:(k + 1) * P].copy(), 0.0).copy())
keys = {hash(f.tobytes()): i for i, f in enumerate(want)}
for tag, lib, free in (("reference", reflib, False), ("mi355x", tsdrlib.LIB, True)):
    out = f"/tmp/diag_cfg0_{tag}.npy"
    r = tsdrlib.throughput_subprocess(lib, rawfile, f"{path} {fs} float", h, fv, 1.5, free=free, timeout=120, dump=out, dump_frames=60, dump_skip=10)
    fr = np.load(out)
    hits = [keys.get(hash(f.tobytes())) for f in fr]
    print(tag, "frames/s", round(r["frames_per_s"], 2), "delivered frames 10..69 as oracle frame numbers:", hits[:40])
    miss = [i for i, x in enumerate(hits) if x is None][:3]
    for i in miss:  # not an oracle frame: how far from the nearest one?
        d = [float(np.mean(np.abs(fr[i] - w)[(np.abs(fr[i]) < 250) & (np.abs(w) < 250)])) for w in want]
        j = int(np.argmin(d))
        print("   frame", 10 + i, "nearest oracle frame", j, "mean |diff|", round(d[j], 6), "exact pixels", int(np.sum(fr[i] == want[j])), "of", P)

# Does "the first n blocks were refused by the ring" (dsp_dropped_compensation_add's failure branch, dsp.c:338-345: every refused
# block of 262 144 samples becomes a skip of block = round(2 S) = 266 667 samples) explain the reference's frames?
ref_frames = np.load("/tmp/diag_cfg0_reference.npy")
rk = {hash(f.tobytes()): i for i, f in enumerate(ref_frames)}
block = int(round(((geo.width * h) << 1) * geo.pixeltimeoversampletime))
for n in range(0, 13):
    start = n * block
    pix_n, _ = orc.demod_resample_stream(iq[2 * start:], geo)
    pp_n = orc.PostProcess(geo)
    hits = []
    for k in range(min(40, pix_n.size // P)):
        f = pp_n.run(pix_n[k * P:(k + 1) * P].copy(), 0.0)
        hits.append(rk.get(hash(f.tobytes())))
    found = [(k, v) for k, v in enumerate(hits) if v is not None]
    print("skipped", n, "blocks of", block, "samples at the start ->", len(found), "of the driver's first 40 frames are frames the reference delivered", found[:3])
