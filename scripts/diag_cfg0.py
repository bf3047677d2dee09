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

# What the reference's threaded library delivers instead.  Its rings start at 2 floats and grow by the adds (circbuff.c:64-110); a block
# a ring refuses meanwhile becomes a skip of d x block samples, block = round(2 S) = 266 667 (dsp.c:338-345, TSDRLibrary.c:283-284): a
# third of a sample off the raster per block.  Test: is every delivered frame an exact AFFINE image (the autogain's normalisation,
# dsp.c:74 — its state depends on the frames lost before, so the values differ, the pixels do not) of a RAW driver frame of the recording
# with d x 266 667 samples removed?  (Round 5, this container: frames 7..59 of a run are images of raw frames 14..66 at d = 2, residual
# 6e-8, consecutive; frame 2 of raw frame 4 at d = 1; frames 0, 1, 3, 6 straddle a gap.)
ref_frames = np.load("/tmp/diag_cfg0_reference.npy")
block = int(round(((geo.width * h) << 1) * geo.pixeltimeoversampletime))
sub = slice(0, P, 11)
found = {}
for d in range(0, 7):
    px, _ = orc.demod_resample_stream(iq[2 * d * block:2 * (d * block + 9_000_000)], geo)
    raw = [px[k * P:(k + 1) * P][sub].astype(np.float64) for k in range(px.size // P)]
    for i, f in enumerate(ref_frames):
        if i in found:
            continue
        y = f[sub].astype(np.float64)
        m = np.abs(y) < 250
        for j, x in enumerate(raw):
            vx = x[m] - x[m].mean()
            a = float((vx * (y[m] - y[m].mean())).sum() / (vx * vx).sum())
            b = float(y[m].mean() - a * x[m].mean())
            res = float(np.max(np.abs(a * x[m] + b - y[m])))
            if res < 1e-5:
                found[i] = (d, j, round(a, 5), round(b, 5), res)
                break
    print("d =", d, ": delivered frames explained so far:", len(found), "of", len(ref_frames), flush=True)
for i in range(len(ref_frames)):
    print("delivered frame", 10 + i, "-> (blocks removed, raw driver frame, a, b, max residual)", found.get(i))
