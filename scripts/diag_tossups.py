#!/usr/bin/env python3
"""Diagnostic (GPU): how often, and where, the sync detector's decisions are toss-ups on the bench stream."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from tempestsdr_amd import gpu
fs, h, fv = 100_000_000, 1125, 60.0
W = bench.geometry(fs, h, fv); P = W * h
chunk = int(0.1 * fs / fv); nchunks = int(fs) // chunk; nsamples = nchunks * chunk
dev = torch.device("cuda", 0)
g = gpu.TsdrGpu(0)
iq = bench.synth_iq_torch(fs, "1920x1080", fv, nsamples, 0, 0x5EED0003, dev); torch.cuda.synchronize()
d_iq = bench.DevPtr(iq)
rs = gpu.Resampler(g); pp = gpu.PostProcess(g)
up, down = W * h * fv, float(fs)
pix = torch.empty(int(nsamples * up / down) + 64 + P, dtype=torch.float32, device=dev)
out = torch.empty((pix.numel() // P + 1) * P, dtype=torch.float32, device=dev)
d_pix, d_out = bench.DevPtr(pix), bench.DevPtr(out)
carry = 0
for it in range(12):
    n = rs.process(d_iq, 1, chunk, nchunks, up, down, 0, d_pix, in_offset=0, out_offset=carry)
    avail = carry + n; F = avail // P
    infos = pp.run(d_pix, F, W, h, d_out, motionblur=0.0, want_info=True)
    fl, toss, fresh = pp.redo_raw()
    print(f"pass {it}: F={F} toss x/y {toss[:,0].sum()}/{toss[:,1].sum()} fresh x/y {fresh[:,0].sum()}/{fresh[:,1].sum()} upfront {fl.sum() - fresh.sum()}"
          f" dx {infos[0].dx}..{infos[F-1].dx} strip x {infos[0].stripx}..{infos[F-1].stripx} dy {infos[0].dy}..{infos[F-1].dy} strip y {infos[0].stripy}..{infos[F-1].stripy}")
    rem = avail - F * P
    if rem and F: g._ck(g.lib.tsdrgpu_copy(g.h, d_pix.at(0), d_pix.at(F * P), rem * 4))
    carry = rem
