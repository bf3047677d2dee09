#!/usr/bin/env python3
"""Randomised differential run of the HIP path against the oracle (not part of the pytest suites: a soak).

  resampler : random ratios (library-like r ~ 2, and arbitrary 0.05..7), random chunk sizes / counts, float and
              IQ input, state carried over several calls                         -> bit-exact pixels + carried state
  post-proc : random frame geometries 3x3 .. ~900x300, the 4 stage orders x autoshift x motion blur, random batch
              splits, frames with sentinels / constant frames                    -> bit-exact frames, identical state
  autocorr  : random sample rates                                                -> plots within 1e-4*max
  fft       : random power-of-two sizes 2 .. 2^17, both directions                -> 2e-6*max|X|
  plot      : random plot sizes, widget widths, zoom / offset states              -> identical columns, lowest/highest, argmax
  tracking  : random frame sizes / phases, several calls                          -> exact per-frame min/max
  superb    : four randomly delayed hops of a periodic signal                     -> identical hop offsets, 1e-4*max signal
  bands     : random geometries cut into 1..5 row bands, motion blur 0 / above, batches run in the FUSED band form (range
              exchanged first, one trip) and the two-trip form in random alternation on the same objects, frames with
              sentinels / -0.0 / constant frames / noiseless patterns              -> bit-exact frames, identical state

usage (on a GPU box):  python scripts/fuzz_parity.py [cases] [seed]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from oracle import oracle as orc  # noqa: E402
from tempestsdr_amd import gpu  # noqa: E402
import cases  # noqa: E402

# exact ties (tsdrgpu_postproc_set_exact_ties) on: the sync state must be identical in every case; FUZZ_DEFAULT_TIES=1
# runs the default mode instead, whose rare toss-up flips (a few per 10^5 decisions) then show up as mismatches
EXACT_TIES = os.environ.get("FUZZ_DEFAULT_TIES", "0") != "1"


def fuzz_resampler(g, rng):
    if rng.random() < 0.6:
        x = rng.uniform(3.0, 4000.0)
        up, down = float(int(2 * x)), x  # the library's r = floor(2x)/x
    else:
        up, down = float(rng.uniform(0.05, 7.0)), 1.0
    iq = bool(rng.integers(0, 2))
    nearest = bool(rng.random() < 0.2)  # PARAM_NEAREST_NEIGHBOUR_RESAMPLING, dsp.c:274-276
    rs_o, rs_g = orc.Resampler(), gpu.Resampler(g)
    for _ in range(int(rng.integers(1, 4))):
        chunk = int(rng.integers(1, 3000))
        nch = int(rng.integers(1, 9))
        n = chunk * nch
        mag = rng.random(n).astype(np.float32) * np.float32(rng.choice([1.0, 50.0, 1e-3]))
        if iq:
            ph = rng.random(n) * 6.28
            host = np.empty(2 * n, np.float32)
            host[0::2] = (mag * np.cos(ph)).astype(np.float32)
            host[1::2] = (mag * np.sin(ph)).astype(np.float32)
            mag = orc.am_demod(host)
        else:
            host = mag
        want = np.concatenate([rs_o.process(mag[c * chunk:(c + 1) * chunk], up, down, nearest) for c in range(nch)])
        cap = rs_g.count(chunk, nch, up, down)
        d_out = g.empty(cap + 8)
        npix = rs_g.process(g.to_device(host), iq, chunk, nch, up, down, int(nearest), d_out)
        got = d_out.download()[:npix]
        if npix != want.size or not np.array_equal(got, want, equal_nan=True):
            return f"resampler up={up} down={down} iq={iq} nearest={nearest} chunk={chunk} nch={nch}"
        con, off = rs_g.state()
        if (con, off) != (rs_o.st.contrib, rs_o.st.offset):
            return f"resampler state up={up} down={down} chunk={chunk}"
    return None


def fuzz_postproc(g, rng):
    if rng.random() < 0.15:  # tiny geometries
        h = int(rng.integers(2, 12))
        fs = int(rng.integers(2, 40)) * 30 * h
    else:
        h = int(rng.integers(3, 300))
        fs = int(rng.integers(20_000, 3_000_000))
    geo = orc.geometry(fs, h, 60.0)
    w = geo.width
    if w < 2 or w > 3200 or w * h > 400_000:
        return None
    cfg = (int(rng.integers(0, 2)), int(rng.integers(0, 2)), int(rng.integers(0, 2)), 0,
           float(rng.choice([0.0, 0.0, 0.25, 0.5, 0.9375])))
    lbs, aap, ash, pll, mb = cfg
    # batches of >= 8 frames take the frame-parallel pass and, in the fused run at motion blur 0, the flat fused trip
    F = int(rng.integers(1, 9)) if rng.random() < 0.6 else int(rng.integers(8, 21))
    frames = [cases.frame_pattern(w, h, int(rng.integers(0, 50)), rng) for _ in range(F)]
    for fr in frames:
        r = rng.random()
        if r < 0.15:
            fr[rng.integers(0, w * h, 5)] = np.float32(rng.choice([256.0, 512.0, 1024.0, 2048.0, -300.0]))
        elif r < 0.2:
            fr[:] = np.float32(rng.random())                      # uniform frame
        elif r < 0.25:
            fr[:] = np.where(fr > np.median(fr), np.float32(0.75), np.float32(0.125))  # two-valued
        elif r < 0.3:
            fr *= np.float32(rng.choice([200.0, 1e-4, -1.0]))     # large / tiny / negative ranges
        elif r < 0.33:
            fr[rng.integers(0, w * h, max(1, w * h // 3))] = np.float32(1024.0)  # a third of the frame sentinel
        elif r < 0.36:
            fr[rng.integers(0, w * h, 3)] = np.float32(-0.0)      # -0.0: the frame-parallel forms must hand the batch to the literal pass
        elif r < 0.43:  # periodic structure: checkerboard / stripes with random pitch
            yy, xx = np.mgrid[0:h, 0:w]
            px, py = int(rng.integers(1, 20)), int(rng.integers(1, 20))
            pat = ((xx // px + (yy // py) * int(rng.integers(0, 2))) % 2) * np.float32(0.55) + np.float32(0.2)
            fr[:] = pat.astype(np.float32).reshape(-1)
    opp = orc.PostProcess(geo)
    want, states = [], []
    for fr in frames:
        want.append(opp.run(fr.copy(), mb, 0.1, lbs, aap, ash, pll, 0))
        states.append(opp.state())
    pp = gpu.PostProcess(g)
    pp.set_exact_ties(EXACT_TIES)
    d_in = g.to_device(np.concatenate(frames))
    d_out = g.empty(F * w * h)
    infos, s = [], 0
    while s < F:
        k = int(rng.integers(1, F - s + 1)) if rng.random() < 0.5 else F - s
        mode = int(rng.integers(0, 3))
        if mode == 0:
            infos += pp.run(d_in, k, w, h, d_out, mb, 0.1, lbs, aap, ash, pll, 0, frames_offset=s * w * h, out_offset=s * w * h)
        elif mode == 1:
            pp.begin(d_in, k, w, h, mb, 0.1, lbs, aap, ash, pll, 0, frames_offset=s * w * h)
            infos += pp.finish(d_out, out_offset=s * w * h)
        else:  # fused run with the per-frame min/max supplied
            mn = np.array([fr[np.abs(fr) <= 250].min() if np.any(np.abs(fr) <= 250) else np.inf for fr in frames[s:s + k]], np.float32)
            mx = np.array([fr[np.abs(fr) <= 250].max() if np.any(np.abs(fr) <= 250) else -np.inf for fr in frames[s:s + k]], np.float32)
            d_mn, d_mx = g.to_device(mn), g.to_device(mx)
            pp.begin_minmax(d_in, k, w, h, d_mn.ptr, d_mx.ptr, d_out, mb, 0.1, lbs, aap, ash, pll, 0,
                            frames_offset=s * w * h, out_offset=s * w * h)
            infos += pp.finish(d_out, out_offset=s * w * h)
        s += k
    got = d_out.download().reshape(F, -1)
    for k in range(F):
        si, sd = states[k]
        i = infos[k]
        if (i.dx, i.vx, i.stripx, i.dy, i.vy, i.stripy, i.locked) != tuple(si[:7]):
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            np.savez_compressed(os.path.join(ROOT, "gpurun_out", f"fuzz_pp_{fs}_{h}.npz"), frames=np.stack(frames), fs=fs, h=h,
                                cfg=np.array(cfg), frame=k)
            return (f"postproc state fs={fs} h={h} w={w} cfg={cfg} F={F} frame={k} gpu="
                    f"{(i.dx, i.vx, i.stripx, i.dy, i.vy, i.stripy, i.locked)} oracle={tuple(int(v) for v in si[:7])}")
        if not np.array_equal(got[k], want[k], equal_nan=True):
            return f"postproc frame fs={fs} h={h} w={w} cfg={cfg} F={F} frame={k} maxdiff={np.nanmax(np.abs(got[k] - want[k]))}"
        if not np.array_equal(np.signbit(got[k]), np.signbit(want[k])):
            return f"postproc frame fs={fs} h={h} w={w} cfg={cfg} F={F} frame={k}: sign of a zero differs"
    return None


def fuzz_postproc_pll(g, rng):
    """PLL on: one frame per call, the caller applies the nudge (rate and state must track the oracle's)."""
    h = int(rng.integers(20, 200))
    fs = int(rng.integers(100_000, 2_500_000))
    geo = orc.geometry(fs, h, 60.0)
    w = geo.width
    if w < 8 or w * h > 250_000:
        return None
    cfg = (int(rng.integers(0, 2)), int(rng.integers(0, 2)), int(rng.integers(0, 2)), 1, float(rng.choice([0.0, 0.5, 0.9375])))
    lbs, aap, ash, pll, mb = cfg
    drift = int(rng.integers(0, 6))
    pp_o, pp_g = orc.PostProcess(geo), gpu.PostProcess(g)
    pp_g.set_exact_ties(EXACT_TIES)
    d_in, d_out = g.empty(w * h), g.empty(w * h)
    rate = 60.0
    hist = []

    def dump(what):
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        np.savez_compressed(os.path.join(ROOT, "gpurun_out", f"fuzz_pll_{fs}_{h}.npz"), frames=np.stack(hist), fs=fs, h=h, cfg=np.array(cfg))
        return what

    for k in range(int(rng.integers(3, 14))):
        if geo.width != w:
            break
        fr = cases.frame_pattern(w, h, k * drift, rng)
        hist.append(fr.copy())
        want = pp_o.run(fr.copy(), mb, 0.1, lbs, aap, ash, 1, 0)
        d_in.upload(fr)
        info = pp_g.run(d_in, 1, w, h, d_out, mb, 0.1, lbs, aap, ash, 1, 0)[0]
        rate -= info.frameratediff
        si, sd = pp_o.state()
        if rate != geo.refreshrate or info.pll_fired != si[7]:
            return dump(f"pll rate fs={fs} h={h} cfg={cfg} frame={k} gpu={rate} oracle={geo.refreshrate}")
        if (info.dx, info.vx, info.stripx, info.dy, info.vy, info.stripy, info.locked) != tuple(si[:7]):
            return dump(f"pll state fs={fs} h={h} cfg={cfg} frame={k} gpu={(info.dx, info.vx, info.stripx, info.dy, info.vy, info.stripy, info.locked)} oracle={tuple(int(v) for v in si[:7])}")
        if not np.array_equal(d_out.download(), want, equal_nan=True):
            return f"pll frame fs={fs} h={h} cfg={cfg} frame={k}"
    return None


def fuzz_autocorr_multi(g, rng):
    """several windows, IQ or magnitude input, running mean over two calls, argmax rule"""
    fs = int(rng.integers(100_000, 900_000))
    ac_o, ac = orc.Autocorr(fs), gpu.Autocorr(g, fs)
    nwin = int(rng.integers(2, 6))
    from_iq = bool(rng.integers(0, 2))
    period = fs // int(rng.integers(56, 86))
    n = ac.capture * nwin
    t = np.arange(n)
    mag = (rng.random(n) * 0.4 + (t % period < period // 9) + 0.3 * ((t % max(period // int(rng.integers(100, 400)), 2)) == 0)).astype(np.float32)
    if from_iq:
        ph = 0.37 * t
        host = np.empty(2 * n, np.float32)
        host[0::2] = (mag * np.cos(ph)).astype(np.float32)
        host[1::2] = (mag * np.sin(ph)).astype(np.float32)
        mag = orc.am_demod(host)
    else:
        host = mag
    for k in range(nwin):
        ac_o.run(mag[k * ac.capture:(k + 1) * ac.capture])
    d = g.to_device(host)
    if rng.random() < 0.5:  # exact mode: bit-identical plots and argmax
        ac.set_exact(True)
        ac.run(d, from_iq, ac.capture, nwin)
        f, l, _ = ac.plots()
        if not (np.array_equal(f, ac_o.frame) and np.array_equal(l, ac_o.line)):
            return f"autocorr exact fs={fs} nwin={nwin} iq={from_iq}"
        if ac.argmax() != (int(np.argmax(ac_o.frame)), int(np.argmax(ac_o.line))):
            return f"autocorr exact argmax fs={fs}"
        return None
    # certified mode (the engine's default): float32 transform + argmax certificate, exact replay when it fails;
    # the answer must be the oracle's argmax, the float32 plots within the bound the certificate rests on
    ac.set_certify(int(rng.integers(1, 3)))
    first = int(rng.integers(1, nwin))
    ac.run(d, from_iq, ac.capture, first)
    ac.run(d, from_iq, ac.capture, nwin - first, in_offset=first * ac.capture * (2 if from_iq else 1))
    f, l, calls = ac.plots()
    if calls != nwin:
        return f"autocorr calls fs={fs}"
    if np.max(np.abs(f - ac_o.frame)) > 1e-4 * np.max(ac_o.frame) or np.max(np.abs(l - ac_o.line)) > 1e-4 * np.max(ac_o.line):
        return f"autocorr multi fs={fs} nwin={nwin} iq={from_iq}"
    fi, li = ac.argmax()
    if (fi, li) != (int(np.argmax(f)), int(np.argmax(l))):
        return f"autocorr argmax rule fs={fs}"
    c = ac.certificate()
    if max(np.max(np.abs(f - ac_o.frame)), np.max(np.abs(l - ac_o.line))) > 0.5 * 8e-6 * c.r0:
        return f"autocorr certificate premise fs={fs} nwin={nwin}: plots further from the oracle's than KAPPA/2 * R0"
    if (c.frame_certified and fi != int(np.argmax(ac_o.frame))) or (c.line_certified and li != int(np.argmax(ac_o.line))):
        return f"autocorr CERTIFIED argmax differs fs={fs} nwin={nwin}"
    fi, li, promoted = ac.argmax_certified()
    if (fi, li) != (int(np.argmax(ac_o.frame)), int(np.argmax(ac_o.line))):
        return f"autocorr certified-mode argmax fs={fs} nwin={nwin} promoted={promoted}"
    if promoted:
        f, l, _ = ac.plots()
        if not (np.array_equal(f, ac_o.frame) and np.array_equal(l, ac_o.line)):
            return f"autocorr promoted plots fs={fs} nwin={nwin}"
    return None


def fuzz_autocorr(g, rng):
    fs = int(rng.integers(100_000, 1_200_000))
    ac_o, ac = orc.Autocorr(fs), gpu.Autocorr(g, fs)
    period = fs // int(rng.integers(56, 86))
    x = (rng.random(ac.capture) * 0.5 + (np.arange(ac.capture) % period < period // 10)).astype(np.float32)
    ac_o.run(x)
    ac.run(g.to_device(x), False, ac.capture, 1)
    f, l, _ = ac.plots()
    if np.max(np.abs(f - ac_o.frame)) > 1e-4 * np.max(ac_o.frame) or np.max(np.abs(l - ac_o.line)) > 1e-4 * np.max(ac_o.line):
        return f"autocorr fs={fs}"
    return None


def fuzz_fft(g, rng):
    n = 1 << int(rng.integers(1, 18))
    inverse = bool(rng.integers(0, 2))
    z = (rng.standard_normal(2 * n) * rng.choice([1.0, 1e3, 1e-3])).astype(np.float32)
    want = orc.fft_perform(z, inverse)
    d = g.to_device(z)
    if rng.random() < 0.5:  # exact form
        g.fft_perform(d, n, inverse, exact=True)
        return None if np.array_equal(d.download(), want) else f"fft exact n={n} inverse={inverse}"
    g._ck(g.lib.tsdrgpu_fft(g.h, d.ptr, n, int(inverse)))
    got = d.download()
    if np.max(np.abs(got - want)) > 2e-6 * max(np.max(np.abs(want)), 1e-30) * max(1.0, np.log2(n) / 4):
        return f"fft n={n} inverse={inverse} err={np.max(np.abs(got - want)) / np.max(np.abs(want))}"
    return None


def fuzz_plot(g, rng):
    size = int(rng.integers(1, 60_000))
    nwidth = int(rng.integers(1, 2500))
    data = rng.random(size) * rng.choice([1.0, 1e6, 1e-9])
    if rng.random() < 0.3:
        data[rng.integers(0, size, 4)] = data.max() * 2  # ties for the argmax
    if rng.random() < 0.2:
        data[:] = data[0]
    so, sg = orc.PlotScale(), gpu.PlotScale()
    zoom = float(rng.choice([1.0, 1.0, rng.uniform(0.001, 1.0)]))
    offpx = int(rng.choice([0, 0, rng.integers(-200, 5000)]))
    for sc in (so, sg):
        span = float(size) * zoom
        sc.one_val_in_pixels = nwidth / span
        sc.one_px_in_values = span / nwidth
        sc.offset_px = offpx
        sc.offset_val = offpx * sc.one_px_in_values
        sc.min_value = 0.0
    want = orc.plot_populate(data, nwidth, so)
    d = g.empty(2 * size, np.float32)
    g._ck(g.lib.tsdrgpu_upload(g.h, d.ptr, data.ctypes.data, data.nbytes))
    g.sync()
    got = g.plot_columns(d.ptr, size, nwidth, sg)
    if not np.array_equal(got[0], want[0]) or got[1:] != want[1:]:
        return f"plot size={size} nwidth={nwidth} zoom={zoom} offpx={offpx}"
    return None


def fuzz_tracking(g, rng):
    P = int(rng.integers(4096, 60_000))
    phase = int(rng.integers(0, P))
    rs = gpu.Resampler(g)
    rs.track_frames(P, phase)
    x = rng.uniform(3.0, 4000.0)
    up, down = float(int(2 * x)), x
    stream, mns, mxs = [], [], []
    for _ in range(int(rng.integers(1, 4))):
        chunk, nch = int(rng.integers(50, 9000)), int(rng.integers(1, 12))
        n = chunk * nch
        v = (rng.random(n) * 3 - 1).astype(np.float32)
        if rng.random() < 0.5:
            v[rng.integers(0, n, 3)] = np.float32(rng.choice([900.0, -700.0]))
        cap = rs.count(chunk, nch, up, down)
        d_out = g.empty(cap + 8)
        npix = rs.process(g.to_device(v), False, chunk, nch, up, down, 0, d_out)
        stream.append(d_out.download()[:npix])
        mn, mx = rs.frame_minmax()
        mns += list(mn)
        mxs += list(mx)
    s = np.concatenate([np.full(phase, np.float32(1e9), np.float32)] + stream)
    nfr = s.size // P
    if len(mns) != nfr:
        return f"tracking count P={P} phase={phase}"
    for f in range(nfr):
        fr = s[f * P:(f + 1) * P]
        if f == 0:
            fr = fr[phase:]
        ok = fr[np.abs(fr) <= 250.0]
        want = (ok.min(), ok.max()) if ok.size else (np.float32(np.inf), np.float32(-np.inf))
        if (mns[f], mxs[f]) != want:
            return f"tracking P={P} phase={phase} frame={f}"
    return None


def fuzz_superb(g, rng):
    # four hops of a periodic signal, each delayed by a random amount (superbandwidth.c:121-152)
    # (two cases in five with hops of 2^16 .. 2^19 points: the three-trip plan of tsdrgpu_superb_stitch; the others take the
    # pass-per-radix plan)
    fs = int(rng.integers(600_000, 3_000_000)) if rng.random() < 0.4 else int(rng.integers(20_000, 400_000))
    fv = float(rng.choice([50.0, 60.0, 75.0]))
    sif = int(fs / fv)
    gathered = int(rng.integers(2 * sif + 8, 10 * sif))
    t = np.arange(gathered + 2 * sif)
    base = (0.4 + 0.5 * ((t % sif) < sif // 7) + 0.1 * np.sin(t * 0.01)).astype(np.float64)
    hops = []
    for k in range(4):
        d = int(rng.integers(0, sif))
        mag = base[d:d + gathered] + rng.standard_normal(gathered) * 0.01
        ph = 0.21 * np.arange(gathered) + k
        h = np.empty(2 * gathered, np.float32)
        h[0::2] = (mag * np.cos(ph)).astype(np.float32)
        h[1::2] = (mag * np.sin(ph)).astype(np.float32)
        hops.append(h)
    want, offs = orc.superb_stitch(hops, sif)
    d_out = g.empty(want.size)
    if rng.random() < 0.5:  # exact form: everything bit-identical
        got_offs, total = g.superb_stitch([g.to_device(h) for h in hops], gathered, sif, d_out, exact=True)
        nfft = 1 << int(np.floor(np.log2(max(total, 1))))
        if 2 * total != want.size or not np.array_equal(got_offs, offs) or not np.array_equal(d_out.download()[:2 * nfft], want[:2 * nfft]):
            return f"superb exact fs={fs} fv={fv} gathered={gathered}"
        return None
    got_offs, total = g.superb_stitch([g.to_device(h) for h in hops], gathered, sif, d_out)
    if 2 * total != want.size:
        return f"superb size fs={fs} gathered={gathered}"
    if not np.array_equal(got_offs, offs):
        # offsets are argmaxes of cross-correlations: accept a different lag only if it is a tie within tolerance
        return f"superb offsets fs={fs} fv={fv} gathered={gathered} gpu={list(got_offs)} oracle={list(offs)}"
    got = d_out.download()
    if np.max(np.abs(got - want)) > 1e-4 * np.max(np.abs(want)):
        return f"superb signal fs={fs} gathered={gathered} err={np.max(np.abs(got - want)) / np.max(np.abs(want))}"
    return None


def _band_exchange(g, ptrs, n, dtype, op):
    bufs = [np.empty(n, dtype) for _ in ptrs]
    for b, p in zip(bufs, ptrs):
        g._ck(g.lib.tsdrgpu_download(g.h, b.ctypes.data, p, b.nbytes))
    g.sync()
    tot = op(bufs)
    for p in ptrs:
        g._ck(g.lib.tsdrgpu_upload(g.h, p, tot.ctypes.data, tot.nbytes))
    g.sync()


BANDS_RAN = {"batches": 0, "fused": 0, "frames": 0, "relays": 0, "bands": 0}


def fuzz_bands(g, rng):
    h = int(rng.integers(40, 400))
    fs = int(rng.integers(60_000, 3_000_000))
    geo = orc.geometry(fs, h, 60.0)
    w = geo.width
    if w < 8 or w > 3200 or w * h > 500_000:
        return None
    nb = int(rng.integers(1, 6))
    cuts = sorted({32 * int(rng.integers(1, max(2, h // 32 + 1))) for _ in range(nb - 1)})
    cuts = [c for c in cuts if 0 < c < h]
    edges = [0] + cuts + [h]
    rows = [(a, b - a) for a, b in zip(edges[:-1], edges[1:])]
    mb = float(rng.choice([0.0, 0.0, 0.25, 0.9375]))
    opp = orc.PostProcess(geo)
    pps = [gpu.PostProcess(g) for _ in rows]
    for pp in pps:
        pp.set_exact_ties(EXACT_TIES)
    for batch in range(int(rng.integers(1, 4))):
        F = int(rng.integers(1, 6)) if rng.random() < 0.5 else int(rng.integers(8, 13))
        frames = [cases.frame_pattern(w, h, int(rng.integers(0, 50)), rng) for _ in range(F)]
        for fr in frames:
            r = rng.random()
            if r < 0.15:
                fr[rng.integers(0, w * h, 5)] = np.float32(rng.choice([256.0, 1024.0, -300.0]))
            elif r < 0.22:
                fr[:] = np.float32(rng.random())
            elif r < 0.28:
                fr[rng.integers(0, w * h, 3)] = np.float32(-0.0)
            elif r < 0.36:
                yy, xx = np.mgrid[0:h, 0:w]
                px = int(rng.integers(1, 20))
                fr[:] = (((xx // px) % 2) * np.float32(0.55) + np.float32(0.2)).astype(np.float32).reshape(-1)
        want = [opp.run(fr.copy(), mb, 0.1, 0, 0, 0, 0, 0) for fr in frames]
        si, _ = opp.state()
        fr3 = np.stack(frames).reshape(F, h, w)
        fused = rng.random() < 0.65
        d_bands, d_outs, keep = [], [], []
        for (y0, n) in rows:
            d_bands.append(g.to_device(np.ascontiguousarray(fr3[:, y0:y0 + n, :]).reshape(-1)))
            d_outs.append(g.empty(F * w * n))
        if fused:
            xm = []
            for pp, (y0, n), d_b in zip(pps, rows, d_bands):
                band = fr3[:, y0:y0 + n, :]
                ok = ~((band > 250.0) | (band < -250.0))
                mn = np.array([band[f][ok[f]].min() if ok[f].any() else np.inf for f in range(F)], np.float32)
                mx = np.array([band[f][ok[f]].max() if ok[f].any() else -np.inf for f in range(F)], np.float32)
                d_mn, d_mx = g.to_device(mn), g.to_device(mx)
                keep += [d_mn, d_mx]
                xm.append(pp.band_begin_minmax(d_b, F, w, h, y0, n, d_mn.at(0), d_mx.at(0), motionblur=mb))
            _band_exchange(g, [x[0] for x in xm], xm[0][1], np.float32, np.maximum.reduce)
            xs = [pp.band_fused(d_o) for pp, d_o in zip(pps, d_outs)]
            _band_exchange(g, [x[0] for x in xs], xs[0][1], np.float64, lambda b: np.sum(b, axis=0))
        else:
            bg = [pp.band_begin(d_b, F, w, h, y0, n, motionblur=mb) for pp, (y0, n), d_b in zip(pps, rows, d_bands)]
            _band_exchange(g, [b[0] for b in bg], bg[0][1], np.float64, lambda b: np.sum(b, axis=0))
            _band_exchange(g, [b[2] for b in bg], bg[0][3], np.float32, np.maximum.reduce)
        while True:
            res = [pp.band_advance(d_o, k, len(pps)) for k, (pp, d_o) in enumerate(zip(pps, d_outs))]
            if len({r[0] for r in res}) != 1:
                return f"bands: the ranks disagree about a relay fs={fs} h={h} edges={edges} blur={mb}"
            if not res[0][0]:
                infos = res[0][3]
                break
            _band_exchange(g, [r[1] for r in res], res[0][2], np.float64, lambda b: np.sum(b, axis=0))
            BANDS_RAN["relays"] += 1
        BANDS_RAN["batches"] += 1
        BANDS_RAN["fused"] += int(fused)
        BANDS_RAN["frames"] += F
        BANDS_RAN["bands"] += len(rows)
        got = np.concatenate([d_o.download().reshape(F, n, w) for d_o, (_, n) in zip(d_outs, rows)], axis=1).reshape(F, -1)
        tag = f"bands fs={fs} h={h} w={w} edges={edges} blur={mb} batch={batch} F={F} {'fused' if fused else 'two-trip'}"
        for k in range(F):
            if not np.array_equal(got[k], want[k], equal_nan=True):
                return f"{tag}: frame {k} differs in {int(np.sum(got[k] != want[k]))} pixels"
            if not np.array_equal(np.signbit(got[k]), np.signbit(want[k])):
                return f"{tag}: frame {k}: sign of a zero differs"
        i = infos[-1]
        if (i.dx, i.vx, i.stripx, i.dy, i.vy, i.stripy, i.locked) != tuple(si[:7]):
            return f"{tag}: state {(i.dx, i.vx, i.stripx, i.dy, i.vy, i.stripy, i.locked)} oracle {tuple(int(v) for v in si[:7])}"
    for pp in pps:
        pp.destroy()
    return None


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    redzones = None
    if os.environ.get("TSDR_TEST_REDZONES", "0") == "1":  # the GPU suite's red zones (tests/conftest.py) around this script's buffers too
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import conftest as redzones
        redzones._install_redzones()
    g = gpu.TsdrGpu(0)
    fails, ran = [], {"resampler": 0, "postproc": 0, "autocorr": 0, "fft": 0, "plot": 0, "tracking": 0, "superb": 0, "pll": 0, "acmulti": 0, "bands": 0}
    for c in range(ncases):
        kind = ("resampler", "postproc", "postproc", "resampler", "autocorr", "fft", "plot", "tracking", "superb", "pll",
                "acmulti", "bands")[c % 12] if not os.environ.get("FUZZ_ONLY") else os.environ["FUZZ_ONLY"]
        fn = {"resampler": fuzz_resampler, "postproc": fuzz_postproc, "autocorr": fuzz_autocorr, "fft": fuzz_fft,
              "plot": fuzz_plot, "tracking": fuzz_tracking, "superb": fuzz_superb, "pll": fuzz_postproc_pll,
              "acmulti": fuzz_autocorr_multi, "bands": fuzz_bands}[kind]
        try:
            r = fn(g, rng)
        except Exception as e:  # noqa: BLE001
            r = f"{kind} raised {e!r}"
        ran[kind] += 1
        if redzones is not None:
            import gc
            gc.collect()
            for a in list(redzones._redzone_live):
                a.redzones_intact(f"case {c} ({kind})")
            if redzones._redzone_violations:
                r = (r or "") + " RED ZONE: " + "; ".join(redzones._redzone_violations)
                redzones._redzone_violations.clear()
        if r:
            fails.append((c, r))
            print("MISMATCH", c, r, flush=True)
    if ran.get("bands"):
        print(f"bands: {BANDS_RAN['batches']} batches ({BANDS_RAN['fused']} fused) of {BANDS_RAN['frames']} frames in {BANDS_RAN['bands']} band runs, {BANDS_RAN['relays']} relay steps")
    print(f"fuzz: {ncases} cases {ran}, {len(fails)} mismatches, seed {seed}" + (", red zones around every buffer" if redzones is not None else ""))
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
