"""Golden-vector cases for the hot path.  Inputs are built from numpy's PCG64
integer stream and plain IEEE arithmetic only (no libm), so they are
bit-identical on every machine; outputs come from a backend:

  backend "ref": the REAL reference (oracle/_ref/libtsdr_ref.so) - used by
                 make_golden.py in the build container to write golden.npz
  backend "orc": the CPU restatement (oracle/liboracle.so) - used by
                 tests/test_oracle_golden.py everywhere

Large outputs are stored as SHA-256 digests (bit-exact pin) plus a few full
arrays that the GPU tests use as fixtures.
"""
import ctypes as C
import hashlib

import numpy as np


def sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8).copy()


def frame_pattern(w, h, k, rng):
    """A frame with a blanking band that drifts by (3,1) pixels per frame."""
    x = (np.arange(w)[None, :] + 3 * k) % w
    y = (np.arange(h)[:, None] + k) % h
    img = np.where((x * 8 // w) % 2 == 0, np.float32(0.3), np.float32(0.8)).astype(np.float32)
    img = img + np.where(y >= h // 2, ((x // 8 + y // 8) % 2).astype(np.float32) * np.float32(0.2), np.float32(0))
    img = np.where((x >= (w * 4) // 5) | (y >= (h * 9) // 10), np.float32(0.05), img)
    noise = rng.random(w * h, dtype=np.float32).reshape(h, w) * np.float32(0.04)
    return (img + noise).astype(np.float32).reshape(-1)


RESAMPLE = dict(fs=2_000_000, h=131, fv=60.0, chunks=4)
PP = dict(fs=400_000, h=61, fv=60.0, frames=8)
PP_CFGS = [  # (lbs, aap, autoshift, pll, motionblur)
    (0, 0, 0, 0, 0.0), (0, 0, 1, 0, 0.75), (1, 0, 0, 0, 0.0), (1, 0, 1, 1, 0.5),
    (0, 1, 0, 0, 0.0), (0, 1, 0, 1, 0.9375), (1, 1, 0, 0, 0.25), (1, 1, 1, 0, 0.0)]
FFT_N = 4096
AC = dict(fs=300_000, windows=3)
SUPERB = dict(fs=120_000, fv=60.0)


def run_cases(backend, orc):
    out = {}
    rng = np.random.default_rng(20260923)
    ref = orc.ref() if backend == "ref" else None

    # ---- geometry ---------------------------------------------------------
    geoms = [(8_000_000, 525, 60.0), (25_000_000, 806, 60.0), (100_000_000, 1125, 60.0),
             (200_000_000, 2250, 60.0), (12_600_000, 525, 60.0)]
    gw = []
    for fs, h, fv in geoms:
        if ref:
            t = ref.ref_new(h, fv, fs, 0.0, None)
            gw.append([ref.ref_width(t), ref.ref_pixelrate(t), ref.ref_pixeltimeoversampletime(t)])
            ref.ref_free(t)
        else:
            g = orc.geometry(fs, h, fv)
            gw.append([g.width, g.pixelrate, g.pixeltimeoversampletime])
    out["geometry"] = np.array(gw, np.float64)

    # ---- demod --------------------------------------------------------------
    iq = (rng.random(2 * 5000, dtype=np.float32) - np.float32(0.5)) * np.float32(3)
    out["demod_in"] = iq
    if ref:
        d = iq.copy()
        ref.complex_to_real(d, 5000)
        out["demod_out"] = d[:5000].copy()
    else:
        out["demod_out"] = orc.am_demod(iq)

    # ---- resampler, area + nearest, 4 consecutive chunks --------------------
    g = orc.geometry(RESAMPLE["fs"], RESAMPLE["h"], RESAMPLE["fv"])
    up, down = g.width * g.height * g.refreshrate, float(RESAMPLE["fs"])
    chunk = orc.chunk_size(RESAMPLE["fs"], RESAMPLE["fv"])
    x = rng.random(RESAMPLE["chunks"] * chunk, dtype=np.float32)
    out["resample_in"] = x
    for nearest in (0, 1):
        outs, states = [], []
        if ref:
            r = ref.ref_resampler_new()
            st = np.zeros(2)
            for c in range(RESAMPLE["chunks"]):
                buf = np.zeros(3 * chunk, np.float32)
                n = ref.ref_resampler_process(r, x[c * chunk:(c + 1) * chunk], chunk, buf, up, down, nearest)
                outs.append(buf[:n].copy())
                ref.ref_resampler_state(r, st)
                states.append(st.copy())
            ref.ref_resampler_free(r)
        else:
            r = orc.Resampler()
            for c in range(RESAMPLE["chunks"]):
                outs.append(r.process(x[c * chunk:(c + 1) * chunk], up, down, nearest))
                states.append(np.array([r.st.contrib, r.st.offset]))
        out[f"resample_out_{nearest}"] = np.concatenate(outs)
        out[f"resample_counts_{nearest}"] = np.array([o.size for o in outs], np.int64)
        out[f"resample_state_{nearest}"] = np.array(states)

    # ---- post-processing, 8 parameter sets x 8 frames ------------------------
    fs, h, fv = PP["fs"], PP["h"], PP["fv"]
    frames_rng = np.random.default_rng(99)
    g0 = orc.geometry(fs, h, fv)
    frames = [frame_pattern(g0.width, h, k, frames_rng) for k in range(PP["frames"])]
    out["pp_frames"] = np.stack(frames)
    for ci, (lbs, aap, ash, pll, mb) in enumerate(PP_CFGS):
        digests, si_all, sd_all, rr = [], [], [], []
        si, sd = np.zeros(10, np.int32), np.zeros(4)
        if ref:
            t = ref.ref_new(h, fv, fs, mb, None)
            ref.ref_setparam(t, 0, ash)
            ref.ref_setparam(t, 1, pll)
        else:
            g = orc.geometry(fs, h, fv)
            pp = orc.PostProcess(g)
        first = None
        for k in range(PP["frames"]):
            fr = frames[k].copy()
            if ref:
                if ref.ref_width(t) != g0.width:  # PLL moved the width: stop this trace
                    break
                p = ref.ref_post_process(t, fr, mb, 0.1, lbs, aap)
                res = np.ctypeslib.as_array(p, shape=(fr.size,)).copy()
                ref.ref_postprocess_state(t, si, sd)
                rate = ref.ref_refreshrate(t)
            else:
                if g.width != g0.width:
                    break
                res = pp.run(fr, mb, 0.1, lbs, aap, ash, pll, 0)
                si, sd = pp.state()
                rate = g.refreshrate
            if first is None:
                first = res
            last = res
            digests.append(sha(res))
            row = si.copy()
            row[7] = row[8] = 0
            si_all.append(row)
            sd_all.append(sd.copy())
            rr.append(rate)
        if ref:
            ref.ref_free(t)
        out[f"pp{ci}_sha"] = np.stack(digests)
        out[f"pp{ci}_int"] = np.stack(si_all)
        out[f"pp{ci}_dbl"] = np.stack(sd_all)
        out[f"pp{ci}_rate"] = np.array(rr)
        out[f"pp{ci}_first"] = first
        out[f"pp{ci}_last"] = last

    # ---- FFT, autocorrelation, accumulate --------------------------------------
    z = rng.random(2 * FFT_N, dtype=np.float32) - np.float32(0.5)
    out["fft_in"] = z
    for inv in (0, 1):
        if ref:
            b = z.copy()
            ref.fft_perform(b, FFT_N, inv)
        else:
            b = orc.fft_perform(z, inv)
        out[f"fft_out_{inv}"] = b
    fs = AC["fs"]
    size = orc.capture_size(fs)
    flo, flen, llo, llen = orc.lag_windows(fs)
    xs = [rng.random(size, dtype=np.float32) * np.float32(1 + 0.25 * k) for k in range(AC["windows"])]
    # give the windows a periodic component so the plots have a peak
    per = 4999
    for xk in xs:
        xk += (np.arange(size) % per < 300).astype(np.float32)
    out["ac_in"] = np.stack(xs)
    fr, ln = np.zeros(flen), np.zeros(llen)
    if ref:
        for k, xk in enumerate(xs):
            corr = np.zeros(2 * size, np.float32)
            ref.fft_autocorrelation(corr, xk, size)
            ref.ref_accumulate(fr, corr, flo, flen, k + 1)
            ref.ref_accumulate(ln, corr, llo, llen, k + 1)
    else:
        ac = orc.Autocorr(fs)
        for xk in xs:
            corr = ac.run(xk)
        fr, ln = ac.frame, ac.line
    out["ac_corr_last_sha"] = sha(corr)
    out["ac_corr_last_head"] = corr[:4096].copy()
    out["ac_frame_plot"] = fr.copy()
    out["ac_line_plot"] = ln.copy()

    # ---- super-bandwidth stitch -------------------------------------------------
    fs, fv = SUPERB["fs"], SUPERB["fv"]
    sif = int(fs / fv)
    gathered = 10 * sif
    base = rng.random(2 * gathered + 2000, dtype=np.float32) - np.float32(0.5)
    hops = []
    for i in range(4):
        s = 2 * (53 * i)
        hp = base[s:s + 2 * gathered] + (rng.random(2 * gathered, dtype=np.float32) - np.float32(0.5)) * np.float32(0.1)
        hops.append(hp.astype(np.float32))
    out["superb_hops"] = np.stack(hops)
    if ref:
        t = ref.ref_new(100, fv, fs, 0.0, None)
        rh = [hh.copy() for hh in hops]
        ptrs = (C.c_void_p * 4)(*[hh.ctypes.data for hh in rh])
        per_n = ref.fft_getrealsize(gathered)
        res = np.zeros(4 * per_n * 2, np.float32)
        n = ref.ref_superb_stitch(t, ptrs, 4, gathered, sif, fs, res)
        ref.ref_free(t)
        res = res[:2 * n]
    else:
        res, _ = orc.superb_stitch(hops, sif)
    out["superb_out_sha"] = sha(res)
    out["superb_out_head"] = res[:2048].copy()
    return out
