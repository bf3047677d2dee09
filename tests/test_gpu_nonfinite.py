"""Non-finite input on the HIP path vs the oracle.

The reference's behaviour on NaN / Inf pixels and samples is decided by C comparison semantics, not by design:
`if (val > max) ... else if (val < min)` with min = max = v[0] before any test (dsp.c:50-59: a NaN at pixel 0 poisons the
autogain for good, a NaN elsewhere is invisible to it), the sentinel test `val > 250.0 || val < -250` (an infinity is a
"special colour", a NaN is not), `bestfitcurr > *bestfit` in the sync detector (syncdetector.c:36-55: once the strip total
is not finite no window ever replaces window 0).  Exactly where fminf / fmaxf, wave reductions and tree sums differ.  The
oracle is pinned to the compiled reference on such frames (tests/test_oracle_vs_ref.py::test_post_process_nonfinite_frames);
here the kernels are held to the oracle: frames compared with equal_nan, integer sync state identical, autogain state
identical as bit patterns of "NaN or value".
"""
import numpy as np
import pytest

from tempestsdr_amd import gpu
from gpu_util import ctx
import cases

pytestmark = pytest.mark.gpu

NAN, INF = np.float32(np.nan), np.float32(np.inf)


def _poison(kind, fr, w, h, rng):
    n = w * h
    if kind == "nan_scattered":
        fr[rng.integers(1, n, 5)] = NAN
    elif kind == "nan_first":
        fr[0] = NAN
    elif kind == "nan_last":
        fr[n - 1] = NAN
    elif kind == "nan_all":
        fr[:] = NAN
    elif kind == "nan_column":
        fr.reshape(h, w)[:, w // 3] = NAN
    elif kind == "nan_row":
        fr.reshape(h, w)[h // 2, :] = NAN
    elif kind == "posinf":
        fr[rng.integers(1, n, 3)] = INF
    elif kind == "neginf":
        fr[rng.integers(1, n, 3)] = -INF
    elif kind == "inf_first":
        fr[0] = INF
    elif kind == "neginf_first":
        fr[0] = -INF
    elif kind == "both_inf":
        fr[rng.integers(1, n, 2)] = INF
        fr[rng.integers(1, n, 2)] = -INF
    elif kind == "nan_and_inf":
        fr[rng.integers(1, n, 2)] = NAN
        fr[rng.integers(1, n, 2)] = INF
    elif kind == "huge":  # finite, but the column sums overflow float32
        fr.reshape(h, w)[:, w // 2] = np.float32(200.0)
        fr.reshape(h, w)[:, w // 2 + 1] = np.float32(-249.0)
        fr[5] = np.float32(3e38)
        fr[w + 5] = np.float32(3e38)
    else:
        raise ValueError(kind)


KINDS = ["nan_scattered", "nan_first", "nan_last", "nan_all", "nan_column", "nan_row", "posinf", "neginf", "inf_first", "neginf_first",
         "both_inf", "nan_and_inf", "huge"]
ORDERS = [(0, 0, 0, 0.0), (0, 0, 0, 0.75), (1, 0, 1, 0.5), (0, 1, 0, 0.25), (1, 1, 1, 0.9), (0, 0, 1, 0.0), (1, 0, 0, 0.0)]


def _sequence(kind, w, h, seed, F=7, bad=(2, 3)):
    rng = np.random.default_rng(seed)
    frames = [cases.frame_pattern(w, h, 3 * k, rng) for k in range(F)]
    for k in bad:
        _poison(kind, frames[k], w, h, rng)
    return frames


def _oracle(orc, frames, w, h, lbs, aap, ash, mb):
    geo = orc.geometry(1, h, 1.0)
    geo.width = w
    opp = orc.PostProcess(geo)
    outs, states = [], []
    for fr in frames:
        outs.append(opp.run(fr.copy(), mb, 0.1, lbs, aap, ash, 0, 0).copy())
        states.append(opp.state())
    return np.stack(outs), states


def _same_f32(a, b):
    a, b = np.float32(a), np.float32(b)
    return (np.isnan(a) and np.isnan(b)) or a == b


def _check(got, infos, want, states, tag):
    for k, (info, (si, sd)) in enumerate(zip(infos, states)):
        assert (info.dx, info.vx, info.stripx, info.dy, info.vy, info.stripy) == tuple(si[:6]), (tag, k, "sync state")
        assert _same_f32(info.lastmin, sd[0]) and _same_f32(info.lastmax, sd[1]), (tag, k, "autogain", info.lastmin, info.lastmax, sd[:2])
    for k in range(len(want)):
        bad = ~((got[k] == want[k]) | (np.isnan(got[k]) & np.isnan(want[k])))
        assert not bad.any(), (tag, k, int(bad.sum()), np.flatnonzero(bad)[:5], got[k][bad][:5], want[k][bad][:5])


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("order", ORDERS)
@pytest.mark.parametrize("w,h", [(40, 33), (333, 131)])
def test_post_process_run_with_nonfinite_pixels(orc, kind, order, w, h):
    """tsdrgpu_postproc_run, every stage order, frame by frame and as one batch: two poisoned frames between clean ones (the
    frames AFTER carry the damage: the autogain's IIR, the IIR screen buffer, the sync state)."""
    g = ctx()
    lbs, aap, ash, mb = order
    frames = _sequence(kind, w, h, 1000 + w)
    want, states = _oracle(orc, frames, w, h, lbs, aap, ash, mb)
    n, F = w * h, len(frames)
    d_in = g.to_device(np.concatenate(frames))
    for batch in (1, F):
        pp = gpu.PostProcess(g)
        d_out = g.empty(F * n)
        infos = []
        for s in range(0, F, batch):
            infos += pp.run(d_in, min(batch, F - s), w, h, d_out, mb, 0.1, lbs, aap, ash, 0, 0, frames_offset=s * n, out_offset=s * n)
        _check(d_out.download().reshape(F, n), infos, want, states, (kind, order, batch))


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("order", [(0, 0, 0, 0.0), (0, 0, 0, 0.75), (1, 0, 1, 0.5), (1, 1, 0, 0.25)])
def test_split_run_with_nonfinite_pixels(orc, kind, order):
    """tsdrgpu_postproc_begin / _finish (the chain on the side lane)"""
    g = ctx()
    lbs, aap, ash, mb = order
    w, h = 333, 131
    frames = _sequence(kind, w, h, 2000)
    want, states = _oracle(orc, frames, w, h, lbs, aap, ash, mb)
    n, F = w * h, len(frames)
    d_in = g.to_device(np.concatenate(frames))
    pp = gpu.PostProcess(g)
    d_out = g.empty(F * n)
    infos = []
    for s, k in ((0, 4), (4, F - 4)):
        pp.begin(d_in, k, w, h, mb, 0.1, lbs, aap, ash, 0, 0, frames_offset=s * n)
        infos += pp.finish(d_out, out_offset=s * n)
    _check(d_out.download().reshape(F, n), infos, want, states, (kind, order))


def _autogain_minmax(frames):
    """what the resampler's frame tracking hands the fused run: min / max over the pixels dsp_autogain_run's first pass does not skip
    (dsp.c:57) and that take part in its comparisons — a NaN compares false either way, an infinity is skipped as a special colour"""
    mn, mx = [], []
    for fr in frames:
        ok = fr[np.abs(fr) <= 250.0]  # (False for NaN)
        mn.append(ok.min() if ok.size else INF)
        mx.append(ok.max() if ok.size else -INF)
    return np.array(mn, np.float32), np.array(mx, np.float32)


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("mb", [0.0, 0.75])
def test_fused_run_with_nonfinite_pixels(orc, kind, mb):
    """tsdrgpu_postproc_begin_minmax / _finish: the flat trip (motion blur 0) and the tile-walking one (motion blur > 0)"""
    g = ctx()
    w, h = 333, 131
    frames = _sequence(kind, w, h, 3000)
    want, states = _oracle(orc, frames, w, h, 0, 0, 0, mb)
    n, F = w * h, len(frames)
    d_in = g.to_device(np.concatenate(frames))
    mn, mx = _autogain_minmax(frames)
    d_mn, d_mx = g.to_device(mn), g.to_device(mx)
    pp = gpu.PostProcess(g)
    d_out = g.empty(F * n)
    infos = []
    for s, k in ((0, 4), (4, F - 4)):
        pp.begin_minmax(d_in, k, w, h, d_mn.at(s), d_mx.at(s), d_out, mb, 0.1, 0, 0, 0, 0, 0, frames_offset=s * n, out_offset=s * n)
        infos += pp.finish(d_out, out_offset=s * n)
    _check(d_out.download().reshape(F, n), infos, want, states, (kind, mb))


@pytest.mark.parametrize("kind", ["nan_scattered", "nan_first", "posinf", "neginf", "nan_run"])
@pytest.mark.parametrize("from_iq", [0, 1])
@pytest.mark.parametrize("r", [1.99935, 0.37, 3.25])
def test_resampler_with_nonfinite_samples(orc, kind, from_iq, r):
    """a1 + a2: NaN / Inf samples through the fused demodulation + area resampler, state carried over calls: the pixels a bad
    sample touches (and the carried `contrib`) are the oracle's, NaN for NaN"""
    g = ctx()
    chunk, nch = 1013, 6
    rng = np.random.default_rng(int(r * 100) + from_iq)
    n = chunk * nch
    rs = gpu.Resampler(g)
    ors = orc.Resampler()
    for call in range(3):
        if from_iq:
            iq = (rng.random(2 * n) - 0.5).astype(np.float32)
        else:
            iq = rng.random(n).astype(np.float32)
        per = 2 if from_iq else 1
        if call != 1:
            if kind == "nan_scattered":
                iq[rng.integers(per, per * n, 4)] = NAN
            elif kind == "nan_first":
                iq[0] = NAN
            elif kind == "posinf":
                iq[rng.integers(per, per * n, 2)] = INF
            elif kind == "neginf":
                iq[rng.integers(per, per * n, 2)] = -INF
            elif kind == "nan_run":
                iq[per * (chunk - 3):per * (chunk + 3)] = NAN  # across a chunk boundary
        mag = orc.am_demod(iq) if from_iq else iq
        want = np.concatenate([ors.process(mag[c * chunk:(c + 1) * chunk], r, 1.0) for c in range(nch)])
        d_in = g.to_device(iq)
        d_out = g.empty(int(n * r) + 64)
        got_n = rs.process(d_in, from_iq, chunk, nch, r, 1.0, 0, d_out)
        assert got_n == want.size
        got = d_out.download()[:got_n]
        assert np.array_equal(got, want, equal_nan=True), (call, int(np.sum(~((got == want) | (np.isnan(got) & np.isnan(want))))))
        c, o = rs.state()
        assert (np.isnan(c) and np.isnan(ors.st.contrib)) or c == ors.st.contrib, (c, ors.st.contrib)
        assert o == ors.st.offset


@pytest.mark.parametrize("kind", ["nan_one", "inf_one", "nan_tail"])
@pytest.mark.parametrize("exact", [0, 1])
def test_autocorr_window_with_nonfinite_samples(orc, kind, exact):
    """a9-a11: one capture window holding a NaN / an infinity.  Inside the transformed part the reference's FFT spreads it over the
    whole window (every lag NaN); behind the power-of-two prefix it is ignored (fft.c:5-11,49-64).  Exact mode: NaN for NaN, the rest
    bit-identical; float32 plan: the same lags are NaN and the argmax is the reference's (index 0: `>` never beats a NaN at lag 0)."""
    g = ctx()
    fs = 300_000
    ac = gpu.Autocorr(g, fs)
    ac.set_exact(bool(exact))
    rng = np.random.default_rng(5)
    sig = (0.3 + 0.5 * ((np.arange(ac.capture) // 37) % 2) + 0.05 * rng.random(ac.capture)).astype(np.float32)
    if kind == "nan_one":
        sig[ac.n // 3] = NAN
    elif kind == "inf_one":
        sig[ac.n // 5] = INF
    else:
        assert ac.capture > ac.n
        sig[ac.n + 1:] = NAN  # ignored by the transform
    oac = orc.Autocorr(fs)
    oac.run(sig)
    ac.run(g.to_device(sig), False, ac.capture, 1)
    f, l, _ = ac.plots()
    assert np.array_equal(np.isnan(f), np.isnan(oac.frame)) and np.array_equal(np.isnan(l), np.isnan(oac.line))
    if kind == "nan_tail":
        assert not np.isnan(f).any()
        assert np.max(np.abs(f - oac.frame)) <= 1e-4 * np.max(oac.frame)
    else:
        assert np.isnan(f).all() and np.isnan(l).all()
    if exact:
        assert np.array_equal(f, oac.frame, equal_nan=True) and np.array_equal(l, oac.line, equal_nan=True)
    fi, li = ac.argmax()
    assert (fi, li) == (int(np.argmax(oac.frame)) if not np.isnan(oac.frame).any() else 0,
                        int(np.argmax(oac.line)) if not np.isnan(oac.line).any() else 0)


@pytest.mark.parametrize("kind", ["all_nan", "nan_first", "nan_scattered", "nan_column_starts", "inf"])
@pytest.mark.parametrize("size,nwidth", [(50000, 800), (2315, 640), (300, 800)])
def test_plot_columns_with_nonfinite_values(orc, kind, size, nwidth):
    """f4: PlotVisualizer.populateData starts every maximum from a first value and moves on a LARGER one only
    (PlotVisualizer.java:203-239): a NaN at lag 0 pins max_index, lowest and highest; a NaN at the first lag of a pixel column
    is that column's value; a NaN anywhere else is never taken.  Device columns == the oracle's loop, NaN for NaN."""
    g = ctx()
    rng = np.random.default_rng(size + nwidth)
    data = rng.random(size) + 0.2 * np.sin(np.arange(size) / 37.0)
    if kind == "all_nan":
        data[:] = np.nan
    elif kind == "nan_first":
        data[0] = np.nan
    elif kind == "nan_scattered":
        data[rng.integers(1, size, 40)] = np.nan
    elif kind == "nan_column_starts":
        per = max(1, size // nwidth)
        data[per * rng.integers(1, max(2, size // per), 25)] = np.nan
    else:
        data[rng.integers(1, size, 3)] = np.inf
        data[rng.integers(1, size, 3)] = -np.inf
    d = g.empty(2 * size, np.float32)
    g._ck(g.lib.tsdrgpu_upload(g.h, d.ptr, data.ctypes.data, data.nbytes))
    g.sync()
    want = orc.plot_populate(data, nwidth)
    got = g.plot_columns(d.ptr, size, nwidth)
    assert np.array_equal(got[0], want[0], equal_nan=True)
    for a, b in zip(got[1:3], want[1:3]):
        assert a == b or (np.isnan(a) and np.isnan(b))
    assert got[3] == want[3]


@pytest.mark.parametrize("kind", ["nan_scattered", "nan_first", "nan_all", "nan_row", "posinf", "inf_first", "both_inf"])
@pytest.mark.parametrize("lbs,aap,autoshift", [(0, 0, 0), (1, 0, 1), (1, 1, 0)])
def test_band_run_with_nonfinite_pixels(orc, kind, lbs, aap, autoshift, monkeypatch):
    """SURVEY 8(e) row 2 on poisoned frames: three row bands, the exchanges done on the host the way RCCL does them — ncclMax
    drops a NaN like fmaxf, so the NaN at pixel 0 that poisons the reference's autogain (dsp.c:50-59) must travel as a flag,
    not as a value — frames and per-frame records equal the ORACLE's."""
    import test_gpu_bands as tb
    g = ctx()
    W, H, blur = 333, 160, 0.5
    edges = [0, 64, 96, H]
    frames = _sequence(kind, W, H, 4000 + lbs)
    fr = np.stack(frames).reshape(len(frames), H, W)
    want, states = _oracle(orc, frames, W, H, lbs, aap, autoshift, blur)

    plain = tb._host_collective

    def rccl_like(g_, kind_, ptrs, count):
        if kind_ != gpu.BAND_MAX_F32:
            return plain(g_, kind_, ptrs, count)
        bufs = [np.empty(count, np.float32) for _ in ptrs]
        for b, p in zip(bufs, ptrs):
            g_._ck(g_.lib.tsdrgpu_download(g_.h, b.ctypes.data, p, b.nbytes))
        g_.sync()
        tot = np.fmax.reduce(bufs)  # NaN-dropping, like ncclMax on floats
        for p in ptrs:
            g_._ck(g_.lib.tsdrgpu_upload(g_.h, p, tot.ctypes.data, tot.nbytes))
        g_.sync()

    monkeypatch.setattr(tb, "_host_collective", rccl_like)
    pps = [gpu.PostProcess(g) for _ in range(3)]
    got, infos, kinds = tb._run_bands_general(g, pps, edges, fr, motionblur=blur, lowpass_before_sync=lbs, autogain_after_proc=aap, autoshift=autoshift)
    assert gpu.BAND_MAX_F32 in kinds
    _check(got.reshape(len(frames), -1), infos, want, states, (kind, lbs, aap, autoshift))
