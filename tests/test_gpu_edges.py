"""Edge cases through the C ABI: empty and ragged inputs, odd geometries, size
changes between calls, maximum sizes, many-frame batches."""
import ctypes as C

import numpy as np
import pytest

from tempestsdr_amd import gpu
from gpu_util import ctx
import cases

pytestmark = pytest.mark.gpu
RNG = np.random.default_rng(21)


def test_empty_calls_are_noops(orc):
    g = ctx()
    d = g.empty(16)
    g.am_demod(d, d, 0)
    rs = gpu.Resampler(g)
    assert rs.process(d, 0, 7, 0, 2.0, 1.0, 0, d) == 0
    assert rs.state() == (0.0, 0.0)
    pp = gpu.PostProcess(g)
    assert pp.run(d, 0, 4, 4, d) == []
    ac = gpu.Autocorr(g, 300_000)
    ac.run(d, 0, ac.capture, 0)
    f, l, calls = ac.plots()
    assert calls == 0 and not f.any() and not l.any()
    g.decode_samples(d, "int16", d, 0)
    g.frame_to_rgb(d, g.empty(16, np.int32), 0)


def test_bad_arguments_are_rejected(orc):
    g = ctx()
    d = g.empty(64)
    with pytest.raises(gpu.TsdrGpuError):
        g.fft_perform(d, 24, 0)  # not a power of two
    rs = gpu.Resampler(g)
    with pytest.raises(gpu.TsdrGpuError):
        rs.process(d, 0, 16, 2, 4.0, 1.0, 0, d)  # output buffer too small (needs 128)
    with pytest.raises(gpu.TsdrGpuError):
        rs.process(d, 0, 0, 1, 2.0, 1.0, 0, d)  # chunk of zero samples
    with pytest.raises(gpu.TsdrGpuError):
        gpu.Autocorr(g, 100)  # lag windows empty at this rate
    pp = gpu.PostProcess(g)
    with pytest.raises(gpu.TsdrGpuError):
        pp.run(d, 1, 0, 4, d)


@pytest.mark.parametrize("w,h", [(5, 3), (7, 129), (257, 33), (1033, 806), (64, 64), (4000, 11)])
@pytest.mark.parametrize("cfg", [(0, 0, 0, 0.5), (1, 0, 1, 0.0), (0, 1, 0, 0.25)])
def test_ragged_frame_geometries(orc, w, h, cfg):
    """widths/heights that are not multiples of any tile, vector width or wave size"""
    g = ctx()
    lbs, aap, ash, mb = cfg
    rng = np.random.default_rng(w * 1000 + h)
    frames = [cases.frame_pattern(w, h, k, rng) for k in range(5)]
    geo = orc.geometry(1, h, 1.0)  # only width/height are used by the post-processing oracle
    geo.width = w
    opp = orc.PostProcess(geo)
    pp = gpu.PostProcess(g)
    d_in = g.to_device(np.concatenate(frames))
    d_out = g.empty(5 * w * h)
    infos = pp.run(d_in, 5, w, h, d_out, mb, 0.1, lbs, aap, ash, 0, 0)
    got = d_out.download().reshape(5, -1)
    for k, fr in enumerate(frames):
        want = opp.run(fr.copy(), mb, 0.1, lbs, aap, ash, 0, 0)
        si, sd = opp.state()
        assert (infos[k].dx, infos[k].stripx, infos[k].dy, infos[k].stripy) == (si[0], si[2], si[3], si[5]), (k, w, h)
        assert np.array_equal(got[k], want), (k, w, h)


@pytest.mark.parametrize("w,h", [(1, 1), (2, 1), (1, 2), (3, 1), (1, 7), (2, 2), (3, 2), (2, 3), (4, 3), (5, 2), (1, 300), (300, 1), (2, 4097), (4099, 2), (4099, 1)])
@pytest.mark.parametrize("cfg", [(0, 0, 0, 0.5), (1, 0, 1, 0.0), (0, 1, 0, 0.25), (1, 1, 1, 0.9)])
def test_degenerate_frame_geometries(orc, w, h, cfg):
    """the smallest rasters: what the library is handed when a host sets a resolution the stream cannot fill (the reference
    accepts any height > 0, TSDRLibrary.c:552-565, and derives the width by truncation) — one pixel, one row, one column
    included: the kernels equal the oracle bit for bit in every stage order, through the plain and the split run (the compiled
    reference and the oracle agree at these sizes, tests/test_oracle_vs_ref.py::test_post_process_degenerate_geometries).
    A strip of ONE entry makes every window fit 0/0 (syncdetector.c:26-58 divides by n - size): the reference keeps window 0."""
    g = ctx()
    lbs, aap, ash, mb = cfg
    rng = np.random.default_rng(w * 7919 + h)
    F = 9  # (>= 8: the frame-parallel pass too)
    frames = [(rng.random(w * h) * 2 - 0.5).astype(np.float32) for _ in range(F)]
    geo = orc.geometry(1, h, 1.0)
    geo.width = w
    opp = orc.PostProcess(geo)
    want, states = [], []
    for fr in frames:
        want.append(opp.run(fr.copy(), mb, 0.1, lbs, aap, ash, 0, 0))
        states.append(opp.state()[0].copy())
    d_in = g.to_device(np.concatenate(frames))
    for form in ("run", "split", "one_by_one"):
        pp = gpu.PostProcess(g)
        d_out = g.empty(F * w * h)
        if form == "run":
            infos = pp.run(d_in, F, w, h, d_out, mb, 0.1, lbs, aap, ash, 0, 0)
        elif form == "split":
            pp.begin(d_in, F, w, h, mb, 0.1, lbs, aap, ash, 0, 0)
            infos = pp.finish(d_out)
        else:
            infos = []
            for k in range(F):
                infos += pp.run(d_in, 1, w, h, d_out, mb, 0.1, lbs, aap, ash, 0, 0, frames_offset=k * w * h, out_offset=k * w * h)
        got = d_out.download().reshape(F, -1)
        for k in range(F):
            si = states[k]
            assert (infos[k].dx, infos[k].stripx, infos[k].dy, infos[k].stripy) == (si[0], si[2], si[3], si[5]), (form, k, w, h)
            assert np.array_equal(got[k], want[k], equal_nan=True), (form, k, w, h)


@pytest.mark.parametrize("w,h", [(16385, 2), (2, 16385), (20000, 3), (3, 20011), (33333, 100), (100, 33333), (100003, 1), (1, 100003), (1_000_003, 2)])
@pytest.mark.parametrize("cfg", [(0, 0, 0, 0.5), (1, 0, 1, 0.0), (1, 1, 1, 0.9)])
def test_strips_longer_than_the_lds(orc, w, h, cfg):
    """The reference bounds width x height (4000 x 4000, TSDRLibrary.c:31,489), not each: 100 MS/s with a 100-line raster at 60 Hz is
    33 333 pixels per line.  A strip of more than 16 384 entries is blurred and scanned in HBM instead of LDS
    (k_strip_prepare<true>): same frames, same sync state, every stage order."""
    g = ctx()
    lbs, aap, ash, mb = cfg
    rng = np.random.default_rng(w + 31 * h)
    F = 3
    y, x = np.mgrid[0:h, 0:w]
    frames = []
    for k in range(F):
        img = 0.3 + 0.5 * (((x + 5 * k) // max(2, w // 9)) % 2) + 0.1 * ((y // max(1, h // 5)) % 2)
        img[:, : max(1, w // 11)] = 0.05
        img[: max(1, h // 20), :] = 0.05
        frames.append((img + rng.standard_normal((h, w)) * 0.02).astype(np.float32).reshape(-1))
    geo = orc.geometry(1, h, 1.0)
    geo.width = w
    opp = orc.PostProcess(geo)
    pp = gpu.PostProcess(g)
    d_in = g.to_device(np.concatenate(frames))
    d_out = g.empty(F * w * h)
    infos = pp.run(d_in, F, w, h, d_out, mb, 0.1, lbs, aap, ash, 0, 0)
    got = d_out.download().reshape(F, -1)
    for k, fr in enumerate(frames):
        want = opp.run(fr.copy(), mb, 0.1, lbs, aap, ash, 0, 0)
        si, sd = opp.state()
        assert (infos[k].dx, infos[k].vx, infos[k].stripx, infos[k].dy, infos[k].vy, infos[k].stripy) == tuple(si[:6]), (k, w, h)
        assert np.array_equal(got[k], want), (k, w, h, int(np.sum(got[k] != want)))


def test_the_longest_strip_and_the_pixel_bound(orc):
    """16 000 000 x 1: the longest strip the reference's own bound admits; one pixel more is refused (the reference's tsdr_readasync
    refuses it too, TSDRLibrary.c:489)."""
    g = ctx()
    w, h = 16_000_000, 1
    rng = np.random.default_rng(77)
    fr = (0.3 + 0.5 * ((np.arange(w) // 1_000_000) % 2) + 0.02 * rng.standard_normal(w)).astype(np.float32)
    fr[: w // 11] = 0.05
    geo = orc.geometry(1, h, 1.0)
    geo.width = w
    opp = orc.PostProcess(geo)
    want = opp.run(fr.copy(), 0.25, 0.1, 0, 0, 0, 0, 0)
    si, _ = opp.state()
    pp = gpu.PostProcess(g)
    d_in = g.to_device(fr)
    d_out = g.empty(w)
    info = pp.run(d_in, 1, w, h, d_out, 0.25)[0]
    assert (info.dx, info.stripx, info.dy, info.stripy) == (si[0], si[2], si[3], si[5])
    assert np.array_equal(d_out.download(), want)
    with pytest.raises(gpu.TsdrGpuError, match="4000"):
        pp.run(d_in, 1, 4001, 4000, d_out, 0.25)


def test_resolution_change_between_calls(orc):
    """dsp_post_process keeps autogain/sync state across a size change and zeroes the screen
    buffer only when it has to grow (dsp.c:152-173)."""
    g = ctx()
    geo = orc.geometry(1, 1, 1.0)
    opp = orc.PostProcess(geo)
    pp = gpu.PostProcess(g)
    rng = np.random.default_rng(5)
    for (w, h) in [(200, 131), (120, 90), (260, 140), (200, 131)]:
        geo.width, geo.height = w, h
        for k in range(3):
            fr = cases.frame_pattern(w, h, k, rng)
            want = opp.run(fr.copy(), 0.5, 0.1, 0, 0, 0, 0, 0)
            d_in = g.to_device(fr)
            d_out = g.empty(w * h)
            info = pp.run(d_in, 1, w, h, d_out, 0.5)[0]
            si, sd = opp.state()
            assert (info.dx, info.dy) == (si[0], si[3])
            assert np.array_equal(d_out.download(), want), (w, h, k)


def test_many_frames_one_batch(orc):
    g = ctx()
    fs, h, fv = 2_000_000, 131, 60.0
    geo = orc.geometry(fs, h, fv)
    w = geo.width
    rng = np.random.default_rng(8)
    F = 150
    frames = [cases.frame_pattern(w, h, k, rng) for k in range(F)]
    opp = orc.PostProcess(geo)
    pp = gpu.PostProcess(g)
    d_in = g.to_device(np.concatenate(frames))
    d_out = g.empty(F * w * h)
    infos = pp.run(d_in, F, w, h, d_out, 0.9375)
    got = d_out.download().reshape(F, -1)
    for k in range(F):
        want = opp.run(frames[k].copy(), 0.9375)
        if k % 10 == 0 or k == F - 1:
            assert np.array_equal(got[k], want), k
    si, sd = opp.state()
    assert (infos[-1].dx, infos[-1].dy, infos[-1].stripx, infos[-1].stripy) == (si[0], si[3], si[2], si[5])


def test_maximum_frame_size(orc):
    """MAX_ARR_SIZE = 4000*4000 pixels (TSDRLibrary.c:31): one frame through the default order."""
    g = ctx()
    w, h = 4000, 4000
    rng = np.random.default_rng(9)
    fr = cases.frame_pattern(w, h, 3, rng)
    geo = orc.geometry(1, h, 1.0)
    geo.width = w
    want = orc.PostProcess(geo).run(fr.copy(), 0.0)
    pp = gpu.PostProcess(g)
    d_in = g.to_device(fr)
    d_out = g.empty(w * h)
    pp.run(d_in, 1, w, h, d_out, 0.0)
    assert np.array_equal(d_out.download(), want)


@pytest.mark.parametrize("fs", [100_000, 1_000_000, 8_000_000, 25_000_000, 200_000_000])
def test_autocorr_geometry_all_rates(orc, fs):
    g = ctx()
    ac = gpu.Autocorr(g, fs)
    assert (ac.flo, ac.flen, ac.llo, ac.llen) == orc.lag_windows(fs)
    assert ac.capture == orc.capture_size(fs)
    assert ac.n == orc.lib.orc_fft_getrealsize(ac.capture)


def test_autocorr_200Msps_window(orc):
    """BASELINE config 5's window: N = 2^23."""
    g = ctx()
    fs = 200_000_000
    ac = gpu.Autocorr(g, fs)
    assert ac.n == 1 << 23
    period = fs // 60
    x = RNG.random(ac.capture).astype(np.float32) * np.float32(0.3)
    x += (np.arange(ac.capture) % period < period // 10).astype(np.float32)
    ac.run(g.to_device(x), 0, ac.capture, 1)
    f, l, _ = ac.plots()
    fi, li = ac.argmax()
    assert abs((ac.flo + fi) - period) <= 16  # a noisy pulse train: the triangle's top is a few lags wide
    oac = orc.Autocorr(fs)
    oac.run(x)
    assert np.max(np.abs(f - oac.frame)) <= 1e-4 * np.max(oac.frame)
    assert oac.frame[fi] >= np.max(oac.frame) * (1 - 2e-4)


def test_resampler_tiny_and_huge_chunks(orc):
    g = ctx()
    for chunk, nch, r in [(1, 50, 2.0), (2, 33, 1.999), (3, 1000, 0.7), (700_001, 2, 1.99935)]:
        mag = RNG.random(chunk * nch).astype(np.float32)
        ref_rs = orc.Resampler()
        want = []
        for c in range(nch):
            o = ref_rs.process(mag[c * chunk:(c + 1) * chunk], r, 1.0)
            o[min(ref_rs.last_emitted, o.size):] = 0.0
            want.append(o)
        want = np.concatenate(want)
        rs = gpu.Resampler(g)
        d_out = g.empty(want.size + 8)
        n = rs.process(g.to_device(mag), 0, chunk, nch, r, 1.0, 0, d_out)
        assert n == want.size and np.array_equal(d_out.download(n), want), (chunk, nch, r)
        assert rs.state() == (ref_rs.st.contrib, ref_rs.st.offset)


def test_round3_entry_points_reject_misuse_and_take_empty_input(orc):
    """The certified detector, the band resampler, the band chain and the sharded stitch: empty calls are no-ops, calls out
    of order or out of range fail with TSDRGPU_E* (never a crash, never a silent wrong answer)."""
    g = ctx()
    d = g.empty(1 << 16)
    # certified autocorrelation
    ac = gpu.Autocorr(g, 300_000)
    ac.set_certify(1, retain_bytes=1)  # rounded up to one window
    ac.run(d, 0, ac.capture, 0)        # no window: nothing retained, nothing run
    fi, li, promoted = ac.argmax_certified()  # an all-zero plot: every lag ties -> replayed (nothing to replay) -> index 0
    assert (fi, li, promoted) == (0, 0, 1) and ac.certificate().exact_epoch
    ac.reset()
    assert not ac.certificate().exact_epoch or ac.certificate().promotions == 1
    with pytest.raises(gpu.TsdrGpuError):
        ac.set_certify(3)
    ac.set_certify(0)
    with pytest.raises(gpu.TsdrGpuError):
        ac.promote()  # certified mode is off
    # band resampler
    rs = gpu.Resampler(g)
    assert rs.process_band(d, 0, 100, 0, 2.0, 1.0, 16, 16, 0, 16, 0, d, 4) == (0, 0)
    for bad in (dict(y0=8, rows=16), dict(y0=0, rows=0), dict(phase=16 * 16), dict(cap=0)):
        kw = dict(y0=0, rows=16, phase=0, cap=4)
        kw.update(bad)
        with pytest.raises(gpu.TsdrGpuError):
            rs.process_band(d, 0, 100, 2, 2.0, 1.0, 16, 16, kw["y0"], kw["rows"], kw["phase"], d, kw["cap"])
    assert rs.state() == (0.0, 0.0)  # a refused call leaves the carried state alone
    rs.track_frames(4096)
    with pytest.raises(gpu.TsdrGpuError):
        rs.process_band(d, 0, 100, 2, 2.0, 1.0, 16, 16, 0, 16, 0, d, 4)  # frame tracking is a full-frame feature
    # band chain
    pp = gpu.PostProcess(g)
    with pytest.raises(gpu.TsdrGpuError):
        pp._nframes = 1
        pp.band_advance(d, 0, 1)  # no band run is open
    with pytest.raises(gpu.TsdrGpuError):
        pp.band_begin(d, 1, 64, 64, 32, 64)  # rows beyond the frame
    pp.band_begin(d, 1, 64, 64, 32, 32)
    with pytest.raises(gpu.TsdrGpuError):
        pp.band_advance(d, 2, 2)  # band index out of range
    # sharded stitch
    with pytest.raises(gpu.TsdrGpuError):
        gpu.SuperbShard(g, 4, 4, 20_000, 2_000)  # my_hop out of range
    sh = gpu.SuperbShard(g, 2, 1, 20_000, 2_000)
    with pytest.raises(gpu.TsdrGpuError):
        sh.spectrum(d)  # before the reference
    with pytest.raises(gpu.TsdrGpuError):
        sh.finish(d)
    sh.destroy()


def test_fused_run_must_be_finished_into_its_own_buffer():
    """tsdrgpu_postproc_begin_minmax queues the frames into the d_out it is given; _finish with another buffer is refused
    (the gated literal redo would otherwise land elsewhere than the frames), the run stays open and can still be closed."""
    g = ctx()
    w, h, F = 300, 64, 8
    n = w * h
    rng = np.random.default_rng(2)
    frames = rng.random(F * n, dtype=np.float32)
    d_in, d_out, d_other = g.to_device(frames), g.empty(F * n), g.empty(F * n)
    fr = frames.reshape(F, n)
    d_mn, d_mx = g.to_device(fr.min(axis=1)), g.to_device(fr.max(axis=1))
    pp = gpu.PostProcess(g)
    pp.begin_minmax(d_in, F, w, h, d_mn.at(0), d_mx.at(0), d_out)
    with pytest.raises(gpu.TsdrGpuError):
        pp.finish(d_other)
    with pytest.raises(gpu.TsdrGpuError):
        pp.run(d_in, F, w, h, d_other)  # still open
    info = pp.finish(d_out)
    ref = gpu.PostProcess(g)
    want = g.empty(F * n)
    info_r = ref.run(d_in, F, w, h, want)
    assert np.array_equal(d_out.download().view(np.uint32), want.download().view(np.uint32))
    assert [(i.dx, i.dy, i.lastmin, i.lastmax) for i in info] == [(i.dx, i.dy, i.lastmin, i.lastmax) for i in info_r]
