"""SURVEY §8(f) rows on the GPU vs the oracle: bit-exact (integer / byte work)."""
import numpy as np
import pytest

from gpu_util import ctx

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fmt,dtype,tid", [("int8", np.int8, 1), ("int16", np.int16, 2), ("uint8", np.uint8, 3),
                                            ("uint16", np.uint16, 4), ("float", np.float32, 0)])
def test_decode_samples_exhaustive(orc, fmt, dtype, tid):
    g = ctx()
    if dtype == np.float32:
        raw = np.random.default_rng(1).standard_normal(100_001).astype(np.float32)
    else:
        info = np.iinfo(dtype)
        raw = np.arange(info.min, info.max + 1).astype(dtype)  # every representable value
        raw = np.concatenate([raw, raw[::-1], raw[:7]])
    n = raw.size
    d_raw = g.empty((raw.nbytes + 3) // 4, np.uint32)
    g._ck(g.lib.tsdrgpu_upload(g.h, d_raw.ptr, raw.ctypes.data, raw.nbytes))
    g.sync()
    d_out = g.empty(n)
    g.decode_samples(d_raw, fmt, d_out, n)
    want = np.empty(n, np.float32)
    orc.lib.orc_decode_samples(raw.ctypes.data, tid, want, n)
    assert np.array_equal(d_out.download(), want)


@pytest.mark.parametrize("inverted", [0, 1])
def test_frame_to_rgb(orc, inverted):
    g = ctx()
    rng = np.random.default_rng(2)
    n = 507 * 525
    fr = (rng.random(n) * 1.3 - 0.15).astype(np.float32)  # below 0, inside (0,1], above 1
    fr[::97] = 256.0
    fr[1::97] = 512.0
    fr[2::97] = 1024.0
    fr[3::97] = 2048.0  # transparent: keeps the previous pixel
    fr[4::97] = 300.0   # an unknown special value
    fr[5] = np.nan
    fr[6] = 1.0
    fr[7] = 0.0
    # every gray level boundary
    fr[1000:1256] = (np.arange(256) / 255.0).astype(np.float32)
    prev = rng.integers(0, 1 << 24, n).astype(np.int32)
    want = prev.copy()
    orc.lib.orc_frame_to_rgb(fr, want, n, inverted)
    d_fr = g.to_device(fr)
    d_rgb = g.to_device(prev)
    g.frame_to_rgb(d_fr, d_rgb, n, inverted)
    got = d_rgb.download()
    assert np.array_equal(got, want)
    keep = fr == 2048.0
    assert keep.sum() > 1000 and np.array_equal(got[keep], prev[keep])


def test_decode_then_pipeline_matches_float_path(orc):
    """int16 IQ decoded on the device feeds the same resampler as the plugin's float block."""
    from tempestsdr_amd import gpu, synth
    g = ctx()
    fs, h, fv = 8_000_000, 525, 60.0
    geo = orc.geometry(fs, h, fv)
    chunk = orc.chunk_size(fs, fv)
    iq = synth.synth_iq(fs, "640x480", fv, 4 * chunk, seed=9)
    raw = np.clip(np.round(iq * 32767.0), -32768, 32767).astype(np.int16)
    host_float = np.empty(raw.size, np.float32)
    orc.lib.orc_decode_samples(raw.ctypes.data, 2, host_float, raw.size)
    want, _ = orc.demod_resample_stream(host_float, geo)
    d_raw = g.empty((raw.nbytes + 3) // 4, np.uint32)
    g._ck(g.lib.tsdrgpu_upload(g.h, d_raw.ptr, raw.ctypes.data, raw.nbytes))
    g.sync()
    d_iq = g.empty(raw.size)
    g.decode_samples(d_raw, "int16", d_iq, raw.size)
    rs = gpu.Resampler(g)
    up, down = geo.width * geo.height * geo.refreshrate, float(fs)
    d_out = g.empty(want.size + 8)
    n = rs.process(d_iq, 1, chunk, 4, up, down, 0, d_out)
    assert n == want.size and np.array_equal(d_out.download(n), want)


@pytest.mark.parametrize("size,nwidth,zoom,offpx", [(668756, 1237, 1.0, 0), (2315, 640, 1.0, 0), (300, 800, 1.0, 0),
                                                     (50000, 800, 0.13, 411), (50000, 800, 0.01, 37000), (1, 16, 1.0, 0),
                                                     (977, 977, 1.0, 0), (50000, 800, 1.0, -50), (50000, 800, 1.0, 900)])
def test_plot_columns_bit_exact(orc, size, nwidth, zoom, offpx):
    """f4: PlotVisualizer.populateData on the device plot == the oracle's loop, all outputs identical."""
    from tempestsdr_amd import gpu as G
    g = ctx()
    rng = np.random.default_rng(size + nwidth)
    data = rng.random(size) + 0.2 * np.sin(np.arange(size) / 37.0)
    data[rng.integers(0, size, 3)] = 1.5  # exact ties: the first one is the argmax
    d = g.empty(2 * size, np.float32)  # bytes for `size` doubles
    g._ck(g.lib.tsdrgpu_upload(g.h, d.ptr, data.ctypes.data, data.nbytes))
    g.sync()
    if zoom == 1.0 and offpx == 0:
        so, sg = None, None
    else:
        so, sg = orc.PlotScale(), G.PlotScale()
        for s in (so, sg):
            span = float(size) * zoom
            s.one_val_in_pixels = nwidth / span
            s.one_px_in_values = span / nwidth
            s.offset_px = offpx
            s.offset_val = offpx * s.one_px_in_values
            s.min_value = 0.0
    want = orc.plot_populate(data, nwidth, so)
    got = g.plot_columns(d.ptr, size, nwidth, sg)
    assert np.array_equal(got[0], want[0])
    assert got[1:] == want[1:]


def test_plot_columns_on_autocorr_plot(orc):
    """The frame-lag plot of an autocorrelation run, decimated on the device, vs the oracle on the downloaded plot."""
    from tempestsdr_amd import gpu as G
    g = ctx()
    fs = 2_000_000
    ac = G.Autocorr(g, fs)
    x = (np.random.default_rng(5).random(ac.capture) + (np.arange(ac.capture) % (fs // 60) < 900)).astype(np.float32)
    ac.run(g.to_device(x), False, ac.capture, 1)
    f, l, _ = ac.plots()
    p, n = ac.device_plots()
    assert n == f.size + l.size
    got = g.plot_columns(p, f.size, 800)
    want = orc.plot_populate(f, 800)
    assert np.array_equal(got[0], want[0]) and got[1:] == want[1:]
    assert got[3] == int(np.argmax(f)) or f[0] == f.max()
    got = g.plot_columns(p + 8 * f.size, l.size, 800)
    want = orc.plot_populate(l, 800)
    assert np.array_equal(got[0], want[0]) and got[1:] == want[1:]
