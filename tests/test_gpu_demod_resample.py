"""a1 + a2 on the GPU vs the oracle: bit-exact (integer-like contract: the
resampler decides which sample lands in which pixel)."""
import numpy as np
import pytest

from tempestsdr_amd import gpu, synth
from gpu_util import ctx, golden

pytestmark = pytest.mark.gpu
RNG = np.random.default_rng(7)


@pytest.mark.parametrize("n,off_in,off_out", [(1, 0, 0), (3, 0, 0), (4096, 0, 0), (100_003, 0, 0),
                                               (100_003, 2, 0), (100_003, 0, 1), (1 << 20, 4, 4)])
def test_am_demod_bit_exact(orc, n, off_in, off_out):
    g = ctx()
    iq = (RNG.standard_normal(2 * n) * 3).astype(np.float32)
    d_iq = g.empty(2 * n + 8)
    d_iq.upload(iq, off_in)
    d_out = g.empty(n + 8)
    g.am_demod(d_iq, d_out, n, iq_offset=off_in, out_offset=off_out)
    got = d_out.download(n, off_out)
    assert np.array_equal(got, orc.am_demod(iq))


def test_am_demod_golden():
    g = ctx()
    gold = golden()
    d_iq = g.to_device(gold["demod_in"])
    d_out = g.empty(5000)
    g.am_demod(d_iq, d_out, 5000)
    assert np.array_equal(d_out.download(), gold["demod_out"])


GEOMS = [(8_000_000, 525, 60.0), (25_000_000, 806, 60.0), (100_000_000, 1125, 60.0),
         (12_600_000, 525, 60.0), (10_000_000, 625, 50.0), (200_000_000, 2250, 60.004)]


@pytest.mark.parametrize("fs,h,fv", GEOMS)
@pytest.mark.parametrize("nearest", [0, 1])
@pytest.mark.parametrize("from_iq", [0, 1])
def test_resample_library_geometry(orc, fs, h, fv, nearest, from_iq):
    """6 consecutive 0.1-frame chunks in ONE launch + 3 more in a second call
    (state carried on the device) == 9 sequential dsp_resample_process calls."""
    g = ctx()
    geo = orc.geometry(fs, h, fv)
    up, down = geo.width * geo.height * geo.refreshrate, float(fs)
    chunk = orc.chunk_size(fs, fv)
    nch = 9
    if from_iq:
        iq = synth.synth_iq(fs, "640x480", fv, nch * chunk, seed=3)
        mag = orc.am_demod(iq)
        d_in = g.to_device(iq)
    else:
        mag = RNG.random(nch * chunk).astype(np.float32)
        d_in = g.to_device(mag)
    ref_rs = orc.Resampler()
    want = []
    for c in range(nch):
        o = ref_rs.process(mag[c * chunk:(c + 1) * chunk], up, down, nearest)
        k = min(ref_rs.last_emitted, o.size)
        o[k:] = 0.0  # never stored by the reference's loop (aligned case): we define 0
        want.append(o)
    want = np.concatenate(want)

    rs = gpu.Resampler(g)
    d_out = g.empty(want.size + 16)
    assert rs.count(chunk, nch, up, down) == want.size
    n1 = rs.process(d_in, from_iq, chunk, 6, up, down, nearest, d_out)
    n2 = rs.process(d_in, from_iq, chunk, 3, up, down, nearest, d_out,
                    in_offset=6 * chunk * (2 if from_iq else 1), out_offset=n1)
    assert n1 + n2 == want.size
    got = d_out.download(want.size)
    bad = np.flatnonzero(got != want)
    assert bad.size == 0, (bad[:8], got[bad[:8]], want[bad[:8]])
    con, off = rs.state()
    assert off == ref_rs.st.offset
    if not nearest:
        assert con == ref_rs.st.contrib


@pytest.mark.parametrize("chunk,nch", [(1013, 7), (166_666, 2), (1, 5), (62, 3), (249, 4)])
@pytest.mark.parametrize("r", [0.05, 0.37, 0.5, 0.999, 1.0, 1.0000001, 1.5, 1.99935, 2.0, 3.25, 4.0 / 3.0, 7.0, 8.0, 8.0000001, 13.5])
def test_resample_rates(orc, r, chunk, nch):
    """every ratio class: below 1 and above 8 run the pixel-group kernel (k_rs_area), 1 <= r <= 8 the
    sample-parallel one (k_rs_area_up; its workgroups take 1 to 8 rounds of 62 samples per wave depending on r),
    chunk sizes around the kernels' 62-sample / 248-sample granularity"""
    g = ctx()
    if chunk * nch * max(r, 1.0) > 4e6:
        nch = 1
    mag = RNG.random(nch * chunk).astype(np.float32)
    ref_rs = orc.Resampler()
    want = []
    for c in range(nch):
        o = ref_rs.process(mag[c * chunk:(c + 1) * chunk], r, 1.0)
        o[min(ref_rs.last_emitted, o.size):] = 0.0
        want.append(o)
    want = np.concatenate(want)
    rs = gpu.Resampler(g)
    d_in = g.to_device(mag)
    d_out = g.empty(want.size + 16)
    n = rs.process(d_in, 0, chunk, nch, r, 1.0, 0, d_out)
    assert n == want.size
    assert np.array_equal(d_out.download(n), want)
    con, off = rs.state()
    assert (con, off) == (ref_rs.st.contrib, ref_rs.st.offset)


def test_resample_golden(orc):
    from cases import RESAMPLE
    g = ctx()
    gold = golden()
    geo = orc.geometry(RESAMPLE["fs"], RESAMPLE["h"], RESAMPLE["fv"])
    up, down = geo.width * geo.height * geo.refreshrate, float(RESAMPLE["fs"])
    chunk = orc.chunk_size(RESAMPLE["fs"], RESAMPLE["fv"])
    d_in = g.to_device(gold["resample_in"])
    for nearest in (0, 1):
        want = gold[f"resample_out_{nearest}"]
        rs = gpu.Resampler(g)
        d_out = g.empty(want.size + 8)
        n = rs.process(d_in, 0, chunk, RESAMPLE["chunks"], up, down, nearest, d_out)
        assert n == want.size
        assert np.array_equal(d_out.download(n), want)
        con, off = rs.state()
        assert off == gold[f"resample_state_{nearest}"][-1][1]
        if not nearest:
            assert con == gold[f"resample_state_{nearest}"][-1][0]


def test_resample_full_size_properties(orc):
    """BASELINE config 3 (100 MS/s, 1125 lines): one second of signal is too
    long for the oracle's loop in a test, so check size-independent properties:
    a constant input resamples to the same constant, counts match the host
    recurrence, and a random 64-chunk slice matches the oracle exactly."""
    g = ctx()
    fs, h, fv = 100_000_000, 1125, 60.0
    geo = orc.geometry(fs, h, fv)
    up, down = geo.width * geo.height * geo.refreshrate, float(fs)
    chunk = orc.chunk_size(fs, fv)
    nch = 120  # 0.2 s
    d_in = g.to_device(np.full(nch * chunk, 0.625, np.float32))
    rs = gpu.Resampler(g)
    total = rs.count(chunk, nch, up, down)
    d_out = g.empty(total)
    assert rs.process(d_in, 0, chunk, nch, up, down, 0, d_out) == total
    out = d_out.download()
    assert np.all(out == np.float32(0.625))
    # linearity in the data: resample(a*x) == a*resample(x) for a power of two
    x = RNG.random(nch * chunk).astype(np.float32)
    d_in.upload(x)
    rs.reset()
    rs.process(d_in, 0, chunk, nch, up, down, 0, d_out)
    y1 = d_out.download()
    d_in.upload(x * np.float32(4.0))
    rs.reset()
    rs.process(d_in, 0, chunk, nch, up, down, 0, d_out)
    assert np.array_equal(d_out.download(), y1 * np.float32(4.0))
    # oracle on the first 8 chunks
    ref_rs = orc.Resampler()
    want = np.concatenate([ref_rs.process(x[c * chunk:(c + 1) * chunk], up, down) for c in range(8)])
    assert np.array_equal(y1[:want.size], want)


@pytest.mark.parametrize("up", [1.99935, 0.62, 9.3, 1.0, 8.0])  # sample-parallel kernel (1 <= r <= 8) and the pixel-group one
@pytest.mark.parametrize("from_iq", [0, 1])
@pytest.mark.parametrize("P,phase0", [(4096, 0), (5003, 4321), (43623, 17), (100_000, 99_999)])
def test_resampler_frame_minmax_tracking(from_iq, P, phase0, up):
    """tsdrgpu_resampler_track_frames: per-frame min/max of the emitted pixel stream, carried across calls,
    == numpy on the downloaded stream (order-independent reductions: exact), sentinels (|v| > 250) excluded."""
    g = ctx()
    rng = np.random.default_rng(P + from_iq)
    rs = gpu.Resampler(g)
    rs.track_frames(P, phase0)
    down = 1.0
    stream, got_mn, got_mx = [], [], []
    for call, (chunk, nchunks) in enumerate([(3333, 7), (1000, 1), (16667, 13), (50, 3), (7777, 40)]):
        n = chunk * nchunks
        x = rng.random(n).astype(np.float32) * 3.0 - 0.5
        x[rng.integers(0, n, 5)] = 700.0  # produce pixels above the sentinel threshold
        x[rng.integers(0, n, 5)] = -900.0
        if from_iq:
            ph = rng.random(n) * 6.28
            mag = np.abs(x)
            host = np.empty(2 * n, np.float32)
            host[0::2] = (mag * np.cos(ph)).astype(np.float32)
            host[1::2] = (mag * np.sin(ph)).astype(np.float32)
        else:
            host = x
        d_in = g.to_device(host)
        cap = rs.count(chunk, nchunks, up, down)
        d_out = g.empty(cap + 8)
        npix = rs.process(d_in, from_iq, chunk, nchunks, up, down, 0, d_out)
        assert npix == cap
        stream.append(d_out.download()[:npix])
        mn, mx = rs.frame_minmax()
        got_mn += list(mn)
        got_mx += list(mx)
    s = np.concatenate([np.zeros(phase0, np.float32) + np.float32(1e9), np.concatenate(stream)])  # phase0 pixels came "before"
    nfr = s.size // P
    assert len(got_mn) == nfr
    for f in range(nfr):
        fr = s[f * P:(f + 1) * P]
        if f == 0:
            fr = fr[phase0:]  # only the tracked part of the first frame is known to the resampler
        ok = fr[np.abs(fr) <= 250.0]
        assert got_mn[f] == ok.min() and got_mx[f] == ok.max(), f"frame {f}"
    with pytest.raises(gpu.TsdrGpuError):
        rs.track_frames(1000)  # too small
    rs.track_frames(0)
    with pytest.raises(gpu.TsdrGpuError):
        rs.frame_minmax()
