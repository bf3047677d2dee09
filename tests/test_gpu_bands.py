"""the frame path sharded by row bands (tsdrgpu_postproc_band_begin / _finish).

Two "ranks" (two post-processing objects; in production one process per GPU) each hold a band of rows of every frame.
Their exchange — a sum all-reduce of the strip partials and a max all-reduce of {-min, max, pixel 0} — is done here by
the test on the host (adding / maximising the two device buffers), the production collective being
tsdrgpu_comm_allreduce_f64 / _f32max over RCCL, which is exercised with a one-rank communicator.  The concatenated
band outputs, the per-frame sync / autogain records and the state carried over several batches must equal the
single-GPU run bit for bit (its fast mode: a band cannot walk a column through the other bands for the literal
re-collapse of toss-up strips).  Reference arithmetic: dsp.c:41-110, syncdetector.c:171-225."""
import os

import numpy as np
import pytest

from tempestsdr_amd import gpu
from gpu_util import ctx

HERE_ROOT = os.path.abspath(__file__)

pytestmark = pytest.mark.gpu


def _frames(rng, F, W, H, k0):
    y, x = np.mgrid[0:H, 0:W]
    out = np.empty((F, H, W), np.float32)
    for f in range(F):
        img = 0.3 + 0.5 * (((x + 3 * (k0 + f)) // 37) % 2) + 0.1 * (((y // 16) + (x // 16)) % 2)
        img[:, : W // 9] = 0.05  # horizontal blanking
        img[: H // 20, :] = 0.05  # vertical blanking
        out[f] = (img + rng.standard_normal((H, W)) * 0.02).astype(np.float32)
    return out


def _exchange(g, bufs):
    """all-reduce of the ranks' exchange buffers on the host: sum for the doubles, max for the floats"""
    (ps0, ns, pm0, nm), (ps1, _, pm1, _) = bufs
    s = [np.empty(ns, np.float64) for _ in range(2)]
    m = [np.empty(nm, np.float32) for _ in range(2)]
    for k, (ps, pm) in enumerate(((ps0, pm0), (ps1, pm1))):
        g._ck(g.lib.tsdrgpu_download(g.h, s[k].ctypes.data, ps, s[k].nbytes))
        g._ck(g.lib.tsdrgpu_download(g.h, m[k].ctypes.data, pm, m[k].nbytes))
    g.sync()
    ssum, mmax = s[0] + s[1], np.maximum(m[0], m[1])
    for ps, pm in ((ps0, pm0), (ps1, pm1)):
        g._ck(g.lib.tsdrgpu_upload(g.h, ps, ssum.ctypes.data, ssum.nbytes))
        g._ck(g.lib.tsdrgpu_upload(g.h, pm, mmax.ctypes.data, mmax.nbytes))
    g.sync()


@pytest.mark.parametrize("W,H,split,blur,sentinels", [(507, 525, 256, 0.0, False), (1033, 806, 416, 0.5, False), (2962, 2250, 1152, 0.9375, False),
                                                        (300, 200, 96, 0.0, True)])
def test_two_row_bands_equal_the_single_gpu_run(W, H, split, blur, sentinels):
    g = ctx()
    rng = np.random.default_rng(W + H)
    ref = gpu.PostProcess(g)
    ref.set_exact_ties(False)
    bands = [gpu.PostProcess(g), gpu.PostProcess(g)]
    rows = [(0, split), (split, H - split)]
    for batch, F in enumerate((3, 5, 1)):
        fr = _frames(rng, F, W, H, 10 * batch)
        if sentinels:
            fr[0, 0, 0] = 512.0      # pixel 0 a sentinel: the v[0] quirk of dsp.c:50-51 crosses the exchange
            fr[F - 1, 150, 7] = 1024.0  # a sentinel in band 1
        d_full, d_out = g.to_device(fr.reshape(-1)), g.empty(F * W * H)
        infos = ref.run(d_full, F, W, H, d_out, motionblur=blur)
        want = d_out.download().reshape(F, H, W)
        outs, bufs, d_bands = [], [], []
        for pp, (y0, n) in zip(bands, rows):
            d_b = g.to_device(np.ascontiguousarray(fr[:, y0:y0 + n, :]).reshape(-1))
            d_bands.append(d_b)
            bufs.append(pp.band_begin(d_b, F, W, H, y0, n, motionblur=blur))
        _exchange(g, bufs)
        for pp, (y0, n) in zip(bands, rows):
            d_ob = g.empty(F * W * n)
            binfo = pp.band_finish(d_ob)
            outs.append(d_ob.download().reshape(F, n, W))
            for a, b in zip(infos, binfo):  # every rank ends up with the same record
                ra = (a.lastmin, a.lastmax, a.dx, a.vx, a.stripx, a.dy, a.vy, a.stripy, a.locked, a.avg_speed)
                rb = (b.lastmin, b.lastmax, b.dx, b.vx, b.stripx, b.dy, b.vy, b.stripy, b.locked, b.avg_speed)
                assert ra == rb
        got = np.concatenate(outs, axis=1)
        assert np.array_equal(got, want), (batch, int(np.sum(got != want)))
        if blur == 0.0:
            assert (want == 512.0).any()  # the green sync lines are painted in the right rows / columns of each band


def test_band_exchange_over_rccl_one_rank():
    """A single band covering the whole frame with the exchange done by the production collectives
    (tsdrgpu_comm_allreduce_f64 / _f32max on a one-rank RCCL communicator) equals tsdrgpu_postproc_run."""
    g = ctx()
    W, H, F = 507, 525, 4
    rng = np.random.default_rng(9)
    fr = _frames(rng, F, W, H, 0)
    d_in, d_a, d_b = g.to_device(fr.reshape(-1)), g.empty(F * W * H), g.empty(F * W * H)
    ref = gpu.PostProcess(g)
    ref.set_exact_ties(False)
    ref.run(d_in, F, W, H, d_a)
    pp = gpu.PostProcess(g)
    comm = gpu.Comm(g, 1, 0, gpu.Comm.unique_id(g))
    ps, ns, pm, nm = pp.band_begin(d_in, F, W, H, 0, H)
    comm.allreduce_f64(ps, ns)
    comm.allreduce_f32max(pm, nm)
    pp.band_finish(d_b)
    assert np.array_equal(d_a.download(), d_b.download())
    comm.destroy()
    with pytest.raises(gpu.TsdrGpuError):
        pp.band_begin(d_in, F, W, H, 16, 100)  # bands start on multiples of 32 rows


# ---------------------------------------------------------------------------
# contract-exact bands (tsdrgpu_postproc_band_advance) and the band resampler, against the ORACLE
# ---------------------------------------------------------------------------
def _sum_exchange(g, ptrs, n):
    """sum all-reduce of the ranks' double buffers on the host (production: tsdrgpu_comm_allreduce_f64 over RCCL)"""
    bufs = [np.empty(n, np.float64) for _ in ptrs]
    for b, p in zip(bufs, ptrs):
        g._ck(g.lib.tsdrgpu_download(g.h, b.ctypes.data, p, b.nbytes))
    g.sync()
    tot = np.sum(bufs, axis=0)
    for p in ptrs:
        g._ck(g.lib.tsdrgpu_upload(g.h, p, tot.ctypes.data, tot.nbytes))
    g.sync()


def _run_bands(g, pps, rows, fr, blur):
    """one batch through the band protocol; returns (concatenated output, per-frame records, relay steps taken)"""
    F, H, W = fr.shape
    begun, d_bands, d_outs = [], [], []
    for pp, (y0, n) in zip(pps, rows):
        d_b = g.to_device(np.ascontiguousarray(fr[:, y0:y0 + n, :]).reshape(-1))
        d_bands.append(d_b)
        d_outs.append(g.empty(F * W * n))
        begun.append(pp.band_begin(d_b, F, W, H, y0, n, motionblur=blur))
    # the two all-reduces of _band_begin (sum of the strip partials, max of {-min, max, pixel 0})
    ns, nm = begun[0][1], begun[0][3]
    _sum_exchange(g, [b[0] for b in begun], ns)
    m = [np.empty(nm, np.float32) for _ in begun]
    for k, b in enumerate(begun):
        g._ck(g.lib.tsdrgpu_download(g.h, m[k].ctypes.data, b[2], m[k].nbytes))
    g.sync()
    mm = np.maximum.reduce(m)
    for b in begun:
        g._ck(g.lib.tsdrgpu_upload(g.h, b[2], mm.ctypes.data, mm.nbytes))
    g.sync()
    steps, infos = 0, None
    while True:
        res = [pp.band_advance(d_o, k, len(pps)) for k, (pp, d_o) in enumerate(zip(pps, d_outs))]
        mores = {r[0] for r in res}
        assert len(mores) == 1, "every rank takes the same decision"
        if not res[0][0]:
            infos = [r[3] for r in res]
            break
        assert len({r[2] for r in res}) == 1
        _sum_exchange(g, [r[1] for r in res], res[0][2])
        steps += 1
    outs = [d_o.download().reshape(F, n, W) for d_o, (_, n) in zip(d_outs, rows)]
    for other in infos[1:]:  # every rank ends up with the same record
        for a, b in zip(infos[0], other):
            assert (a.lastmin, a.lastmax, a.dx, a.vx, a.stripx, a.dy, a.vy, a.stripy, a.locked, a.avg_speed) == \
                   (b.lastmin, b.lastmax, b.dx, b.vx, b.stripx, b.dy, b.vy, b.stripy, b.locked, b.avg_speed)
    return np.concatenate(outs, axis=1), infos[0], steps


def _geo(orc, W, H):
    geo = orc.Geometry()
    geo.samplerate, geo.width, geo.height, geo.refreshrate = 8_000_000, W, H, 60.0
    geo.pixelrate = W * H * 60.0
    geo.pixeltimeoversampletime = geo.samplerate / geo.pixelrate
    return geo


def _bench_cuts(H, world):
    """bench.py --bands' edge rule: bands start on multiples of 32 rows, the last one takes the remainder"""
    return tuple(32 * ((H * k // world) // 32) for k in range(1, world))


@pytest.mark.parametrize("W,H,cuts,blur", [(507, 525, (160, 352), 0.0), (1033, 806, (416,), 0.5), (640, 420, (96, 224, 320), 0.9375),
                                           # BASELINE configs[4] in its named shape: 2962x2250 frames, motion blur 15/16, EIGHT bands
                                           (2962, 2250, _bench_cuts(2250, 8), 0.9375)])
def test_exact_row_bands_equal_the_oracle(orc, W, H, cuts, blur):
    """2, 3, 4 and 8 bands in the default, contract-exact mode against the ORACLE (dsp_post_process restated, pinned to the
    compiled reference): frames bit for bit and the sync / autogain records, over batches that hold every kind of
    strip — noisy rasters (toss-ups at most), a blank frame and a noiseless pattern (exact ties: the literal collapse
    is relayed band by band), a sentinel.  State is carried from batch to batch."""
    g = ctx()
    assert all(c % 32 == 0 for c in cuts) and list(cuts) == sorted(set(cuts))
    rng = np.random.default_rng(3 * W + H)
    edges = (0,) + tuple(cuts) + (H,)
    rows = [(a, b - a) for a, b in zip(edges[:-1], edges[1:])]
    pps = [gpu.PostProcess(g) for _ in rows]
    single = gpu.PostProcess(g)  # the single-GPU run in its default (exact) mode
    opp = orc.PostProcess(_geo(orc, W, H))
    total_steps = 0
    for batch, F in enumerate((4, 9, 2)):
        fr = _frames(rng, F, W, H, 10 * batch)
        if batch == 0:
            fr[1] = 0.25                                                # blank: every window position ties
            y, x = np.mgrid[0:H, 0:W]
            fr[2] = (0.3 + 0.5 * ((x // 40) % 2)).astype(np.float32)    # plateaus without noise: ties in both strips
        if batch == 1:
            fr[3, 5, 7] = 1024.0                                        # a sentinel pixel
        want = np.stack([opp.run(fr[k].reshape(-1).copy(), blur, 0.1, 0, 0, 0, 0, 0).reshape(H, W) for k in range(F)])
        d_full, d_out = g.to_device(fr.reshape(-1)), g.empty(F * W * H)
        sinfo = single.run(d_full, F, W, H, d_out, motionblur=blur)
        assert np.array_equal(d_out.download().reshape(F, H, W), want)
        got, infos, steps = _run_bands(g, pps, rows, fr, blur)
        total_steps += steps
        assert steps % len(rows) == 0  # whole relays only
        if batch == 0:
            assert steps >= len(rows)  # the blank frame needed one
        assert np.array_equal(got, want), (batch, int(np.sum(got != want)))
        for a, b in zip(sinfo, infos):
            assert (a.lastmin, a.lastmax, a.dx, a.vx, a.stripx, a.dy, a.vy, a.stripy, a.locked, a.avg_speed) == \
                   (b.lastmin, b.lastmax, b.dx, b.vx, b.stripx, b.dy, b.vy, b.stripy, b.locked, b.avg_speed)
    si, sd = opp.state()
    last = infos[-1]
    assert (last.dx, last.vx, last.stripx, last.dy, last.vy, last.stripy, last.locked) == tuple(si[:7])
    # the speculated form (the default): every batch was queued to its end first; the batch with the blank frame raised a flag, was put
    # back and taken literally — and the batches behind it started from the right state, or the frames above would differ
    for pp in pps:
        runs, replays = pp.band_spec_stats()
        if os.environ.get("TSDRGPU_BAND_SPECULATE", "1") != "0":
            assert runs == 3 and 1 <= replays <= 3, (runs, replays)
        else:
            assert runs == 0 and replays == 0
    assert len({pp.band_spec_stats() for pp in pps}) == 1  # every rank speculates, and gives up, alike


def test_exact_row_bands_without_speculation():
    """the literal band run alone (TSDRGPU_BAND_SPECULATE=0, read once per process: hence a process of its own): the same tests"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TSDRGPU_BAND_SPECULATE="0")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_bands.py"), "-q", "-m", "gpu", "-p", "no:cacheprovider",
                          "-k", "test_exact_row_bands_equal_the_oracle and not 2962"], capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert out.returncode == 0 and " passed" in out.stdout, out.stdout[-3000:] + out.stderr[-1000:]


@pytest.mark.parametrize("fs,h,y0,rows", [(8_000_000, 525, 160, 192), (8_000_000, 525, 0, 525), (25_000_000, 806, 416, 390),
                                          (2_000_000, 131, 32, 64), (7_000_000, 525, 352, 173)])
def test_band_resampler_equals_the_oracle_stream(orc, fs, h, y0, rows):
    """tsdrgpu_resample_band over several calls (frames straddle the calls; the incomplete frame is carried into slot 0):
    every band row equals the same row of the ORACLE's pixel stream, bit for bit, and the carried state (offset, contrib)
    is the full call's."""
    from tempestsdr_amd import synth
    g = ctx()
    fv = 60.0
    geo = orc.geometry(fs, h, fv)
    W, P = geo.width, geo.width * h
    chunk = orc.chunk_size(fs, fv)
    nch = 47
    mode = {525: "640x480", 806: "1024x768", 131: "640x480"}[h]
    iq = synth.synth_iq(fs, mode, fv, nch * chunk, seed=0x5EED0002)
    want, _ = orc.demod_resample_stream(iq, geo)
    d_iq = g.to_device(iq)
    up, down = W * h * fv, float(fs)
    rs_band, rs_full = gpu.Resampler(g), gpu.Resampler(g)
    cap = 8
    d_band = g.empty(cap * rows * W)
    frames = {}
    phase, done_chunks, frame0 = 0, 0, 0
    for k in (5, 1, 13, 10, 18):  # chunks per call
        n, touched = rs_band.process_band(d_iq, 1, chunk, k, up, down, W, h, y0, rows, phase, d_band, cap, in_offset=2 * done_chunks * chunk)
        assert n == rs_full.count(chunk, k, up, down) and touched == (phase + n + P - 1) // P
        d_scratch = g.empty(n + 16)
        assert rs_full.process(d_iq, 1, chunk, k, up, down, 0, d_scratch, in_offset=2 * done_chunks * chunk) == n
        assert rs_band.state() == rs_full.state()
        got = d_band.download().reshape(cap, rows, W)
        complete = (phase + n) // P
        for j in range(complete):
            frames[frame0 + j] = got[j].copy()
        phase = (phase + n) % P
        if phase:  # the incomplete frame goes on in slot 0 of the next call
            part = got[complete].reshape(-1).copy()
            g._ck(g.lib.tsdrgpu_upload(g.h, d_band.at(0), part.ctypes.data, part.nbytes))
            g.sync()
        frame0 += complete
        done_chunks += k
    assert len(frames) >= 3
    for j, fr in frames.items():
        ref = want[j * P:(j + 1) * P].reshape(h, W)[y0:y0 + rows]
        assert np.array_equal(fr, ref), (j, int(np.sum(fr != ref)))


@pytest.mark.parametrize("fs,h,y0,rows", [(8_000_000, 525, 160, 192), (8_000_000, 525, 0, 525), (25_000_000, 806, 416, 390), (7_000_000, 525, 352, 173),
                                          (8_000_000, 525, 0, 32), (8_000_000, 525, 512, 13)])
def test_band_resampler_tracks_the_bands_share_of_every_frames_range(orc, fs, h, y0, rows):
    """frame tracking in the band form (what the FUSED band run exchanges): over several calls whose frames straddle the call
    boundaries, every completed frame's min / max over THIS band's pixels equals the min / max of those rows of the ORACLE's pixel
    stream — and the band's pixels are still the oracle's, bit for bit."""
    from tempestsdr_amd import synth
    g = ctx()
    fv = 60.0
    geo = orc.geometry(fs, h, fv)
    W, P = geo.width, geo.width * h
    chunk = orc.chunk_size(fs, fv)
    nch = 47
    mode = {525: "640x480", 806: "1024x768"}[h]
    iq = synth.synth_iq(fs, mode, fv, nch * chunk, seed=0x5EED0003)
    want, _ = orc.demod_resample_stream(iq, geo)
    d_iq = g.to_device(iq)
    up, down = W * h * fv, float(fs)
    rs = gpu.Resampler(g)
    rs.track_frames(P, 0)
    cap = 8
    d_band = g.empty(cap * rows * W)
    phase, done_chunks, frame0, seen = 0, 0, 0, 0
    for k in (5, 1, 13, 10, 18):
        n, touched = rs.process_band(d_iq, 1, chunk, k, up, down, W, h, y0, rows, phase, d_band, cap, in_offset=2 * done_chunks * chunk)
        got = d_band.download().reshape(cap, rows, W)
        complete = (phase + n) // P
        mn, mx = rs.frame_minmax()
        assert len(mn) == complete == len(mx)
        for j in range(complete):
            ref = want[(frame0 + j) * P:(frame0 + j + 1) * P].reshape(h, W)[y0:y0 + rows]
            assert np.array_equal(got[j], ref)
            assert mn[j] == ref.min() and mx[j] == ref.max(), (frame0 + j, mn[j], ref.min(), mx[j], ref.max())
            seen += 1
        phase = (phase + n) % P
        if phase:
            part = got[complete].reshape(-1).copy()
            g._ck(g.lib.tsdrgpu_upload(g.h, d_band.at(0), part.ctypes.data, part.nbytes))
            g.sync()
        frame0 += complete
        done_chunks += k
    assert seen >= 3
    with pytest.raises(RuntimeError):  # another phase than the tracker's
        rs.process_band(d_iq, 1, chunk, 1, up, down, W, h, y0, rows, (phase + 1) % P, d_band, cap)


def _max_exchange(g, ptrs, n):
    m = [np.empty(n, np.float32) for _ in ptrs]
    for b, p in zip(m, ptrs):
        g._ck(g.lib.tsdrgpu_download(g.h, b.ctypes.data, p, b.nbytes))
    g.sync()
    mm = np.maximum.reduce(m)
    for p in ptrs:
        g._ck(g.lib.tsdrgpu_upload(g.h, p, mm.ctypes.data, mm.nbytes))
    g.sync()


def _run_bands_fused(g, pps, rows, fr, blur):
    """one batch through the FUSED band protocol (range exchanged first, one trip, then the chain): like _run_bands"""
    F, H, W = fr.shape
    d_bands, d_outs, keep = [], [], []
    xm = []
    for pp, (y0, n) in zip(pps, rows):
        band = np.ascontiguousarray(fr[:, y0:y0 + n, :])
        d_b = g.to_device(band.reshape(-1))
        d_bands.append(d_b)
        d_outs.append(g.empty(F * W * n))
        # the band's share of the range as the tracked band resampler leaves it: sentinels skipped (dsp.c:57), +-inf when nothing is left
        ok = ~((band > 250.0) | (band < -250.0)) & ~np.isnan(band)
        mn = np.array([band[f][ok[f]].min() if ok[f].any() else np.inf for f in range(F)], np.float32)
        mx = np.array([band[f][ok[f]].max() if ok[f].any() else -np.inf for f in range(F)], np.float32)
        d_mn, d_mx = g.to_device(mn), g.to_device(mx)
        keep += [d_mn, d_mx]
        xm.append(pp.band_begin_minmax(d_b, F, W, H, y0, n, d_mn.at(0), d_mx.at(0), motionblur=blur))
    _max_exchange(g, [x[0] for x in xm], xm[0][1])
    xs = [pp.band_fused(d_o) for pp, d_o in zip(pps, d_outs)]
    _sum_exchange(g, [x[0] for x in xs], xs[0][1])
    steps, infos = 0, None
    while True:
        res = [pp.band_advance(d_o, k, len(pps)) for k, (pp, d_o) in enumerate(zip(pps, d_outs))]
        assert len({r[0] for r in res}) == 1, "every rank takes the same decision"
        if not res[0][0]:
            infos = [r[3] for r in res]
            break
        _sum_exchange(g, [r[1] for r in res], res[0][2])
        steps += 1
    outs = [d_o.download().reshape(F, n, W) for d_o, (_, n) in zip(d_outs, rows)]
    return np.concatenate(outs, axis=1), infos[0], steps


@pytest.mark.parametrize("W,H,cuts,blur", [(507, 525, (160, 352), 0.0), (1033, 806, (416,), 0.5), (640, 420, (96, 224, 320), 0.9375),
                                           (2962, 2250, _bench_cuts(2250, 8), 0.9375), (300, 200, (), 0.25)])
def test_fused_row_bands_equal_the_oracle(orc, W, H, cuts, blur):
    """The FUSED band run (range exchanged first, ONE trip over the raw band, then the replicated contract-exact chain) against the
    ORACLE: 1, 2, 3, 4 and 8 bands, motion blur 0 (painted lines) and above, batches with a blank frame and a noiseless pattern
    (exact ties: the literal collapse is relayed), a sentinel, a -0.0 pixel; state carried from batch to batch, and a batch of the
    two-trip form in between (the two forms share the object's state)."""
    g = ctx()
    rng = np.random.default_rng(5 * W + H)
    edges = (0,) + tuple(cuts) + (H,)
    rows = [(a, b - a) for a, b in zip(edges[:-1], edges[1:])]
    pps = [gpu.PostProcess(g) for _ in rows]
    opp = orc.PostProcess(_geo(orc, W, H))
    for batch, F in enumerate((4, 9, 3, 2)):
        fr = _frames(rng, F, W, H, 10 * batch)
        if batch == 0:
            fr[1] = 0.25
            y, x = np.mgrid[0:H, 0:W]
            fr[2] = (0.3 + 0.5 * ((x // 40) % 2)).astype(np.float32)
        if batch == 1:
            fr[3, 5, 7] = 1024.0
            fr[4, H - 1, W - 1] = -0.0
        want = np.stack([opp.run(fr[k].reshape(-1).copy(), blur, 0.1, 0, 0, 0, 0, 0).reshape(H, W) for k in range(F)])
        if batch == 2:
            got, infos, steps = _run_bands(g, pps, rows, fr, blur)  # the two-trip form, same objects
        else:
            got, infos, steps = _run_bands_fused(g, pps, rows, fr, blur)
        assert steps % len(rows) == 0
        if batch == 0:
            assert steps >= len(rows)
        assert np.array_equal(got, want, equal_nan=True), (batch, int(np.sum(got != want)))
        si, sd = opp.state()
        last = infos[-1]
        assert (last.dx, last.vx, last.stripx, last.dy, last.vy, last.stripy, last.locked) == tuple(si[:7]), batch


def test_fused_band_run_refuses_what_it_cannot_do():
    g = ctx()
    pp = gpu.PostProcess(g)
    W, H, F = 300, 128, 2
    d = g.to_device(np.zeros(F * W * 64, np.float32))
    d_o = g.empty(F * W * 64)
    mn = g.to_device(np.zeros(F, np.float32))
    with pytest.raises(RuntimeError):
        pp.band_begin_minmax(d, F, W, H, 16, 64, mn.at(0), mn.at(0))  # a band must start on a multiple of 32 rows
    with pytest.raises(RuntimeError):
        pp.band_fused(d_o)  # nothing begun
    pp.band_begin_minmax(d, F, W, H, 32, 64, mn.at(0), mn.at(0))
    with pytest.raises(RuntimeError):
        pp.band_advance(d_o, 0, 1)  # the trip has not been made
    with pytest.raises(RuntimeError):
        pp.band_finish(d_o)  # the fast form does not close a fused run
    with pytest.raises(RuntimeError):
        pp.band_fused(d)  # output over the band
    pp.band_fused(d_o)
    more, _, _, info = pp.band_advance(d_o, 1, 2)
    while more:  # (zeros everywhere: ties; one rank of two relays alone — only the protocol is under test here)
        more, _, _, info = pp.band_advance(d_o, 1, 2)
    assert len(info) == F


# ---------------------------------------------------------------------------
# the GENERAL band run (tsdrgpu_postproc_band_open / _band_step): every stage order, autoshift, PLL — against the ORACLE
# ---------------------------------------------------------------------------
def _host_collective(g, kind, ptrs, count):
    """the collective a step asked for, done on the host over the ranks' device buffers (production: tsdrgpu_comm_* over RCCL)"""
    if kind == gpu.BAND_ALLGATHER_F32:
        world = len(ptrs)
        full = np.empty(world * count, np.float32)
        for r, p in enumerate(ptrs):  # rank r's part sits at r * count of ITS buffer
            part = np.empty(count, np.float32)
            g._ck(g.lib.tsdrgpu_download(g.h, part.ctypes.data, p + 4 * r * count, part.nbytes))
            g.sync()
            full[r * count:(r + 1) * count] = part
        for p in ptrs:
            g._ck(g.lib.tsdrgpu_upload(g.h, p, full.ctypes.data, full.nbytes))
        g.sync()
        return
    dt = np.float64 if kind == gpu.BAND_SUM_F64 else np.float32
    bufs = [np.empty(count, dt) for _ in ptrs]
    for b, p in zip(bufs, ptrs):
        g._ck(g.lib.tsdrgpu_download(g.h, b.ctypes.data, p, b.nbytes))
    g.sync()
    tot = np.sum(bufs, axis=0) if kind == gpu.BAND_SUM_F64 else np.maximum.reduce(bufs)
    for p in ptrs:
        g._ck(g.lib.tsdrgpu_upload(g.h, p, tot.ctypes.data, tot.nbytes))
    g.sync()


def _run_bands_general(g, pps, edges, fr, **prm):
    F, H, W = fr.shape
    rows = [(a, b - a) for a, b in zip(edges[:-1], edges[1:])]
    d_bands = [g.to_device(np.ascontiguousarray(fr[:, y0:y0 + n, :]).reshape(-1)) for (y0, n) in rows]
    d_outs = [g.empty(F * W * n) for (_, n) in rows]
    for k, pp in enumerate(pps):
        pp.band_open(d_bands[k], F, W, H, edges, k, **prm)
    kinds = []
    while True:
        res = [pp.band_step(d_o) for pp, d_o in zip(pps, d_outs)]
        assert len({(r[0], r[2]) for r in res}) == 1, "every rank asks for the same collective"
        kind, _, count, _ = res[0]
        if kind == gpu.BAND_DONE:
            infos = [r[3] for r in res]
            break
        kinds.append(kind)
        _host_collective(g, kind, [r[1] for r in res], count)
    rec = lambda i: np.array([i.lastmin, i.lastmax, i.dx, i.vx, i.stripx, i.dy, i.vy, i.stripy, i.locked, i.avg_speed, i.pll_fired, i.frameratediff], np.float64)
    for other in infos[1:]:
        for a, b in zip(infos[0], other):  # (NaN for NaN: a poisoned autogain is the same poisoned autogain on every rank)
            assert np.array_equal(rec(a), rec(b), equal_nan=True), (rec(a), rec(b))
    outs = [d_o.download().reshape(F, n, W) for d_o, (_, n) in zip(d_outs, rows)]
    return np.concatenate(outs, axis=1), infos[0], kinds


@pytest.mark.parametrize("lbs,aap", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("autoshift", [0, 1])
def test_general_band_run_every_stage_order_equals_the_oracle(orc, lbs, aap, autoshift):
    """dsp_post_process's four stage orders (dsp.c:134-239; lbs = PARAM_LOW_PASS_BEFORE_SYNC — the GUI's default —, aap =
    PARAM_AUTOGAIN_AFTER_PROCESSING) with and without PARAM_INT_AUTOSHIFT (the 2-D roll crosses the bands: all-gather),
    three bands, batches that hold noisy rasters, a blank frame, a noiseless pattern and a sentinel, state carried over
    the batches: frames and per-frame records bit-identical to the ORACLE."""
    g = ctx()
    W, H, blur = 640, 420, 0.5
    edges = [0, 128, 288, H]
    rng = np.random.default_rng(100 * lbs + 10 * aap + autoshift)
    pps = [gpu.PostProcess(g) for _ in range(3)]
    opp = orc.PostProcess(_geo(orc, W, H))
    seen = set()
    for batch, F in enumerate((3, 9, 2)):
        fr = _frames(rng, F, W, H, 10 * batch)
        if batch == 0:
            fr[1] = 0.25
            y, x = np.mgrid[0:H, 0:W]
            fr[2] = (0.3 + 0.5 * ((x // 40) % 2)).astype(np.float32)
        if batch == 1:
            fr[3, 5, 7] = 1024.0
        want = np.stack([opp.run(fr[k].reshape(-1).copy(), blur, 0.1, lbs, aap, autoshift, 0, 0).reshape(H, W) for k in range(F)])
        got, infos, kinds = _run_bands_general(g, pps, edges, fr, motionblur=blur, lowpass_before_sync=lbs, autogain_after_proc=aap, autoshift=autoshift)
        seen |= set(kinds)
        assert np.array_equal(got, want), (batch, int(np.sum(got != want)))
    assert (gpu.BAND_ALLGATHER_F32 in seen) == bool(autoshift)
    assert gpu.BAND_SUM_F64 in seen and gpu.BAND_MAX_F32 in seen
    si, sd = opp.state()
    last = infos[-1]
    assert (last.dx, last.vx, last.stripx, last.dy, last.vy, last.stripy, last.locked) == tuple(si[:7])


def test_general_band_run_default_order_equals_the_band_advance_form(orc):
    """library-default order without autoshift: the general machine gives what _band_begin / _band_advance give"""
    g = ctx()
    W, H, blur = 507, 525, 0.0
    edges = [0, 160, 352, H]
    rows = [(a, b - a) for a, b in zip(edges[:-1], edges[1:])]
    rng = np.random.default_rng(5)
    a_pps, b_pps = [gpu.PostProcess(g) for _ in rows], [gpu.PostProcess(g) for _ in rows]
    for batch, F in enumerate((4, 6)):
        fr = _frames(rng, F, W, H, 7 * batch)
        if batch == 0:
            fr[2] = 0.5
        want, winfo, _ = _run_bands(g, a_pps, rows, fr, blur)
        got, ginfo, _ = _run_bands_general(g, b_pps, edges, fr, motionblur=blur)
        assert np.array_equal(got, want)
        assert (want == 512.0).any()  # green lines
        for a, b in zip(winfo, ginfo):
            assert (a.lastmin, a.lastmax, a.dx, a.dy, a.stripx, a.stripy, a.locked) == (b.lastmin, b.lastmax, b.dx, b.dy, b.stripx, b.stripy, b.locked)


def test_general_band_run_with_the_pll_equals_the_oracle(orc):
    """PARAM_INT_FRAMERATE_PLL in band mode: the PLL is part of the replicated chain, so every rank reports the same nudge
    (syncdetector.c:133-153); one frame per run, like tsdrgpu_postproc_run's callers do it."""
    g = ctx()
    geo = orc.geometry(2_000_000, 131, 60.0)  # a geometry the library derives itself, so that a nudge keeps it consistent
    W, H = geo.width, 131
    edges = [0, 64, H]
    rng = np.random.default_rng(77)
    pps = [gpu.PostProcess(g) for _ in range(2)]
    single = gpu.PostProcess(g)
    opp = orc.PostProcess(geo)
    fired = 0
    rate = geo.refreshrate
    for k in range(14):
        if geo.width != W:  # the oracle applies the nudge to its geometry (TSDRLibrary.c:540-550): the frame size changed, stop like
            break           # tests/test_gpu_postproc.py::test_post_process_pll does
        fr = _frames(rng, 1, W, H, 3 * k)  # the pattern drifts: the sync detector sees a moving blanking interval
        want = opp.run(fr[0].reshape(-1).copy(), 0.0, 0.1, 0, 0, 0, 1, 0).reshape(H, W)
        d_full, d_out = g.to_device(fr.reshape(-1)), g.empty(W * H)
        sinfo = single.run(d_full, 1, W, H, d_out, pll=1)
        got, infos, _ = _run_bands_general(g, pps, edges, fr, pll=1)
        assert np.array_equal(got[0], want)
        a, b = sinfo[0], infos[0]
        assert (a.dx, a.dy, a.locked, a.pll_fired, a.frameratediff, a.avg_speed) == (b.dx, b.dy, b.locked, b.pll_fired, b.frameratediff, b.avg_speed)
        fired += b.pll_fired
        rate -= b.frameratediff
        assert rate == geo.refreshrate, k  # the nudges the bands report are the reference's
        assert b.pll_fired == opp.state()[0][7]
    assert k >= 3


def test_general_band_run_config4_eight_bands_gui_order_with_autoshift(orc):
    """BASELINE configs[4]'s geometry (2962x2250, motion blur 15/16), EIGHT bands with bench.py's edges, the GUI's stage order
    (low-pass before sync) and autoshift on: the roll moves rows across all eight bands."""
    g = ctx()
    W, H, blur = 2962, 2250, 0.9375
    edges = [0] + list(_bench_cuts(H, 8)) + [H]
    rng = np.random.default_rng(8)
    pps = [gpu.PostProcess(g) for _ in range(8)]
    opp = orc.PostProcess(_geo(orc, W, H))
    for batch, F in enumerate((3, 2)):
        fr = _frames(rng, F, W, H, 10 * batch)
        if batch == 0:
            fr[1] = 0.25
        want = np.stack([opp.run(fr[k].reshape(-1).copy(), blur, 0.1, 1, 0, 1, 0, 0).reshape(H, W) for k in range(F)])
        got, infos, kinds = _run_bands_general(g, pps, edges, fr, motionblur=blur, lowpass_before_sync=1, autoshift=1)
        assert gpu.BAND_ALLGATHER_F32 in kinds
        assert np.array_equal(got, want), (batch, int(np.sum(got != want)))


@pytest.mark.parametrize("kernel", ["sample_parallel", "pixel_groups"])
def test_band_resampler_chunks_longer_than_a_frame(orc, kernel):
    """A caller may poll more than a frame's worth of samples per dsp_resample_process call (the library polls 0.1 frame, but
    nothing in the API says so): a chunk's output then touches three or four frames, i.e. more band entries than the two per
    chunk the table used to be sized for (an advisor finding of round 3: the pinned staging slot was overrun).  Both forms of
    the band resampler — the sample-parallel kernel and the pixel-group kernel with its entry table (TSDRGPU_RS_GROUPS=1, read
    once per process: run in a child) — against the ORACLE's stream for the same chunking."""
    import subprocess
    import sys
    if kernel == "pixel_groups" and not os.environ.get("TSDRGPU_RS_GROUPS"):
        out = subprocess.run([sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-x",
                              os.path.abspath(__file__) + "::test_band_resampler_chunks_longer_than_a_frame[pixel_groups]"],
                             env=dict(os.environ, TSDRGPU_RS_GROUPS="1"), capture_output=True, text=True, timeout=300, cwd=os.path.dirname(os.path.dirname(HERE_ROOT)))
        assert out.returncode == 0 and "1 passed" in out.stdout, out.stdout[-1500:] + out.stderr[-500:]
        return
    from tempestsdr_amd import synth
    g = ctx()
    fs, h, fv = 2_000_000, 131, 60.0
    geo = orc.geometry(fs, h, fv)
    W, P = geo.width, geo.width * h
    chunk, nch = 90_000, 7           # 2.7 frames of pixels per chunk
    y0, rows = 32, 64
    iq = synth.synth_iq(fs, "640x480", fv, nch * chunk, seed=0x5EED000A)
    mag = orc.am_demod(iq)
    up, down = W * h * fv, float(fs)
    ors = orc.Resampler()
    want = np.concatenate([ors.process(mag[k * chunk:(k + 1) * chunk], up, down) for k in range(nch)])
    d_iq = g.to_device(iq)
    rs_band = gpu.Resampler(g)
    cap = want.size // P + 3
    d_band = g.empty(cap * rows * W)
    n, touched = rs_band.process_band(d_iq, 1, chunk, nch, up, down, W, h, y0, rows, 0, d_band, cap)
    assert n == want.size and touched == (n + P - 1) // P and touched >= 18
    got = d_band.download().reshape(cap, rows, W)
    for j in range(n // P):
        ref = want[j * P:(j + 1) * P].reshape(h, W)[y0:y0 + rows]
        assert np.array_equal(got[j], ref), (j, int(np.sum(got[j] != ref)))
