"""Pin the CPU restatement (oracle/tsdr_oracle.c) against the REAL reference
compiled from /root/reference into oracle/_ref (bit-exact unless stated).
Skipped where oracle/_ref is absent; tests/test_oracle_golden.py then carries
the same pins through committed golden vectors."""
import ctypes as C
import os

import numpy as np
import pytest

from tempestsdr_amd import synth

RNG = np.random.default_rng(1234)

GEOMS = [(8_000_000, 525, 60.0), (25_000_000, 806, 60.0), (100_000_000, 1125, 60.0),
         (12_600_000, 525, 60.0),  # r == 2.0 exactly: the aligned edge case
         (10_000_000, 625, 50.0), (8_000_000, 525, 59.94), (200_000_000, 2250, 60.004)]


@pytest.mark.parametrize("fs,h,fv", GEOMS)
def test_geometry(orc, ref, fs, h, fv):
    g = orc.geometry(fs, h, fv)
    t = ref.ref_new(h, fv, fs, 0.0, None)
    assert g.width == ref.ref_width(t)
    assert g.pixelrate == ref.ref_pixelrate(t)
    assert g.pixeltimeoversampletime == ref.ref_pixeltimeoversampletime(t)
    ref.ref_free(t)


def test_am_demod(orc, ref):
    iq = RNG.standard_normal(2 * 100_003).astype(np.float32)
    mine = orc.am_demod(iq)
    theirs = iq.copy()
    ref.complex_to_real(theirs, iq.size // 2)
    assert np.array_equal(mine, theirs[:iq.size // 2])


@pytest.mark.parametrize("fs,h,fv", GEOMS)
@pytest.mark.parametrize("nearest", [0, 1])
def test_resampler_chunks(orc, ref, fs, h, fv, nearest):
    g = orc.geometry(fs, h, fv)
    up, down = g.width * g.height * g.refreshrate, float(fs)
    chunk = orc.chunk_size(fs, fv)
    mine = orc.Resampler()
    theirs = ref.ref_resampler_new()
    st = np.zeros(2)
    for c in range(6):
        x = RNG.random(chunk).astype(np.float32)
        a = mine.process(x, up, down, nearest)
        out = np.zeros(a.size + 16, np.float32)
        n = ref.ref_resampler_process(theirs, x, x.size, out, up, down, nearest)
        assert n == a.size
        k = mine.last_emitted  # pixels beyond `emitted` are stale in the reference
        assert k >= n - 1
        assert np.array_equal(a[:min(k, n)], out[:min(k, n)])
        ref.ref_resampler_state(theirs, st)
        assert st[0] == mine.st.contrib and st[1] == mine.st.offset
    ref.ref_resampler_free(theirs)


@pytest.mark.parametrize("r", [0.37, 0.999, 1.0, 1.5, 3.25, 7.0])
def test_resampler_rates(orc, ref, r):
    mine = orc.Resampler()
    theirs = ref.ref_resampler_new()
    for c in range(5):
        x = RNG.random(1000 + 13 * c).astype(np.float32)
        a = mine.process(x, r * 1e6, 1e6)
        out = np.zeros(a.size + 16, np.float32)
        n = ref.ref_resampler_process(theirs, x, x.size, out, r * 1e6, 1e6, 0)
        assert n == a.size
        k = min(mine.last_emitted, n)
        assert np.array_equal(a[:k], out[:k])
    ref.ref_resampler_free(theirs)


def test_autogain_collapse_lowpass(orc, ref):
    w, h = 317, 203
    ag_m, ag_r = orc.Autogain(), orc.Autogain()
    orc.lib.orc_autogain_init(C.byref(ag_m))
    orc.lib.orc_autogain_init(C.byref(ag_r))
    scr_m = np.zeros(w * h, np.float32)
    scr_r = np.zeros(w * h, np.float32)
    for k in range(4):
        f = (RNG.random(w * h) * (1 + k)).astype(np.float32)
        f[5] = 512.0
        if k == 2:
            f[0] = 512.0  # the v[0]-before-sentinel-test quirk, dsp.c:50-57
        om, orf = np.empty_like(f), np.empty_like(f)
        orc.lib.orc_autogain_run(C.byref(ag_m), f.size, f, om, 0.1)
        ref.dsp_autogain_run(C.byref(ag_r), f.size, f, orf, 0.1)
        assert np.array_equal(om, orf)
        assert (ag_m.lastmax, ag_m.lastmin) == (ag_r.lastmax, ag_r.lastmin)
        assert ag_m.snr == ag_r.snr or (np.isnan(ag_m.snr) and np.isnan(ag_r.snr))
        cm, rm = np.empty(w, np.float32), np.empty(h, np.float32)
        cr, rr = np.empty(w, np.float32), np.empty(h, np.float32)
        orc.lib.orc_average_v_h(w, h, om, cm, rm)
        ref.dsp_average_v_h(w, h, orf, cr, rr)
        assert np.array_equal(cm, cr) and np.array_equal(rm, rr)
        orc.lib.orc_timelowpass_run(0.9375, f.size, om, scr_m)
        ref.dsp_timelowpass_run(0.9375, f.size, orf, scr_r)
        assert np.array_equal(scr_m, scr_r)


@pytest.mark.parametrize("n", [2, 3, 4, 5, 7, 806, 1033, 2962])
def test_gaussianblur(orc, ref, n):
    x = RNG.random(n).astype(np.float32) * 100
    a, b = x.copy(), x.copy()
    orc.lib.orc_gaussianblur(a, n)
    ref.gaussianblur(b, n)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("n,minsize,lp", [(1033, 51, 0.9), (806, 8, 0.1), (2962, 148, 0.9), (64, 1, 0.5)])
def test_findthesweetspot(orc, ref, n, minsize, lp):
    dm, dr = orc.Sweetspot(), orc.Sweetspot()
    for k in range(30):
        strip = (RNG.random(n) * 0.1 + 1.0).astype(np.float32) * 500
        lo = (37 * k + 11) % n  # a moving blanking band
        idx = (lo + np.arange(n // 9)) % n
        strip[idx] *= 0.1
        a, b = strip.copy(), strip.copy()
        orc.lib.orc_findthesweetspot(C.byref(dm), a, n, minsize, lp)
        ref.findthesweetspot(C.byref(dr), b, n, minsize, lp)
        assert (dm.dx, dm.vx, dm.absvx, dm.curr_stripsize) == (dr.dx, dr.vx, dr.absvx, dr.curr_stripsize)
        assert np.array_equal(a, b)


def _frames(fs, modename, fv, h, nframes, seed):
    g_w = int(2 * (fs / (fv * h)))
    n = int(nframes * fs / fv) + 10
    return synth.synth_iq(fs, modename, fv, n, seed=seed), g_w


@pytest.mark.parametrize("lbs,aap", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("autoshift,pll,mb", [(0, 0, 0.0), (1, 0, 0.5), (0, 1, 0.9375), (1, 1, 0.0)])
def test_post_process_orders(orc, ref, lbs, aap, autoshift, pll, mb):
    fs, h, fv = 2_000_000, 131, 60.0
    mode = (200, 131, 160, 120)
    g = orc.geometry(fs, h, fv)
    t = ref.ref_new(h, fv, fs, mb, None)
    ref.ref_setparam(t, 0, autoshift)
    ref.ref_setparam(t, 1, pll)
    iq = synth.synth_iq(fs, mode, 60.02, int(12.5 * fs / fv), seed=77)
    pix, _ = orc.demod_resample_stream(iq, g)
    pp = orc.PostProcess(g)
    si, sd = np.zeros(10, np.int32), np.zeros(4)
    pos = 1234
    for k in range(10):
        w = g.width
        assert w == ref.ref_width(t)
        n = w * h
        frame = pix[pos:pos + n].copy()
        pos += n
        fr = frame.copy()
        mine = pp.run(frame, mb, 0.1, lbs, aap, autoshift, pll, 0)
        p = ref.ref_post_process(t, fr, mb, 0.1, lbs, aap)
        theirs = np.ctypeslib.as_array(p, shape=(n,))
        assert np.array_equal(mine, theirs), f"frame {k}"
        mi, md = pp.state()
        ref.ref_postprocess_state(t, si, sd)
        assert np.array_equal(mi[:7], si[:7]) and mi[9] == si[9]
        assert np.array_equal(md, sd, equal_nan=True)
        assert g.refreshrate == ref.ref_refreshrate(t)
        c1, r1 = pp.strips()
        assert np.array_equal(c1, np.ctypeslib.as_array(ref.ref_colsum(t), shape=(w,)))
        assert np.array_equal(r1, np.ctypeslib.as_array(ref.ref_rowsum(t), shape=(h,)))
    ref.ref_free(t)


@pytest.mark.parametrize("n", [2, 8, 1024, 1 << 14])
@pytest.mark.parametrize("inverse", [0, 1])
def test_fft_perform(orc, ref, n, inverse):
    z = RNG.standard_normal(2 * n).astype(np.float32)
    a = orc.fft_perform(z, inverse)
    b = z.copy()
    ref.fft_perform(b, n, inverse)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("size", [1000, 4096, 45_090])
def test_autocorrelation_accumulate(orc, ref, size):
    fs = 800_000
    ac = orc.Autocorr(fs)
    fr = np.zeros(ac.flen)
    ln = np.zeros(ac.llen)
    for k in range(3):
        x = RNG.random(size).astype(np.float32)
        mine = ac.run(x)
        theirs = np.zeros(2 * size, np.float32)
        ref.fft_autocorrelation(theirs, x, size)
        assert np.array_equal(mine, theirs)
        if size > 2 * (ac.flo + ac.flen):
            ref.ref_accumulate(fr, theirs, ac.flo, ac.flen, k + 1)
            ref.ref_accumulate(ln, theirs, ac.llo, ac.llen, k + 1)
            assert np.array_equal(fr, ac.frame) and np.array_equal(ln, ac.line)


def test_lag_windows_and_capture(orc):
    # SURVEY §8 table
    assert orc.lag_windows(8_000_000) == (91954, 53500, 61, 185)
    assert orc.lag_windows(25_000_000) == (287356, 167189, 191, 579)
    assert orc.lag_windows(100_000_000) == (1149425, 668756, 766, 2315)
    assert orc.capture_size(8_000_000) == 450909
    assert orc.capture_size(100_000_000) == 5636363


def test_superb_stitch(orc, ref):
    fs, fv = 400_000, 60.0
    sif = int(fs / fv)
    gathered = 10 * sif
    base = RNG.standard_normal(2 * gathered + 4000).astype(np.float32)
    hops = []
    for i in range(4):
        sh = 2 * (37 * i)
        hp = base[sh:sh + 2 * gathered].copy()
        hp += RNG.standard_normal(hp.size).astype(np.float32) * 0.05
        hops.append(hp)
    mine, offs = orc.superb_stitch(hops, sif)
    t = ref.ref_new(100, fv, fs, 0.0, None)
    rh = [h.copy() for h in hops]
    ptrs = (C.c_void_p * 4)(*[h.ctypes.data for h in rh])
    out = np.zeros(mine.size, np.float32)
    n = ref.ref_superb_stitch(t, ptrs, 4, gathered, sif, fs, out)
    assert 2 * n == mine.size
    assert np.array_equal(mine, out)
    ref.ref_free(t)


@pytest.mark.parametrize("block", [1000, 532350])
def test_dropped_shift(orc, ref, block):
    diff_m, diff_r = 0, C.c_int64(0)
    for s in [0, 5, -3, 12345, -99999, 7 * block, -7 * block - 1, 1 << 33, -(1 << 33) + 5]:
        diff_m = orc.lib.orc_dropped_shift_with(diff_m, block, s)
        ref.dsp_dropped_compensation_shift_with(C.byref(diff_r), block, s)
        assert diff_m == diff_r.value


# ---------------------------------------------------------------------------
# f3: frame -> packed RGB.  The reference is the JNI shim's pixel loop (JavaGUI/jni/TSDRLibraryNDK.c:222-276),
# compiled here from its own source behind oracle/jni_stub/jni.h and driven through its frame callback read_async().
# ---------------------------------------------------------------------------
def _rgb_frame(rng, n):
    v = rng.random(n).astype(np.float32) * np.float32(1.6) - np.float32(0.3)  # below 0, inside (0,1], above 1
    idx = rng.choice(n, size=min(n, 64), replace=False)
    special = np.array([256.0, 512.0, 1024.0, 2048.0, 0.0, 1.0, -0.0, np.nextafter(np.float32(1.0), np.float32(2.0)),
                        255.9999 / 255.0, 1e-30, np.inf, -np.inf, 300.0, 2047.0, 2049.0, 0.5], np.float32)
    v[idx] = special[np.arange(idx.size) % special.size]
    return v


@pytest.mark.parametrize("inverted", [0, 1])
def test_frame_to_rgb_equals_the_reference_jni_loop(orc, inverted):
    if not orc.have_ref_jni():
        pytest.skip("oracle/_ref/libtsdr_ref_jni.so not built (no /root/reference on this box)")
    jni = orc.ref_jni()
    jni.ref_jni_reset()
    rng = np.random.default_rng(40 + inverted)
    prev_ref = prev_orc = None
    for (w, h) in [(64, 48), (64, 48), (507, 525), (507, 525), (33, 7)]:  # a repeated size keeps the viewer's buffer
        n = w * h
        frame = _rgb_frame(rng, n)
        got_ref = np.zeros(n, np.int32)
        jni.ref_jni_frame_to_rgb(frame, w, h, inverted, got_ref)
        # the oracle converts into a caller buffer: transparent pixels keep what it holds, which for the GUI is the
        # previous frame of the same size (a fresh, malloc'ed buffer after a size change: those pixels are undefined
        # in the reference, so they are excluded from the comparison for the first frame of a size)
        if prev_orc is not None and prev_orc.size == n:
            buf = prev_orc.copy()
            defined = np.ones(n, bool)
        else:
            buf = np.zeros(n, np.int32)
            defined = frame != np.float32(2048.0)
        orc.lib.orc_frame_to_rgb(frame, buf, n, inverted)
        assert np.array_equal(buf[defined], got_ref[defined])
        # carry the REFERENCE's buffer forward so that undefined pixels cannot leak into the next comparison
        prev_orc = got_ref.copy()
        prev_ref = got_ref


def test_frame_to_rgb_every_gray_level(orc):
    """(int)(v*255.0f) over a fine sweep of (0, 1]: every one of the 256 levels and their boundaries."""
    if not orc.have_ref_jni():
        pytest.skip("oracle/_ref/libtsdr_ref_jni.so not built")
    jni = orc.ref_jni()
    jni.ref_jni_reset()
    k = np.arange(1, 256 * 4096 + 1, dtype=np.float64) / (256 * 4096)
    frame = k.astype(np.float32)
    edges = (np.arange(1, 256, dtype=np.float32) / np.float32(255.0))
    frame = np.concatenate([frame, edges, np.nextafter(edges, np.float32(0)), np.nextafter(edges, np.float32(2))]).astype(np.float32)
    n = frame.size
    for inverted in (0, 1):
        a = np.zeros(n, np.int32)
        b = np.zeros(n, np.int32)
        jni.ref_jni_frame_to_rgb(frame, n, 1, inverted, a)
        orc.lib.orc_frame_to_rgb(frame, b, n, inverted)
        assert np.array_equal(a, b)
        assert len(np.unique(a & 255)) == 256


def test_threaded_reference_library_is_the_oracle_applied_to_what_it_kept(ref):
    """The reference's THREADED library (tsdr_readasync behind its RawFile plugin, BASELINE configs[0]) against the oracle, end to
    end.  Its frames are not the deterministic driver's — its rings refuse plugin blocks, chunks of pixels and finished frames while
    they grow (circbuff.c:64-110), each refusal compensated so that the frame grid survives (dsp.c:313-368) — but
    scripts/diag_cfg0_replay.py recovers WHAT it kept from the delivered frames themselves (every frame decomposed into exact affine
    images of raw driver frames; refused plugin blocks land on block boundaries) and replays the oracle's resampler and
    dsp_post_process over exactly that: every delivered frame must come out BIT FOR BIT.  What is lost is timing dependent, so a run
    whose pattern the decomposition does not cover is skipped, not failed; a replay that covers it and differs would be a defect of
    the restatement."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "scripts", "diag_cfg0_replay.py"), "40"], capture_output=True, text=True, timeout=900, cwd=root)
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("REPLAY:")]
    if out.returncode != 0 or not line:
        pytest.skip("this run's loss pattern is outside what the decomposition models: " + out.stdout[-300:] + out.stderr[-300:])
    import re
    m = re.search(r"reproduce (\d+) of (\d+) delivered frames", line[0])
    same, n = int(m.group(1)), int(m.group(2))
    # the one outcome that is the restatement's fault and no loss pattern's: a frame that follows directly on reproduced frames,
    # nothing undelivered in between, and differs (the script tells it apart from "lost in a way the search does not model")
    assert "REPLAY-DEFECT" not in out.stdout, out.stdout[-1500:]
    if same != n and "not reproduced" in out.stdout:
        pytest.skip("frames lost in a way the search does not model (a post-processed composite): " + line[0])
    if n < 30:
        pytest.skip(f"only {n} frames delivered in this run (a loaded host)")
    assert same == n, out.stdout[-1500:]


def test_threaded_reference_library_plots_are_the_oracles_running_mean(ref):
    """The detector side of the same session (scripts/diag_cfg0_plots.py): every plot update the threaded library announces is
    BIT-identical to the oracle's running mean over the capture windows its detector took — consecutive stretches of the plugin
    blocks its ring accepted, restarting at a block boundary after a purge (frameratedetector.c:128-187,215-230).  Timing decides
    which windows those are; a run the placement search does not cover is skipped."""
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "scripts", "diag_cfg0_plots.py")], capture_output=True, text=True, timeout=600, cwd=root)
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("PLOTS:")]
    if out.returncode != 0 or not line:
        pytest.skip("no result: " + out.stdout[-300:] + out.stderr[-300:])
    m = re.search(r"PLOTS: (\d+) of (\d+) plot updates", line[0])
    same, n = int(m.group(1)), int(m.group(2))
    # The detector's FIRST window starts at one of the first plugin blocks whatever the timing (the search tries 12 block starts), so a
    # session that announced plots and whose first update no placement reproduces is the restatement's fault; later updates can follow
    # a purge longer than the search covers (skipped, not failed).
    if n >= 1:
        assert same >= 1, out.stdout[-1500:]
    if same != n:
        pytest.skip("a window placement outside the search: " + out.stdout[-400:])
    if n < 5:
        pytest.skip(f"only {n} plot updates in this run (a loaded host)")
    assert same == n


@pytest.mark.parametrize("h", [1, 2, 3, 7, 40])
@pytest.mark.parametrize("w", [1, 2, 3, 4, 5, 9])
def test_post_process_degenerate_geometries(orc, ref, w, h):
    """frames of one pixel, one row, one column ...: what a host gets when it sets a resolution the stream cannot fill
    (width = (int)(2 fs / (fv h)), TSDRLibrary.c:543-546).  The compiled reference neither crashes nor leaves the restatement:
    bit-identical frames in every stage order, which makes the oracle the yardstick for the GPU path at these sizes too
    (tests/test_gpu_edges.py::test_degenerate_frame_geometries)."""
    fv = 10.0
    fs = next((f for f in range(max(1, int(w * fv * h / 2) - 2), int(w * fv * h / 2) + 40) if int(2 * (f / (fv * h))) == w), None)
    assert fs is not None
    rng = np.random.default_rng(100 * w + h)
    for lbs, aap, ash, mb in ((0, 0, 0, 0.0), (1, 0, 1, 0.5), (0, 1, 0, 0.25), (1, 1, 1, 0.9)):
        g = orc.geometry(fs, h, fv)
        assert g.width == w
        t = ref.ref_new(h, fv, fs, mb, None)
        ref.ref_setparam(t, 0, ash)
        pp = orc.PostProcess(g)
        n = w * h
        for k in range(8):
            fr = (rng.random(n) * 2 - 0.5).astype(np.float32)
            mine = pp.run(fr.copy(), mb, 0.1, lbs, aap, ash, 0, 0)
            theirs = np.ctypeslib.as_array(ref.ref_post_process(t, fr.copy(), mb, 0.1, lbs, aap), shape=(n,))
            assert np.array_equal(mine, theirs, equal_nan=True), (lbs, aap, ash, mb, k)
        ref.ref_free(t)


@pytest.mark.parametrize("w,h", [(9, 7), (40, 33)])
def test_post_process_nonfinite_frames(orc, ref, w, h):
    """frames with a few NaN pixels, an all-NaN frame, infinities — what a source that hands over non-finite samples produces: the
    compiled reference and the restatement stay bit-identical (NaN positions included) through every stage order, the frames
    after the bad ones too (the autogain and the sync detector carry the damage).  The yardstick for the GPU path's behaviour
    on such input (its sync search then finds no window to choose: sync_decide keeps window 0 like syncdetector.c:36-38,52-55)."""
    fv = 10.0
    fs = next(f for f in range(max(1, int(w * fv * h / 2) - 2), int(w * fv * h / 2) + 40) if int(2 * (f / (fv * h))) == w)
    rng = np.random.default_rng(9 * w + h)
    for lbs, aap, ash, mb in ((0, 0, 0, 0.0), (1, 0, 1, 0.5), (0, 1, 0, 0.25), (1, 1, 1, 0.9)):
        g = orc.geometry(fs, h, fv)
        t = ref.ref_new(h, fv, fs, mb, None)
        ref.ref_setparam(t, 0, ash)
        pp = orc.PostProcess(g)
        n = w * h
        for k in range(8):
            fr = (rng.random(n) * 2 - 0.5).astype(np.float32)
            if k in (2, 3):
                fr[rng.integers(0, n, 3)] = np.nan
            if k == 4:
                fr[:] = np.nan
            if k == 5:
                fr[rng.integers(0, n, 2)] = np.inf
            mine = pp.run(fr.copy(), mb, 0.1, lbs, aap, ash, 0, 0)
            theirs = np.ctypeslib.as_array(ref.ref_post_process(t, fr.copy(), mb, 0.1, lbs, aap), shape=(n,))
            assert np.array_equal(mine, theirs, equal_nan=True), (lbs, aap, ash, mb, k)
        ref.ref_free(t)
