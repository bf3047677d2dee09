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


def test_plot_columns_equal_the_reference_bytecode():
    """f4 pinned: the device's columns == what the reference's own PlotVisualizer.populateData / ZoomableXScale classes
    computed (tests/golden/java_fixtures_jvm.json: the released jar executed by tests/golden/minijvm.py), for default
    scales and for zoom states reached through the scale's public zoom / drag methods."""
    import hashlib
    import json
    import os
    from tempestsdr_amd import gpu as G
    fx = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "java_fixtures_jvm.json")))
    g = ctx()
    cases = fx["populate"] + fx["populate_zoomed"]
    assert len(cases) == 42
    for case in cases:
        r = np.random.default_rng(case["seed"])
        size, nwidth = case["size"], case["nwidth"]
        data = r.random(size) + 0.2 * np.sin(np.arange(size) / 37.0)
        data[r.integers(0, size, 3)] = 1.5
        d = g.empty(2 * size, np.float32)
        g._ck(g.lib.tsdrgpu_upload(g.h, d.ptr, data.ctypes.data, data.nbytes))
        g.sync()
        s = G.PlotScale()
        for k, v in case["scale"].items():
            setattr(s, k, v)
        vis, lo, hi, mi = g.plot_columns(d.ptr, size, nwidth, s)
        assert (lo, hi, mi) == (case["lowest"], case["highest"], case["max_index"]), (size, nwidth, case.get("actions"))
        assert hashlib.sha256(np.asarray(vis, np.float64).tobytes()).hexdigest() == case["visdata_sha"], (size, nwidth, case.get("actions"))


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


def test_gather2_appends_blocks_back_to_back():
    """tsdrgpu_gather2: n device blocks (ragged sizes, aligned and misaligned sources / destinations) land behind each
    other in both destinations; one destination may be left out.  Byte-exact."""
    import ctypes as C
    g = ctx()
    rng = np.random.default_rng(12)
    sizes = [524288, 4, 1000003, 8, 262145, 12, 65536] + [4096] * 25  # 32 blocks = the most one launch takes
    srcs, host = [], []
    for i, n in enumerate(sizes):
        h = rng.random(n + 3).astype(np.float32)
        d = g.to_device(h)
        srcs.append((d, i % 3))  # element offset 0..2 into the allocation: 4-byte aligned sources too
        host.append(h[i % 3:i % 3 + n])
    want = np.concatenate(host)
    for dst_off, second in ((0, True), (1, True), (0, False)):
        d1 = g.empty(want.size + 8)
        d2 = g.empty(want.size + 8)
        d1.zero()
        d2.zero()
        ptrs = (C.c_void_p * len(sizes))(*[d.at(o) for d, o in srcs])
        nbytes = (C.c_size_t * len(sizes))(*[4 * n for n in sizes])
        g._ck(g.lib.tsdrgpu_gather2(g.h, d1.at(dst_off), d2.at(dst_off) if second else None, ptrs, nbytes, len(sizes)))
        g.sync()
        a = d1.download()
        assert np.array_equal(a[dst_off:dst_off + want.size], want) and not a[:dst_off].any() and not a[dst_off + want.size:].any()
        b = d2.download()
        if second:
            assert np.array_equal(b, a)
        else:
            assert not b.any()
    # bad arguments are refused, nothing is launched
    ptrs = (C.c_void_p * 1)(srcs[0][0].ptr)
    nb = (C.c_size_t * 1)(6)
    assert g.lib.tsdrgpu_gather2(g.h, d1.ptr, None, ptrs, nb, 1) != 0
    assert g.lib.tsdrgpu_gather2(g.h, d1.ptr, None, ptrs, nb, 33) != 0


def test_autocorr_plots_snapshot_is_a_device_copy_of_the_plots():
    import ctypes as C
    from tempestsdr_amd import gpu
    g = ctx()
    ac = gpu.Autocorr(g, 8_000_000)
    rng = np.random.default_rng(3)
    x = rng.random(2 * ac.capture).astype(np.float32)
    ac.run(g.to_device(x), 0, ac.capture, 2)
    f, l, calls = ac.plots()
    snap, n = C.c_void_p(), C.c_uint64()
    g._ck(g.lib.tsdrgpu_autocorr_plots_snapshot(ac.h, C.byref(snap), C.byref(n)))
    g.sync()
    out = np.empty(f.size + l.size, np.float64)
    g._ck(g.lib.tsdrgpu_download(g.h, out.ctypes.data, snap.value, out.nbytes))
    g.sync()
    assert n.value == calls == 2
    assert np.array_equal(out[:f.size], f) and np.array_equal(out[f.size:], l)


@pytest.mark.parametrize("n,sentinels", [(1, False), (2, False), (507 * 525, False), (2962 * 1125, True), (1033 * 806, True)])
def test_frame_snr_matches_dsp_autogain_run(orc, n, sentinels):
    """dsp_autogain_t.snr (dsp.c:69-93), the by-product the post-processing run leaves out: on demand, against the
    oracle's (pinned) sequential f64 loop.  Tolerance 1e-9 relative: tree sums instead of a sequential sum."""
    import ctypes as C
    g = ctx()
    rng = np.random.default_rng(n)
    x = (rng.random(n).astype(np.float32) * np.float32(3.0) + np.float32(0.25))
    if sentinels:
        x[rng.integers(0, n, 50)] = np.float32(512.0)
        x[rng.integers(0, n, 50)] = np.float32(-512.0)
    ag = orc.Autogain()
    orc.lib.orc_autogain_init(C.byref(ag))
    out = np.empty(n, np.float32)
    orc.lib.orc_autogain_run(C.byref(ag), n, x, out, np.float32(0.1))
    got = C.c_float()
    d_x = g.to_device(x)
    g._ck(g.lib.tsdrgpu_frame_snr(g.h, d_x.ptr, n, C.byref(got)))
    want = np.float32(ag.snr)
    if np.isnan(want) or np.isinf(want):
        assert np.isnan(got.value) == np.isnan(want) and np.isinf(got.value) == np.isinf(want)
    else:
        assert abs(got.value - want) <= 1e-9 * abs(want) + np.spacing(want)


def _device_check_binary(name):
    """scripts/micro/<name>: built for gfx950 by tempestsdr_amd.build.build_checks() (the GPU boxes have no hipcc, so the
    binary travels with the snapshot like the libraries); rebuilt here when a compiler is present and the binary is not."""
    import os
    import pytest
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "scripts", "micro", name + ".hip")
    exe = os.path.join(root, "scripts", "micro", name)
    if not os.path.exists(src):
        pytest.skip("scripts/micro/%s.hip is not in this tree" % name)
    if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(src):
        from tempestsdr_amd import build as b
        if os.path.exists(b.HIPCC):
            b.build_checks(verbose=False)
    assert os.path.exists(exe), "scripts/micro/%s was not built (python -m tempestsdr_amd.build) and there is no hipcc here" % name
    return exe


def test_wave_reductions_equal_the_shuffle_tree():
    """tempestsdr_amd/csrc/wave_reduce.h (permlane swaps + DPP row shifts) must give lane 0 the shuffle tree's result bit
    for bit — the partial sums of the frame statistics, and with them the strips, depend on the order of additions."""
    import subprocess
    exe = _device_check_binary("wave_reduce_check")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    print(r.stdout)
    assert r.returncode == 0 and "0 mismatches" in r.stdout, r.stdout + r.stderr


def test_division_and_square_root_on_the_device():
    """scripts/micro/arith_check.hip: on this GPU the compiler's f32 `/` and sqrtf are the host's (correctly rounded) results on
    the hard cases — divisors with an all-ones mantissa under powers of two, every float of two binades for the root — and the
    bare sequences the kernels run (NormDiv, demod1) give the same bits as `/` and sqrtf."""
    import subprocess
    exe = _device_check_binary("arith_check")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(r.stdout)
    assert r.returncode == 0 and r.stdout.strip().endswith("0 mismatches"), r.stdout + r.stderr
