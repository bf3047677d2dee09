"""SURVEY §8(f) rows, CPU side: the oracle's sample decode pinned against the
reference's RawFile plugin binary, and the mode-detection object against a
restatement of the Java GUI's logic."""
import ctypes as C
import math
import os
import sys

import numpy as np
import pytest

from tempestsdr_amd import build, gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RAW = os.path.join(ROOT, "oracle", "_ref", "libTSDRPlugin_RawFile_bench.so")
CB = C.CFUNCTYPE(None, C.POINTER(C.c_float), C.c_uint64, C.c_void_p, C.c_int64)


@pytest.mark.parametrize("fmt,dtype,tid", [("int8", np.int8, 1), ("int16", np.int16, 2), ("uint8", np.uint8, 3),
                                            ("uint16", np.uint16, 4), ("float", np.float32, 0)])
def test_oracle_decode_equals_rawfile_plugin(orc, tmp_path, fmt, dtype, tid):
    """TSDRPlugin_RawFile.c:241-261 through the compiled plugin itself (free-running build)."""
    if not os.path.exists(RAW):
        pytest.skip("oracle/_ref not built")
    n = 512 * 1024  # one plugin block (SAMPLES_TO_READ_AT_ONCE)
    rng = np.random.default_rng(3)
    if dtype == np.float32:
        raw = rng.standard_normal(n).astype(np.float32)
    else:
        info = np.iinfo(dtype)
        raw = rng.integers(info.min, info.max + 1, n).astype(dtype)
        raw[:4] = [info.min, info.max, 0 if info.min < 0 else 128, 1]
    path = tmp_path / f"s.{fmt}"
    raw.tofile(path)
    plug = C.CDLL(RAW)
    plug.tsdrplugin_init.argtypes = [C.c_char_p]
    plug.tsdrplugin_readasync.argtypes = [CB, C.c_void_p]
    params = C.create_string_buffer(f"{path} 8000000 {fmt}".encode())  # the plugin tokenises in place
    assert plug.tsdrplugin_init(params) == 0
    got = []

    def cb(buf, items, ctx, dropped):
        if not got:
            got.append(np.ctypeslib.as_array(buf, shape=(items,)).copy())
        plug.tsdrplugin_stop()

    assert plug.tsdrplugin_readasync(CB(cb), None) == 0
    want = np.empty(n, np.float32)
    orc.lib.orc_decode_samples(raw.ctypes.data, tid, want, n)
    assert np.array_equal(got[0], want)


def java_round(x):
    return int(math.floor(x + 0.5))


def test_modedetect_follows_the_gui_logic():
    build.build(verbose=False)
    md = gpu.ModeDetect()
    fs = 100_000_000
    flo, llo = 1149425, 766
    # Main.onIncommingPlot: fps = fs/(offset+idx); height = round(frame_lag/line_lag); accepted when the same
    # (long)(fps*height) key has already been counted AUTO_FRAMERATE_CONVERGANCE_ITERATIONS = 3 times
    seq = [(517242, 715), (517242, 715), (517240, 715), (517242, 715), (517242, 715), (517242, 715)]
    counts = {}
    for (fi, li) in seq:
        d = md.feed(flo, fi, llo, li, fs)
        frame_lag, line_lag = flo + fi, llo + li
        fps = fs / frame_lag
        height = java_round(frame_lag / line_lag)
        key = int(fps * height)
        accepted = counts.get(key) == 3
        if not accepted:
            counts[key] = counts.get(key, 0) + 1
        assert (d.frame_lag, d.line_lag, d.height) == (frame_lag, line_lag, height)
        assert d.framerate == fps and d.linerate == fs / line_lag
        assert d.accepted == int(accepted) and d.seen == counts[key]
    assert d.accepted == 1
    assert (d.height, round(d.framerate, 3)) == (1125, 60.0)
    assert d.mode_name == b"1920x1080 @ 60Hz" and (d.mode_width, d.mode_height) == (2576, 1125)
    w = int(2 * (fs / (d.framerate * d.height)))  # TSDRLibrary.c:543-546
    assert d.pixelrate == w * d.height * d.framerate
    # no mode with that height: the closest height wins (VideoMode.java:177-187)
    md.reset()
    d = md.feed(0, 1_000_000, 0, 1000, fs)  # height 1000 exists (1280x960 @ 60Hz); fps 100
    assert d.mode_name == b"1280x960 @ 60Hz"
    d = md.feed(0, 2_000_000, 0, 1999, fs)  # height 1001 -> closest height 1000 / 1002
    assert d.mode_height in (1000, 1002)
    with pytest.raises(gpu.TsdrGpuError):
        md.feed(0, 0, 0, 0, fs)


# --------------------------------------------------------------------------
# f4: plot decimation (PlotVisualizer.populateData).  No JDK in the image, so the oracle's C restatement
# is cross-checked against a second, column-oriented formulation of the same Java loop.
# --------------------------------------------------------------------------
def _zoom_state(size, nwidth, zoom=1.0, offset_px=0):
    """ZoomableXScale after setMinMaxValue(0,size), setMaxPixels(nwidth), zoom, setPxOffset (ZoomableXScale.java)."""
    from oracle.oracle import PlotScale
    s = PlotScale()
    span = float(size) * zoom
    s.one_val_in_pixels = nwidth / span
    s.one_px_in_values = span / nwidth
    s.offset_px = offset_px
    s.offset_val = offset_px * s.one_px_in_values
    s.min_value = 0.0
    return s


def _populate_by_columns(data, nwidth, s):
    size = data.size
    first = int(min(max(0 * s.one_px_in_values + s.offset_val + s.min_value, 0), size))
    last = int(min(max(nwidth * s.one_px_in_values + s.offset_val + s.min_value + 1, 0), size))
    ids = np.arange(first, last)
    px = ((ids - s.min_value) * s.one_val_in_pixels).astype(np.int64) - s.offset_px  # (int) truncates toward zero
    inside = (px >= 0) & (px < nwidth)
    cols = {}
    for i, p in zip(ids[inside], px[inside]):
        cols[p] = max(cols.get(p, -np.inf), data[i])
    start = data[min(first, size - 1)]
    flushed = []
    vis = np.empty(nwidth)
    cur, cur_px = start, 0
    for p in sorted(cols):
        if p == cur_px:
            cur = max(cur, cols[p])
        else:
            flushed.append(cur)
            vis[cur_px:p] = cur
            cur, cur_px = cols[p], p
    vis[cur_px:] = cur
    hi = max([data[0]] + flushed)
    lo = min([data[0]] + flushed)
    mi = 0
    if last > first:
        k = first + int(np.argmax(data[first:last]))
        if data[k] > data[0]:
            mi = k
    return vis, lo, hi, mi


@pytest.mark.parametrize("size,nwidth,zoom,offpx", [(5000, 800, 1.0, 0), (668756, 1237, 1.0, 0), (2315, 640, 1.0, 0),
                                                     (300, 800, 1.0, 0), (5000, 800, 0.13, 411), (5000, 800, 0.01, 37000),
                                                     (1, 16, 1.0, 0), (977, 977, 1.0, 0), (5000, 800, 1.0, -50)])
def test_oracle_plot_populate_vs_column_formulation(orc, size, nwidth, zoom, offpx):
    rng = np.random.default_rng(size + nwidth)
    data = rng.random(size) + 0.2 * np.sin(np.arange(size) / 37.0)
    data[rng.integers(0, size, 3)] = 1.5  # exact ties for the argmax
    s = _zoom_state(size, nwidth, zoom, offpx)
    vis, lo, hi, mi = orc.plot_populate(data, nwidth, s)
    vis2, lo2, hi2, mi2 = _populate_by_columns(data, nwidth, s)
    assert np.array_equal(vis, vis2)
    assert (lo, hi, mi) == (lo2, hi2, mi2)


def test_plotscale_default_matches_oracle(orc):
    lib = gpu.load_library()
    for size, nwidth in [(668756, 1237), (5, 800), (2315, 640)]:
        a, b = gpu.PlotScale(), orc.PlotScale()
        lib.tsdrgpu_plotscale_default(size, nwidth, C.byref(a))
        orc.lib.orc_plotscale_default(size, nwidth, C.byref(b))
        assert [getattr(a, f[0]) for f in a._fields_] == [getattr(b, f[0]) for f in b._fields_]


@pytest.mark.parametrize("inverted", [0, 1])
def test_oracle_frame_to_rgb_vs_vectorised_formulation(orc, inverted):
    """f3: the JNI shim's per-pixel branch chain (TSDRLibraryNDK.c:222-276; needs jni.h, cannot be built here)
    restated in C by the oracle, cross-checked against a vectorised numpy formulation."""
    rng = np.random.default_rng(8)
    n = 20_000
    fr = (rng.random(n) * 1.4 - 0.2).astype(np.float32)
    fr[rng.integers(0, n, 50)] = 256.0
    fr[rng.integers(0, n, 50)] = 512.0
    fr[rng.integers(0, n, 50)] = 1024.0
    fr[rng.integers(0, n, 50)] = 2048.0
    fr[rng.integers(0, n, 20)] = np.float32(1.0)
    fr[rng.integers(0, n, 20)] = np.float32(0.0)
    fr[rng.integers(0, n, 20)] = np.float32(300.0)
    prev = rng.integers(0, 1 << 24, n).astype(np.int32)
    got = prev.copy()
    orc.lib.orc_frame_to_rgb(fr, got, n, inverted)
    g8 = (fr * np.float32(255.0)).astype(np.int32)  # C truncation for the values that take this branch
    gray = np.where(inverted, 255 - g8, g8)
    gray = gray | (gray << 8) | (gray << 16)
    want = np.where((fr > 0) & (fr <= 1), gray,
           np.where(fr <= 0, 0xFFFFFF if inverted else 0,
           np.where(fr == 256.0, 255 << 16,
           np.where(fr == 512.0, 255 << 8,
           np.where(fr == 1024.0, 255,
           np.where(fr == 2048.0, prev, 0 if inverted else 0xFFFFFF)))))).astype(np.int32)
    assert np.array_equal(got, want)


# --------------------------------------------------------------------------
# Fixtures from the literal Java transliterations (tests/golden/make_java_fixtures.py -> java_fixtures.json).
# The image has no JVM; the pin for these rows is further down: the reference's released jar executed by the
# bytecode interpreter of tests/golden/minijvm.py (java_fixtures_jvm.json), which the transliteration must equal.
# --------------------------------------------------------------------------
def _java_fixtures():
    import json
    return json.load(open(os.path.join(ROOT, "tests", "golden", "java_fixtures.json")))


def _fixture_plot(case):
    r = np.random.default_rng(case["seed"])
    size = case["size"]
    data = r.random(size) + 0.2 * np.sin(np.arange(size) / 37.0)
    data[r.integers(0, size, 3)] = 1.5
    return data


def test_oracle_plot_populate_equals_java_transliteration(orc):
    import hashlib
    from oracle.oracle import PlotScale
    fx = _java_fixtures()
    assert len(fx["populate"]) >= 10
    for case in fx["populate"]:
        data = _fixture_plot(case)
        s = PlotScale()
        for k, v in case["scale"].items():
            setattr(s, k, v)
        vis, lo, hi, mi = orc.plot_populate(data, case["nwidth"], s)
        assert (lo, hi, mi) == (case["lowest"], case["highest"], case["max_index"]), case["size"]
        assert hashlib.sha256(np.asarray(vis, np.float64).tobytes()).hexdigest() == case["visdata_sha"]
        assert list(vis[:8]) == case["visdata_head"] and list(vis[-4:]) == case["visdata_tail"]


def test_modedetect_equals_java_transliteration():
    build.build(verbose=False)
    fx = _java_fixtures()
    assert fx["n_modes"] == 80
    for run in fx["modedetect"]:
        md = gpu.ModeDetect()
        for step in run["steps"]:
            fo, fi, lo, li = step["in"]
            d = md.feed(fo, fi, lo, li, run["samplerate"])
            want = step["out"]
            assert (d.framerate, d.height, d.linerate) == (want["fps"], want["height"], want["linerate"])
            assert (d.accepted, d.seen) == (want["accepted"], want["seen"])
            assert d.mode_id == want["mode"] and d.mode_name.decode() == want["mode_name"]


# --------------------------------------------------------------------------
# The pin of the Java rows (f2 mode detection, f4 plot decimation): tests/golden/java_fixtures_jvm.json is what the
# reference's OWN compiled classes (Release/JavaGUI/JTempestSDR.jar: ZoomableXScale, PlotVisualizer.populateData,
# Main.onIncommingPlot and its transformers, VideoMode) computed when executed by tests/golden/minijvm.py.
# --------------------------------------------------------------------------
def _jvm_fixtures():
    import json
    return json.load(open(os.path.join(ROOT, "tests", "golden", "java_fixtures_jvm.json")))


def test_transliteration_equals_the_reference_bytecode():
    a, b = _java_fixtures(), _jvm_fixtures()
    assert a["n_modes"] == b["n_modes"] == 80
    assert a["populate"] == b["populate"]
    assert a["closest_mode"] == b["closest_mode"]
    assert len(a["modedetect"]) == len(b["modedetect"])
    for ra, rb in zip(a["modedetect"], b["modedetect"]):
        assert ra["samplerate"] == rb["samplerate"] and len(ra["steps"]) == len(rb["steps"])
        for sa, sb in zip(ra["steps"], rb["steps"]):
            want = dict(sa["out"])
            want.pop("linerate")  # the GUI never computes it
            assert sa["in"] == sb["in"] and want == sb["out"]


def test_oracle_plot_populate_equals_the_reference_bytecode(orc):
    import hashlib
    from oracle.oracle import PlotScale
    fx = _jvm_fixtures()
    assert len(fx["populate"]) == 10 and len(fx["populate_zoomed"]) == 32
    for case in fx["populate"] + fx["populate_zoomed"]:
        data = _fixture_plot(case)
        s = PlotScale()
        for k, v in case["scale"].items():
            setattr(s, k, v)
        vis, lo, hi, mi = orc.plot_populate(data, case["nwidth"], s)
        assert (lo, hi, mi) == (case["lowest"], case["highest"], case["max_index"]), (case["size"], case.get("actions"))
        assert hashlib.sha256(np.asarray(vis, np.float64).tobytes()).hexdigest() == case["visdata_sha"], (case["size"], case.get("actions"))


def test_plotscale_default_equals_the_reference_bytecode(orc):
    lib = gpu.load_library()
    for case in _jvm_fixtures()["plotscale_default"]:
        a, b = gpu.PlotScale(), orc.PlotScale()
        lib.tsdrgpu_plotscale_default(case["size"], case["nwidth"], C.byref(a))
        orc.lib.orc_plotscale_default(case["size"], case["nwidth"], C.byref(b))
        for k, v in case["scale"].items():
            assert getattr(a, k) == v and getattr(b, k) == v, (case["size"], case["nwidth"], k)


def test_modedetect_equals_the_reference_bytecode():
    build.build(verbose=False)
    fx = _jvm_fixtures()
    table = fx["mode_table"]
    accepted = 0
    for run in fx["modedetect"] + fx["modedetect_random"]:
        md = gpu.ModeDetect()
        for step in run["steps"]:
            fo, fi, lo, li = step["in"]
            d = md.feed(fo, fi, lo, li, run["samplerate"])
            want = step["out"]
            assert (d.framerate, d.height) == (want["fps"], want["height"])
            assert (d.accepted, d.seen) == (want["accepted"], want["seen"])
            assert d.mode_id == want["mode"] and d.mode_name.decode() == want["mode_name"]
            assert [d.mode_name.decode(), d.mode_width, d.mode_height, d.mode_refresh] == table[d.mode_id]
            accepted += d.accepted
    assert accepted > 50
    md = gpu.ModeDetect()
    for (fo, fi, lo, li, fs, fps, height, mode) in fx["closest_random"]:
        md.reset()
        d = md.feed(fo, fi, lo, li, fs)
        assert (d.framerate, d.height, d.mode_id) == (fps, height, mode)
    assert len(set(c[7] for c in fx["closest_random"])) > 60  # most of the table is some query's answer


def test_reference_bytecode_live(orc):
    """Where the reference tree is present (the build container), run the jar's classes through the interpreter on
    cases that are NOT in the committed fixtures and hold the oracle and the library to them."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_java_fixtures_jvm as J
    if not J.jar_available():
        pytest.skip("reference jar not present")
    import hashlib
    from oracle.oracle import PlotScale
    assert hashlib.sha256(open(J.JAR, "rb").read()).hexdigest() == _jvm_fixtures()["jar_sha256"]
    build.build(verbose=False)
    vm = J.VM(J.JAR)
    r = np.random.default_rng(int.from_bytes(os.urandom(4), "little"))
    for _ in range(4):
        size, nwidth = int(r.integers(50, 4000)), int(r.integers(100, 1500))
        sx, actions = J.user_zoomed_scale(vm, size, nwidth, r)
        case = J.plot_case(vm, size, nwidth, None, None, int(r.integers(1, 2**31)), sx=sx)
        s = PlotScale()
        for k, v in case["scale"].items():
            setattr(s, k, v)
        vis, lo, hi, mi = orc.plot_populate(_fixture_plot(case), nwidth, s)
        assert (lo, hi, mi) == (case["lowest"], case["highest"], case["max_index"]), (size, nwidth, actions, case["seed"])
        assert hashlib.sha256(np.asarray(vis, np.float64).tobytes()).hexdigest() == case["visdata_sha"], (size, nwidth, actions, case["seed"])
    modes = J.video_modes(vm)
    det, md = J.MainDetector(vm), gpu.ModeDetect()
    fs = 50_000_000
    for _ in range(12):
        fo, fi, lo_, li = 400_000, 433_333 + int(r.integers(-1, 2)), 300, 441 + int(r.integers(-1, 2)) * int(r.random() < 0.2)
        want = det.on_plots(fo, fi, lo_, li, fs)
        d = md.feed(fo, fi, lo_, li, fs)
        assert (d.framerate, d.height, d.accepted, d.seen) == (want["fps"], want["height"], want["accepted"], want["seen"])
        assert d.mode_id == J.closest(vm, modes, want["fps"], want["height"])


def test_minijvm_arithmetic_follows_the_jvm_specification():
    """The bytecode interpreter behind the Java pins (tests/golden/minijvm.py) on hand-assembled method bodies: Java's
    integer division and remainder (truncating, sign of the dividend), 32/64-bit wrap-around, shifts that mask their
    count, saturating d2i / d2l with NaN -> 0, the NaN bias of dcmpl / dcmpg, iinc, a backward branch (JVMS 6.5)."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import minijvm

    vm = minijvm.VM.__new__(minijvm.VM)  # no jar: the snippets touch no class
    vm.names, vm.classes, vm.statics, vm.initialised, vm.hooks, vm.trace = set(), {}, {}, set(), {}, []

    def run(code, desc, *args):
        return vm.run(None, (8, bytes(code)), list(args), desc, static=True)

    ILOAD0, ILOAD1, IRET = 0x1a, 0x1b, 0xac
    assert run([ILOAD0, ILOAD1, 0x6c, IRET], "(II)I", -7, 2) == -3           # idiv truncates toward zero
    assert run([ILOAD0, ILOAD1, 0x70, IRET], "(II)I", -7, 2) == -1           # irem: sign of the dividend
    assert run([ILOAD0, ILOAD1, 0x70, IRET], "(II)I", 7, -2) == 1
    assert run([ILOAD0, ILOAD1, 0x60, IRET], "(II)I", 2**31 - 1, 1) == -2**31  # iadd wraps
    assert run([ILOAD0, ILOAD1, 0x68, IRET], "(II)I", 65536, 65536) == 0       # imul wraps
    assert run([ILOAD0, ILOAD1, 0x6c, IRET], "(II)I", -2**31, -1) == -2**31    # the one overflowing division
    assert run([ILOAD0, ILOAD1, 0x78, IRET], "(II)I", 1, 33) == 2              # ishl masks the count to 5 bits
    assert run([ILOAD0, ILOAD1, 0x7a, IRET], "(II)I", -16, 2) == -4            # ishr is arithmetic
    assert run([ILOAD0, ILOAD1, 0x7c, IRET], "(II)I", -16, 28) == 15           # iushr is logical
    with pytest.raises(ZeroDivisionError):
        run([ILOAD0, ILOAD1, 0x6c, IRET], "(II)I", 1, 0)
    DLOAD0, D2I, D2L, LRET = 0x26, 0x8e, 0x8f, 0xad
    for x, want in [(1e20, 2**31 - 1), (-1e20, -2**31), (float("nan"), 0), (-2.9, -2), (2.9, 2), (float("inf"), 2**31 - 1)]:
        assert run([DLOAD0, D2I, IRET], "(D)I", x) == want
    assert run([DLOAD0, D2L, LRET], "(D)J", 1e30) == 2**63 - 1 and run([DLOAD0, D2L, LRET], "(D)J", -1e30) == -2**63
    DLOAD2 = 0x28
    nan = float("nan")
    assert run([DLOAD0, DLOAD2, 0x97, IRET], "(DD)I", nan, 1.0) == -1 and run([DLOAD0, DLOAD2, 0x98, IRET], "(DD)I", nan, 1.0) == 1
    assert run([DLOAD0, DLOAD2, 0x97, IRET], "(DD)I", 2.0, 1.0) == 1 and run([DLOAD0, DLOAD2, 0x98, IRET], "(DD)I", 1.0, 1.0) == 0
    LLOAD0, LLOAD2 = 0x1e, 0x20
    assert run([LLOAD0, LLOAD2, 0x69, LRET], "(JJ)J", 2**62, 4) == 0                 # lmul wraps at 64 bits
    assert run([LLOAD0, LLOAD2, 0x94, IRET], "(JJ)I", -5, 3) == -1                   # lcmp
    # int s = 0; for (int i = 0; i < n; i++) s += i; return s;   (javac's shape: iinc + backward goto)
    loop = [0x03, 0x3c,              # iconst_0, istore_1        s
            0x03, 0x3d,              # iconst_0, istore_2        i
            0x1c, 0x1a, 0xa2, 0, 13,  # 4: iload_2, iload_0, if_icmpge +13 -> 19
            0x1b, 0x1c, 0x60, 0x3c,  # 9: iload_1, iload_2, iadd, istore_1
            0x84, 2, 1,              # 13: iinc 2, 1
            0xa7, 0xff, 0xf4,        # 16: goto -12 -> 4
            0x1b, IRET]              # 19: iload_1, ireturn
    assert run(loop, "(I)I", 10) == 45 and run(loop, "(I)I", 0) == 0
    # (int) Math.round(x) and (long) (x * y) as Main.roundData / hashHeightAndFPS compile them
    ok, v = vm.native("java/lang/Math", "round", "(D)J", [2.5])
    assert ok and v == 3
    ok, v = vm.native("java/lang/Math", "round", "(D)J", [-2.5])
    assert ok and v == -2  # floor(x + 0.5)
