#!/usr/bin/env python3
"""The Java rows of SURVEY 8(f) pinned to the reference's OWN compiled classes.

tests/golden/make_java_fixtures.py transliterates the Java source by hand; this script instead EXECUTES the released
jar of the reference (Release/JavaGUI/JTempestSDR.jar) with the bytecode interpreter of tests/golden/minijvm.py — the
image has no JVM — on the same cases with the same seeds, and writes what the reference's classes computed to
tests/golden/java_fixtures_jvm.json:

  * plot decimation: gui/scale/ZoomableXScale (constructor, setMaxPixels, reset, setMinMaxValue, calculateValues_unsafe,
    setPxOffset_unsafe) and gui/PlotVisualizer.populateData() run as bytecode; the LogScale of the y axis is a stub
    that snapshots the columns at setLowestHighestValue(), which is where the library's tsdrgpu_plot_columns stops;
  * mode detection: gui/Main.onIncommingPlot(PLOT_ID, ...) runs as bytecode for FRAME and LINE plots — the enum switch,
    the anonymous fps transformer, TransformerAndCallbackHeight.fromIndexAndLength, roundData, hashHeightAndFPS, the
    HashMap<Long, Integer> counting and the == AUTO_FRAMERATE_CONVERGANCE_ITERATIONS acceptance; the two plotters'
    plot() is replaced by a stub that stores (offset, max index, sample rate), onResolutionChange() by a recorder,
    Swing widgets are opaque;
  * gui/VideoMode: the static table built by the class initialiser and findClosestVideoModeId() run as bytecode.

tests/test_extras_cpu.py requires java_fixtures_jvm.json to agree with java_fixtures.json value for value (so the
transliteration, the oracle's C restatement and the library are all held to what the reference's bytecode computes),
and re-runs a small case through the interpreter when the jar is present.

    python tests/golden/make_java_fixtures_jvm.py       # rewrites tests/golden/java_fixtures_jvm.json (~ minutes)
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from minijvm import VM, JArr, JObj  # noqa: E402

REF = os.environ.get("TSDR_REFERENCE", "/root/reference")
JAR = os.path.join(REF, "Release/JavaGUI/JTempestSDR.jar")

PV = "martin/tempest/gui/PlotVisualizer"
ZX = "martin/tempest/gui/scale/ZoomableXScale"
LS = "martin/tempest/gui/scale/LogScale"
MAIN = "martin/tempest/gui/Main"
VMODE = "martin/tempest/gui/VideoMode"
PLOT_ID = "martin/tempest/core/TSDRLibrary$IncomingValueCallback$PLOT_ID"


def jar_available():
    return os.path.exists(JAR)


def x_scale(vm, size, nwidth, zoom=1.0, offpx=0):
    """The x scale as PlotVisualizer holds it when plot() reaches populateData(): new ZoomableXScale(10) (:64),
    setMaxPixels(nwidth) (setBounds :298), reset() (:261), setMinMaxValue(0, size) (:265); then a zoom state."""
    sx = vm.new(ZX)
    vm.invoke(ZX, "<init>", "(D)V", [sx, 10.0])
    vm.invoke(ZX, "setMaxPixels", "(I)V", [sx, nwidth])
    vm.invoke(ZX, "reset", "()V", [sx])
    vm.invoke(ZX, "setMinMaxValue", "(DD)V", [sx, 0.0, float(size)])
    if zoom != 1.0 or offpx:
        sx.fields["scale"] = float(zoom)
        vm.invoke(ZX, "calculateValues_unsafe", "()V", [sx])
        vm.invoke(ZX, "setPxOffset_unsafe", "(I)V", [sx, int(offpx)])
    return sx


def x_scale_state(sx):
    f = sx.fields
    return {"one_val_in_pixels": f["one_val_in_pixels_relative"], "one_px_in_values": f["one_px_in_values_relative"],
            "offset_val": f["offset_val"], "min_value": f["min_value"], "offset_px": f["offset_px"]}


def populate(vm, data, size, nwidth, sx):
    """PlotVisualizer.populateData() (:200-247) on an instance whose fields are set as plot() (:249-272) sets them."""
    pv = vm.new(PV)
    pv.fields.update(data=JArr(7, [float(x) for x in data]), visdata=JArr(7, [0.0] * nwidth), size=size, nwidth=nwidth,
                     scale_x=sx, scale_y=vm.new(LS))
    snap = {}
    vm.hooks[(LS, "valuesValid", "()Z")] = lambda vm_, a: 1

    def set_lo_hi(vm_, a):
        snap.update(lowest=a[1], highest=a[2], visdata=list(pv.fields["visdata"].data))
    vm.hooks[(LS, "setLowestHighestValue", "(DD)V")] = set_lo_hi
    vm.hooks[(LS, "valtopx", "(D)I")] = lambda vm_, a: 0
    vm.invoke(PV, "populateData", "()V", [pv])
    return snap["visdata"], snap["lowest"], snap["highest"], pv.fields["max_index"]


class MainDetector:
    """A gui/Main instance with just the fields onIncommingPlot() (:1232-1277) reads, fed plots like the library's
    callback thread feeds them (FRAME then LINE)."""

    def __init__(self, vm):
        self.vm = vm
        m = self.m = vm.new(MAIN)
        self.changes = []
        self.last = {}
        hm = JObj("java/util/HashMap")
        hm.native = {}
        fps_cls = self.fps_transformer_class()
        fps_t, h_t = vm.new(fps_cls), vm.new(MAIN + "$TransformerAndCallbackHeight")
        fps_t.fields["this$0"] = h_t.fields["this$0"] = m
        m.fields.update(auto_resolution=1, auto_resolution_map=hm, fps_transofmer=fps_t, height_transformer=h_t,
                        frame_plotter=vm.new(PV), line_plotter=vm.new(PV), btnReset=JObj("javax/swing/JToggleButton"),
                        btnAutoResolution=JObj("javax/swing/JToggleButton"))

        def plot(vm_, a):  # PlotVisualizer.plot(double[], int offset, int size, long samplerate): the stub keeps what the getters return
            a[0].fields.update(offset=a[2], samplerate=a[4], max_index=int(a[1].data[0]))
        vm.hooks[(PV, "plot", "([DIIJ)V")] = plot
        vm.hooks[(MAIN, "onResolutionChange", "(DILjava/lang/String;)V")] = lambda vm_, a: self.changes.append((a[1], a[2]))
        key = (MAIN, "hashHeightAndFPS", "(DI)Ljava/lang/Long;")

        def hash_hook(vm_, a):
            self.last = {"fps": a[1], "height": a[2]}
            del vm_.hooks[key]
            try:
                r = vm_.invoke(MAIN, "hashHeightAndFPS", "(DI)Ljava/lang/Long;", a)
            finally:
                vm_.hooks[key] = hash_hook
            self.last["key"] = r.native
            return r
        vm.hooks[key] = hash_hook

    def fps_transformer_class(self):
        """the anonymous TransformerAndCallback whose fromIndex divides the sample rate (Main.java:1288-1303): the
        one that is not TransformerAndCallbackHeight and implements fromIndex"""
        for n in sorted(self.vm.names):
            if n.startswith(MAIN + "$") and n[len(MAIN) + 1:].isdigit():
                cf = self.vm.load(n)
                if cf.super == PV + "$TransformerAndCallback" and ("fromIndex", "(IIJ)D") in cf.methods:
                    return n
        raise RuntimeError("fps transformer class not found in the jar")

    def enum(self, name):
        self.vm.init(PLOT_ID)
        return self.vm.statics[PLOT_ID][name]

    def on_plots(self, frame_offset, frame_max_index, line_offset, line_max_index, samplerate):
        vm, m = self.vm, self.m
        desc = "(L" + PLOT_ID + ";I[DIJ)V"
        self.last = {}
        n_changes = len(self.changes)
        # the stubbed plot() takes the max index from data[0]: the real one finds it in populateData (pinned separately)
        vm.invoke(MAIN, "onIncommingPlot", desc, [m, self.enum("FRAME"), frame_offset, JArr(7, [float(frame_max_index)]), 1, samplerate])
        vm.invoke(MAIN, "onIncommingPlot", desc, [m, self.enum("LINE"), line_offset, JArr(7, [float(line_max_index)]), 1, samplerate])
        accepted = len(self.changes) > n_changes
        if accepted:
            assert self.changes[-1] == (self.last["fps"], self.last["height"])
            assert m.fields["auto_resolution"] == 0
            m.fields["auto_resolution"] = 1  # the transliteration keeps detecting: so do we
        seen = m.fields["auto_resolution_map"].native.get(("java/lang/Long", self.last["key"]))
        return {"fps": self.last["fps"], "height": self.last["height"], "accepted": int(accepted), "seen": seen.native}


def video_modes(vm):
    vm.init(VMODE)
    modes = vm.invoke(VMODE, "getVideoModes", "()[L" + VMODE + ";", []) if ("getVideoModes", "()[L" + VMODE + ";") in vm.load(VMODE).methods else None
    if modes is None:
        for v in vm.statics[VMODE].values():
            if isinstance(v, JArr) and v.data and isinstance(v.data[0], JObj) and v.data[0].cls == VMODE:
                modes = v
    return modes


def closest(vm, modes, fr, h):
    return vm.invoke(VMODE, "findClosestVideoModeId", "(DI[L" + VMODE + ";)I", [float(fr), int(h), modes])


POPULATE_CASES = [(5000, 800, 1.0, 0), (668756, 1237, 1.0, 0), (2315, 640, 1.0, 0), (300, 800, 1.0, 0), (5000, 800, 0.13, 411),
                  (5000, 800, 0.01, 3700), (977, 977, 1.0, 0), (5000, 800, 1.0, -50), (53500, 1024, 0.5, 100), (185, 600, 1.0, 0)]
DETECT_CASES = [(100_000_000, [(1149425, 517242, 766, 715)] * 2 + [(1149425, 517240, 766, 715)] + [(1149425, 517242, 766, 715)] * 3),
                (8_000_000, [(91954, 41379, 61, 193), (91954, 41380, 61, 193), (91954, 41379, 61, 193), (91954, 41379, 61, 193),
                             (91954, 41379, 61, 193), (91954, 36857, 61, 193)]),
                (25_000_000, [(287356, 129311, 191, 326)] * 5)]
CLOSEST_CASES = [(60.0, 1125), (59.9, 525), (75.0, 806), (60.0, 1001), (25.0, 625), (100.0, 509), (43.0, 817), (60.0, 4000), (60.0, 1)]


def plot_data(seed, size):
    r = np.random.default_rng(seed)
    data = r.random(size) + 0.2 * np.sin(np.arange(size) / 37.0)
    data[r.integers(0, size, 3)] = 1.5  # exact ties: the first one must win the argmax
    return data


def plot_case(vm, size, nwidth, zoom, offpx, seed, sx=None):
    data = plot_data(seed, size)
    if sx is None:
        sx = x_scale(vm, size, nwidth, zoom, offpx)
    vis, lo, hi, mi = populate(vm, data, size, nwidth, sx)
    return {"size": size, "nwidth": nwidth, "seed": seed, "scale": x_scale_state(sx), "lowest": lo, "highest": hi, "max_index": mi,
            "visdata_sha": hashlib.sha256(np.asarray(vis, np.float64).tobytes()).hexdigest(), "visdata_head": vis[:8], "visdata_tail": vis[-4:]}


def user_zoomed_scale(vm, size, nwidth, r):
    """a zoom state reached the way a user reaches it: mouse-wheel steps (PlotVisualizer.java:96-104: zoomAround(x,
    ZOOM_AMOUNT or UNZOOM_AMOUNT)) and drags (:71-78: moveOffsetWithPixels(dx)), all public methods with the scale's
    own clamping (autoFixOffset_unsafe) in force"""
    vm.init(PV)
    zin, zout = vm.statics[PV]["ZOOM_AMOUNT"], vm.statics[PV]["UNZOOM_AMOUNT"]
    sx = x_scale(vm, size, nwidth)
    actions = []
    for _ in range(int(r.integers(1, 9))):
        if r.random() < 0.65:
            px, c = int(r.integers(0, nwidth)), (zin if r.random() < 0.75 else zout)
            vm.invoke(ZX, "zoomAround", "(ID)V", [sx, px, c])
            actions.append(["zoom", px, c])
        else:
            dx = int(r.integers(-nwidth, nwidth))
            vm.invoke(ZX, "moveOffsetWithPixels", "(I)V", [sx, dx])
            actions.append(["drag", dx])
    return sx, actions


def more_cases(vm, out, modes, names):
    r = np.random.default_rng(20260925)
    out["plotscale_default"] = []
    for (size, nwidth) in [(668756, 1237), (5, 800), (2315, 640), (1, 16), (53500, 1024), (9, 1920), (10, 300), (11, 300), (4194304, 3840)]:
        out["plotscale_default"].append({"size": size, "nwidth": nwidth, "scale": x_scale_state(x_scale(vm, size, nwidth))})
    out["populate_zoomed"] = []
    for i in range(32):
        size = int(r.choice([185, 977, 2315, 5000, 20000, 53500]))
        nwidth = int(r.choice([300, 640, 800, 1024, 1237, 1920]))
        sx, actions = user_zoomed_scale(vm, size, nwidth, r)
        c = plot_case(vm, size, nwidth, None, None, int(r.integers(1, 2**31)), sx=sx)
        c["actions"] = actions
        out["populate_zoomed"].append(c)
    out["modedetect_random"] = []
    for i in range(8):
        fs = int(r.choice([8_000_000, 10_000_000, 20_000_000, 25_000_000, 50_000_000, 100_000_000]))
        m = modes.data[int(r.integers(0, len(modes.data)))].fields
        frame_lag = fs / m["refreshrate"]
        line_lag = frame_lag / m["height"]
        fo, lo_ = int(frame_lag * 0.7), int(line_lag * 0.5)
        det = MainDetector(vm)
        steps = []
        for _ in range(24):
            fi = int(round(frame_lag)) - fo + int(r.integers(-1, 2)) * int(r.random() < 0.3)
            li = max(1, int(round(line_lag)) - lo_ + int(r.integers(-1, 2)) * int(r.random() < 0.2))
            d = det.on_plots(fo, fi, lo_, li, fs)
            d["mode"] = closest(vm, modes, d["fps"], d["height"])
            d["mode_name"] = names[d["mode"]] if d["mode"] >= 0 else None
            steps.append({"in": [fo, fi, lo_, li], "out": d})
        out["modedetect_random"].append({"samplerate": fs, "steps": steps})
    # single detections over the whole mode table and its gaps: what Main computes from the two plots (fps, height) and the
    # mode VideoMode.findClosestVideoModeId picks for them (the library is reset before each one)
    out["closest_random"] = []
    heights = sorted(set(m.fields["height"] for m in modes.data))
    fs = 100_000_000
    for i in range(400):
        h = int(r.choice(heights)) + int(r.integers(-3, 4)) * int(r.random() < 0.5) if r.random() < 0.8 else int(r.integers(1, 5000))
        h = max(h, 1)
        fr = float(r.choice([23.976, 24, 25, 30, 43, 50, 56, 59.94, 60, 65, 70, 72, 75, 85, 100, 120])) + float(r.normal(0, 1.5))
        frame_lag = int(round(fs / fr))
        line_lag = max(1, int(round(frame_lag / h)))
        fo, lo_ = frame_lag // 2, line_lag // 3
        d = MainDetector(vm).on_plots(fo, frame_lag - fo, lo_, line_lag - lo_, fs)
        out["closest_random"].append([fo, frame_lag - fo, lo_, line_lag - lo_, fs, d["fps"], d["height"], closest(vm, modes, d["fps"], d["height"])])
    out["mode_table"] = [[m.fields["name"], m.fields["width"], m.fields["height"], m.fields["refreshrate"]] for m in modes.data]


def main():
    vm = VM(JAR)
    rng = np.random.default_rng(20260924)
    out = {"source": "Release/JavaGUI/JTempestSDR.jar executed by tests/golden/minijvm.py",
           "jar_sha256": hashlib.sha256(open(JAR, "rb").read()).hexdigest(), "populate": [], "modedetect": [], "closest_mode": []}
    for (size, nwidth, zoom, offpx) in POPULATE_CASES:
        seed = int(rng.integers(1, 2**31))
        out["populate"].append(plot_case(vm, size, nwidth, zoom, offpx, seed))
        print("plot case", size, nwidth, zoom, offpx, "max_index", out["populate"][-1]["max_index"], flush=True)
    modes = video_modes(vm)
    names = [m.fields["name"] for m in modes.data]
    out["n_modes"] = len(modes.data)
    for (fs, seq) in DETECT_CASES:
        det = MainDetector(vm)
        steps = []
        for (fo, fi, lo_, li) in seq:
            d = det.on_plots(fo, fi, lo_, li, fs)
            d["mode"] = closest(vm, modes, d["fps"], d["height"])
            d["mode_name"] = names[d["mode"]] if d["mode"] >= 0 else None
            steps.append({"in": [fo, fi, lo_, li], "out": d})
        out["modedetect"].append({"samplerate": fs, "steps": steps})
    for (fr, h) in CLOSEST_CASES:
        i = closest(vm, modes, fr, h)
        out["closest_mode"].append({"framerate": fr, "height": h, "mode": i, "name": names[i] if i >= 0 else None})
    more_cases(vm, out, modes, names)
    with open(os.path.join(HERE, "java_fixtures_jvm.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", len(out["populate"]), "plot cases,", sum(len(m["steps"]) for m in out["modedetect"]), "detection steps,", len(names), "modes")


if __name__ == "__main__":
    main()
