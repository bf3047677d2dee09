#!/usr/bin/env python3
"""Golden fixtures for the two pieces of the reference that are JAVA, by hand transliteration (round 2; the pin proper
is make_java_fixtures_jvm.py, which executes the reference's compiled classes with a bytecode interpreter and whose
results this file's must equal — tests/test_extras_cpu.py).  Each function below is a LITERAL, line-by-line
transliteration of the cited Java source — same statements, same order, same integer/double conversions — written
independently of oracle/tsdr_oracle.c; the Java lines are quoted beside every statement.  The fixtures it writes
(tests/golden/java_fixtures.json) are what tests/test_extras_cpu.py checks the oracle's C restatements and the
library's tsdrgpu_modedetect_* / tsdrgpu_plot_columns against.

    python tests/golden/make_java_fixtures.py          # rewrites tests/golden/java_fixtures.json
"""
import json
import math
import os
import re

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("TSDR_REFERENCE", "/root/reference")


def java_int(x):
    """Java (int) of a double: truncation toward zero, saturating (JLS 5.1.3)."""
    if x != x:
        return 0
    if x >= 2147483647.0:
        return 2147483647
    if x <= -2147483648.0:
        return -2147483648
    return int(x)


def java_round(x):
    """Math.round(double): (long) Math.floor(x + 0.5)."""
    return int(math.floor(x + 0.5))


# ---- gui/scale/ZoomableXScale.java --------------------------------------------------------------
class ZoomableXScale:
    def __init__(self, min_value, max_value, max_pixels, max_zoom_val=10.0):
        self.min_value, self.max_value, self.max_pixels, self.max_zoom_val = float(min_value), float(max_value), int(max_pixels), float(max_zoom_val)
        self.reset_unsafe()

    def pixels_to_value_absolute(self, pixels):   # :133-137  return pixels * one_px_in_values_relative + offset_val + min_value;
        return pixels * self.one_px_in_values_relative + self.offset_val + self.min_value

    def pixels_to_value_relative(self, pixels):   # :139-143  return pixels * one_px_in_values_relative;
        return pixels * self.one_px_in_values_relative

    def value_to_pixel_absolute(self, val):       # :145-149  return (int) ((val - min_value) * one_val_in_pixels_relative) - offset_px;
        return java_int((val - self.min_value) * self.one_val_in_pixels_relative) - self.offset_px

    def value_to_pixel_relative(self, val):       # :151-155  return (int) (val * one_val_in_pixels_relative);
        return java_int(val * self.one_val_in_pixels_relative)

    def setPxOffset_unsafe(self, offset_px):      # :157-160
        self.offset_px = offset_px
        self.offset_val = self.pixels_to_value_relative(offset_px)

    def calculateValues_unsafe(self):             # :167-178
        self.one_val_in_pixels_relative = self.max_pixels / ((self.max_value - self.min_value) * self.scale)
        self.one_px_in_values_relative = ((self.max_value - self.min_value) * self.scale) / self.max_pixels
        values_in_screen = self.pixels_to_value_relative(self.max_pixels)
        if values_in_screen < self.max_zoom_val:
            self.scale = self.max_zoom_val / (self.max_value - self.min_value)
            self.one_val_in_pixels_relative = self.max_pixels / ((self.max_value - self.min_value) * self.scale)
            self.one_px_in_values_relative = ((self.max_value - self.min_value) * self.scale) / self.max_pixels

    def reset_unsafe(self):                       # :180-186
        self.scale = 1.0
        self.offset_val = 0.0
        self.offset_px = 0
        self.calculateValues_unsafe()

    def zoom_to(self, scale, offset_px):          # a zoom state as the mouse wheel / drag leave it (scale, then setPxOffset)
        self.scale = float(scale)
        self.calculateValues_unsafe()
        self.setPxOffset_unsafe(int(offset_px))

    def state(self):
        return {"one_val_in_pixels": self.one_val_in_pixels_relative, "one_px_in_values": self.one_px_in_values_relative,
                "offset_val": self.offset_val, "min_value": self.min_value, "offset_px": self.offset_px}


# ---- gui/PlotVisualizer.java:200-247, populateData() up to the y scaling ----------------------------
def populateData(data, size, nwidth, scale_x):
    visdata = [0.0] * nwidth
    highest_val = data[0]                                   # double highest_val = data[0];
    lowest_val = highest_val                                # double lowest_val = highest_val;
    max_index = 0                                           # max_index = 0;
    max_val = highest_val                                   # double max_val = highest_val;
    prev_px = 0                                             # int prev_px = 0;
    first_id = java_int(min(max(scale_x.pixels_to_value_absolute(0), 0), size))             # final int first_id = (int) Math.min( Math.max(scale_x.pixels_to_value_absolute(0), 0), size );
    last_id = java_int(min(max(scale_x.pixels_to_value_absolute(nwidth) + 1, 0), size))    # final int last_id = (int) Math.min( Math.max(scale_x.pixels_to_value_absolute(nwidth) + 1, 0), size);
    localmax = data[first_id]                               # double localmax = data[first_id];
    for id_ in range(first_id, last_id):                    # for (int id = first_id; id < last_id; id++) {
        val = data[id_]                                     #   final double val = data[id];
        px = scale_x.value_to_pixel_absolute(id_)           #   final int px = scale_x.value_to_pixel_absolute(id);
        if px >= 0 and px < nwidth:                         #   if (px >= 0 && px < nwidth) {
            if prev_px != px:                               #     if (prev_px != px) {
                if localmax > highest_val:                  #       if (localmax > highest_val) highest_val = localmax; else if (localmax < lowest_val) lowest_val = localmax;
                    highest_val = localmax
                elif localmax < lowest_val:
                    lowest_val = localmax
                for i in range(prev_px, px):                #       for (int i = prev_px; i < px; i++) visdata[i] = localmax;
                    visdata[i] = localmax
                localmax = val                              #       localmax = val;
                prev_px = px                                #       prev_px = px;
            elif val > localmax:                            #     } else if (val > localmax)
                localmax = val                              #       localmax = val;
        if val > max_val:                                   #   if (val > max_val) {
            max_val = val                                   #     max_val = val;
            max_index = id_                                 #     max_index = id;
    for i in range(prev_px, nwidth):                        # for (int i = prev_px; i < nwidth; i++) visdata[i] = localmax;
        visdata[i] = localmax
    return visdata, lowest_val, highest_val, max_index     # scale_y.setLowestHighestValue(lowest_val, highest_val);


# ---- gui/VideoMode.java:25-106 (the table, read from the reference's source) and :163-190 -----------
def video_modes():
    txt = open(os.path.join(REF, "JavaGUI/src/martin/tempest/gui/VideoMode.java")).read()
    return [(m.group(1), int(m.group(2)), int(m.group(3)), float(m.group(4)))
            for m in re.finditer(r'new VideoMode\("([^"]+)",\s*(\d+),\s*(\d+),\s*([0-9.]+)\)', txt)]


def findClosestVideoModeId(framerate, height, modes):       # VideoMode.java:163-190
    mode = -1
    diff = 5000.0
    for i, (_, w, h, r) in enumerate(modes):
        if h == height:
            delta = abs(r - framerate)
            if delta < diff:
                diff = delta
                mode = i
    if mode == -1:
        idiff = 5000
        for i, (_, w, h, r) in enumerate(modes):
            delta = abs(h - height)
            if delta < idiff:
                idiff = delta
                mode = i
    return mode


# ---- gui/Main.java:1227-1277 (+ :82, :1041-1043, :1301-1303, :1346-1350) ----------------------------
class AutoResolution:
    AUTO_FRAMERATE_CONVERGANCE_ITERATIONS = 3               # Main.java:82

    def __init__(self):
        self.auto_resolution_map = {}

    def on_plots(self, frame_offset, frame_max_index, line_offset, line_max_index, samplerate):
        auto_resolution_fps_id, auto_resolution_fps_offset = frame_max_index, frame_offset           # :1240-1241
        fps = samplerate / float(auto_resolution_fps_offset + auto_resolution_fps_id)               # :1301-1303 fromIndex
        linelength = float(line_offset + line_max_index)                                            # :1346-1350 fromIndexAndLength
        height = java_int(java_round((auto_resolution_fps_id + auto_resolution_fps_offset) / linelength))  # :1253, roundData :1041-1043
        key = int(fps * height)                                                                    # :1227-1229 (Long) (long) (fps * height)
        value = self.auto_resolution_map.get(key)                                                   # :1257
        accepted = value is not None and value == self.AUTO_FRAMERATE_CONVERGANCE_ITERATIONS        # :1259
        if not accepted:
            if value is None:
                value = 0                                                                           # :1265
            value += 1                                                                              # :1266
            self.auto_resolution_map[key] = value                                                   # :1267
        return {"fps": fps, "height": height, "linerate": samplerate / linelength, "accepted": int(accepted), "seen": self.auto_resolution_map[key]}


def main():
    rng = np.random.default_rng(20260924)
    out = {"populate": [], "modedetect": [], "closest_mode": []}
    for (size, nwidth, zoom, offpx) in [(5000, 800, 1.0, 0), (668756, 1237, 1.0, 0), (2315, 640, 1.0, 0), (300, 800, 1.0, 0),
                                        (5000, 800, 0.13, 411), (5000, 800, 0.01, 3700), (977, 977, 1.0, 0), (5000, 800, 1.0, -50),
                                        (53500, 1024, 0.5, 100), (185, 600, 1.0, 0)]:
        seed = int(rng.integers(1, 2**31))
        r = np.random.default_rng(seed)
        data = r.random(size) + 0.2 * np.sin(np.arange(size) / 37.0)
        data[r.integers(0, size, 3)] = 1.5  # exact ties: the first one must win the argmax
        sx = ZoomableXScale(0, size, nwidth)
        if zoom != 1.0 or offpx:
            sx.zoom_to(zoom, offpx)
        vis, lo, hi, mi = populateData(list(data), size, nwidth, sx)
        out["populate"].append({"size": size, "nwidth": nwidth, "seed": seed, "scale": sx.state(), "lowest": lo, "highest": hi,
                                "max_index": mi, "visdata_sha": __import__("hashlib").sha256(np.asarray(vis, np.float64).tobytes()).hexdigest(),
                                "visdata_head": vis[:8], "visdata_tail": vis[-4:]})
    modes = video_modes()
    out["n_modes"] = len(modes)
    for (fs, seq) in [(100_000_000, [(1149425, 517242, 766, 715)] * 2 + [(1149425, 517240, 766, 715)] + [(1149425, 517242, 766, 715)] * 3),
                      (8_000_000, [(91954, 41379, 61, 193), (91954, 41380, 61, 193), (91954, 41379, 61, 193), (91954, 41379, 61, 193),
                                   (91954, 41379, 61, 193), (91954, 36857, 61, 193)]),
                      (25_000_000, [(287356, 129311, 191, 326)] * 5)]:
        ar = AutoResolution()
        steps = []
        for (fo, fi, lo_, li) in seq:
            d = ar.on_plots(fo, fi, lo_, li, fs)
            d["mode"] = findClosestVideoModeId(d["fps"], d["height"], modes)
            d["mode_name"] = modes[d["mode"]][0] if d["mode"] >= 0 else None
            steps.append({"in": [fo, fi, lo_, li], "out": d})
        out["modedetect"].append({"samplerate": fs, "steps": steps})
    for (fr, h) in [(60.0, 1125), (59.9, 525), (75.0, 806), (60.0, 1001), (25.0, 625), (100.0, 509), (43.0, 817), (60.0, 4000), (60.0, 1)]:
        i = findClosestVideoModeId(fr, h, modes)
        out["closest_mode"].append({"framerate": fr, "height": h, "mode": i, "name": modes[i][0] if i >= 0 else None})
    with open(os.path.join(HERE, "java_fixtures.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", len(out["populate"]), "plot cases,", sum(len(m["steps"]) for m in out["modedetect"]), "detection steps,", len(modes), "modes")


if __name__ == "__main__":
    main()
