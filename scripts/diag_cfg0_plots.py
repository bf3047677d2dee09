#!/usr/bin/env python3
"""The reference's THREADED library against the oracle, detector side (BASELINE configs[0], companion of diag_cfg0_replay.py):
every plot update the library announces (frame plot, line plot, VALUE_ID_AUTOCORRECT_FRAMES_COUNT = c) is compared with the
oracle's running mean over c capture windows.  The detector's ring takes whole plugin blocks and is purged when it refuses one
(frameratedetector.c:215-230), its thread takes windows of 3.1 fs / 55 samples from it (frameratedetector.c:128-187): the windows
are consecutive stretches of the blocks the ring accepted since its last purge.  Which blocks those were is searched per update
(the first window starts at a block boundary; later ones follow on), the values must then be BIT-identical."""
import ctypes as C
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tempestsdr_amd import tsdrlib, synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402

fs, h, fv = 8_000_000, 525, 60.0
path = "/tmp/diag_cfg0_plots.f32"
iq = synth.synth_iq(fs, "640x480", fv, 2 * fs, seed=0x5EED0000)
iq.tofile(path)
reflib = os.path.join(ROOT, "oracle", "_ref", "libtsdr_ref.so")
rawfile = os.path.join(ROOT, "oracle", "_ref", "libTSDRPlugin_RawFile.so")
lib = tsdrlib.load(reflib)
events, lock = [], threading.Lock()


def on_plot(pid, off, vals, size, rate, ctx):
    a = np.ctypeslib.as_array(vals, shape=(size,)).copy()
    with lock:
        events.append(("plot", pid, off, a))


def on_value(vid, a0, a1, ctx):
    with lock:
        events.append(("value", vid, a0, a1))


cbs = (tsdrlib.FRAME_CB(lambda *a: None), tsdrlib.VALUE_CB(on_value), tsdrlib.PLOT_CB(on_plot))
hnd = C.c_void_p()
lib.tsdr_init(C.byref(hnd), cbs[1], cbs[2], None)
pbuf = C.create_string_buffer(f"{path} {fs} float".encode())
assert lib.tsdr_loadplugin(hnd, rawfile.encode(), pbuf) == 0
lib.tsdr_setgain(hnd, 0.5)
assert lib.tsdr_setresolution(hnd, h, fv) == 0
th = threading.Thread(target=lambda: lib.tsdr_readasync(hnd, cbs[0], None))
th.start()
time.sleep(1.3)
lib.tsdr_stop(hnd)
th.join(30)
updates, cur = [], {}
for e in events:
    if e[0] == "plot":
        cur[e[1]] = e[3]
    elif e[1] == 2 and 0 in cur and 1 in cur:  # VALUE_ID_AUTOCORRECT_FRAMES_COUNT closes an update
        updates.append((int(e[3]), cur[0], cur[1]))
        cur = {}
print(len(updates), "plot updates; window counts announced:", [u[0] for u in updates])
cap = orc.capture_size(fs)
mag = orc.am_demod(iq)  # the detector is fed demodulated blocks (TSDRLibrary.c:283-290)
BLK = 262144
ac = orc.Autocorr(fs)
pos, ok, lost_blocks = 0, 0, []
for k, (calls, fp, lp) in enumerate(updates):
    if calls != ac.calls + 1:
        print("update", k, ": the count jumped from", ac.calls, "to", calls, "(a reset) - stopping here")
        break
    placed = None
    # the window follows on from the last one, or — after a purge — starts at a later block boundary
    cands = [pos] + [b * BLK for b in range(pos // BLK + 1, pos // BLK + 12)]
    for start in cands:
        if start + cap > mag.size:
            break
        trial = orc.Autocorr(fs)
        trial.calls, trial.frame, trial.line = ac.calls, ac.frame.copy(), ac.line.copy()
        trial.run(mag[start:start + cap].copy())
        if np.array_equal(trial.frame, fp) and np.array_equal(trial.line, lp):
            placed = start
            ac = trial
            break
    if placed is None:
        print("update", k, ": no window placement reproduces the plots")
        break
    if placed != pos:
        lost_blocks.append((k, pos, placed))
    pos = placed + cap
    ok += 1
print(f"PLOTS: {ok} of {len(updates)} plot updates of the threaded library are BIT-identical to the oracle's running mean over the windows it took"
      + (f"; the ring was purged before update(s) {[(k, f'resumed at block {b // BLK}') for k, a, b in lost_blocks]}" if lost_blocks else "; consecutive windows from the first sample"))
