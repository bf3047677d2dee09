#!/usr/bin/env python3
"""Randomised end-to-end run of the tsdr_* drop-in (plugin -> libTSDRLibrary.so -> HIP) against the oracle's
deterministic driver: random sample rate / lines / refresh (raster slightly off it) / plugin block size / stage order / autoshift /
motion blur / frame-rate PLL.
Every delivered frame must equal an oracle frame, in order, starting with the first.

usage (on a GPU box):  python scripts/fuzz_engine.py [sessions] [seed]
"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as orc  # noqa: E402
from tempestsdr_amd import synth  # noqa: E402
import host_util as hu  # noqa: E402


def oracle_frames(iq, fs, h, fv, cfg):
    """chunk by chunk with the geometry of the moment; the PLL (when on) changes it between frames"""
    geo = orc.geometry(fs, h, fv)
    pp, rs = orc.PostProcess(geo), orc.Resampler()
    mb, lbs, aap, ash, pll = cfg
    mag = orc.am_demod(iq)
    pos, buf, frames = 0, np.zeros(0, np.float32), []
    while True:
        chunk = int(0.1 * fs / geo.refreshrate)
        if pos + chunk > mag.size:
            break
        buf = np.concatenate([buf, rs.process(mag[pos:pos + chunk], geo.width * geo.height * geo.refreshrate, float(fs))])
        pos += chunk
        while buf.size >= geo.width * geo.height:
            P, w = geo.width * geo.height, geo.width
            frames.append((w, pp.run(buf[:P].copy(), mb, 0.1, lbs, aap, ash, pll, 0)))
            buf = buf[P:]
    return frames


def one(rng, plugin, tmp):
    h = int(rng.integers(40, 700))
    fv = float(rng.choice([50.0, 59.94, 60.0, 75.0]))
    fs = int(rng.integers(300_000, 20_000_000))
    geo = orc.geometry(fs, h, fv)
    if geo.width < 8 or geo.width * h > 1_500_000:
        return None
    tw = max(8, int(round(fs / (fv * h))))
    mode = (tw, h, (tw * 4) // 5, (h * 9) // 10)
    nsamp = int(rng.uniform(6.5, 12.5) * fs / fv)
    fv_true = fv * float(rng.choice([1.0, 1.0, 1.0003, 1.004, 0.997]))  # raster slightly off the configured rate
    iq = synth.synth_iq(fs, mode, fv_true, nsamp, seed=int(rng.integers(1, 1 << 30)))
    path = os.path.join(tmp, "iq.f32")
    iq.tofile(path)
    block = 2 * int(rng.integers(1_000, max(1_001, min(600_000, nsamp // 4))))  # the test plugin sends whole blocks only
    cfg = (float(rng.choice([0.0, 0.0, 0.25, 0.9375])), int(rng.integers(0, 2)), int(rng.integers(0, 2)), int(rng.integers(0, 2)),
           int(rng.integers(0, 2)))
    mb, lbs, aap, ash, pll = cfg
    per = block // 2
    # half of the sessions whose recording fits the input queue (64 blocks) run the source flat out: the backlog behind the
    # engine's start-up then goes through batches of >= 8 frames, i.e. the fused run with the resampler's frame tracking
    free_running = nsamp // per <= 60 and rng.random() < 0.5
    params = f"{path} {fs} {block} {0 if free_running else 3000}"
    stream = iq
    if pll == 0 and rng.random() < 0.4 and nsamp // per >= 6:  # (with the PLL the skip unit moves with the geometry)
        # the plugin loses drop_n samples after block drop_at and reports them with the next block; the library
        # then skips whole multiples of one frame's samples (dsp.c:313-368) — same bookkeeping on the oracle side
        drop_at = int(rng.integers(1, nsamp // per - 3))
        drop_n = int(rng.integers(1, max(2, min(nsamp // 6, 4 * per))))
        params += f" {drop_at} {drop_n}"
        unit = int(round(((geo.width * geo.height) << 1) * geo.pixeltimeoversampletime))
        diff, pos, fwd = 0, 0, []
        for b in range((nsamp - drop_n) // per):
            dropped = drop_n if b == drop_at + 1 else 0
            seg = iq[2 * pos:2 * (pos + per)]
            pos += per
            if b == drop_at:
                pos += drop_n
            diff = orc.lib.orc_dropped_shift_with(diff, unit, dropped)
            if per <= diff:
                diff -= per
            else:
                fwd.append(seg[2 * diff:])
                diff = 0
        stream = np.concatenate(fwd)
    want = oracle_frames(stream, fs, h, fv, cfg)
    if len(want) < 5:
        return None
    s = hu.Session()
    tail = params.split(" ", 1)[1]
    desc = f"fs={fs} h={h} fv={fv} fv_true={fv_true:.4f} w={geo.width} block={block} cfg={cfg} params={tail} frames={len(want)}"
    try:
        assert s.lib.tsdr_loadplugin(s.h, plugin.encode(), params.encode()) == 0, s.err()
        s.lib.tsdr_setbasefreq(s.h, 400_000_000)
        s.lib.tsdr_setgain(s.h, 0.5)
        assert s.lib.tsdr_setresolution(s.h, h, fv) == 0
        s.lib.tsdr_motionblur(s.h, mb)
        s.lib.tsdr_setparameter_int(s.h, 6, lbs)
        s.lib.tsdr_setparameter_int(s.h, 7, aap)
        s.lib.tsdr_setparameter_int(s.h, 0, ash)
        s.lib.tsdr_setparameter_int(s.h, 1, pll)
        s.start()
        ok = s.wait_frames(len(want) - 3, 20)
        rc = s.stop()
        if not ok or rc != 0 or s.status != 0:
            return f"session failed ({ok}, {rc}, {s.status}, {s.err()}) {desc}"
        k, first = 0, None
        for (w_, h_, a) in s.frames:
            while k < len(want) and not (want[k][0] == w_ and h_ == h and np.array_equal(a, want[k][1], equal_nan=True)):
                k += 1
            if k >= len(want):
                return f"a delivered frame matches no oracle frame {desc}"
            first = k if first is None else first
            k += 1
        if first != 0:
            return f"first delivered frame is oracle frame {first} {desc}"
        if len(s.frames) < len(want) - 4:
            return f"only {len(s.frames)} of {len(want)} frames {desc}"
    finally:
        s.close()
    return ""


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    plugin = hu.build_test_plugin()
    ran = fails = 0
    with tempfile.TemporaryDirectory() as tmp:
        for c in range(n):
            r = one(rng, plugin, tmp)
            if r is None:
                continue
            ran += 1
            if r:
                fails += 1
                print("MISMATCH", c, r, flush=True)
    print(f"engine fuzz: {ran} sessions, {fails} failures, seed {seed}")
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
