#!/usr/bin/env python3
"""Whole-library, PCIe-inclusive measurement through the tsdr_* API.

Sources: (mem) this repository's in-memory replay plugin (tempestsdr_amd/libTSDRPlugin_Mem.so, zero-copy hand-over),
(rawfile) the reference's RawFile plugin rebuilt free-running (PERFORMANCE_BENCHMARK=1, oracle/_ref; it fread()s
every block).  Libraries: this repository's libTSDRLibrary.so (MI355X) and, with --reference, the reference's own
library compiled from its sources (oracle/_ref/libtsdr_ref.so exports the same tsdr_* API) on the host cores of
the same box.  Effective MS/s = frames delivered per wall second x samples per frame.

usage: e2e_bench.py [--fs 100000000 --height 1125 --seconds 4 --reference]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tempestsdr_amd import synth, tsdrlib  # noqa: E402

MODES = {525: "640x480", 806: "1024x768", 1125: "1920x1080", 2250: "3840x2160"}


def main():
    import faulthandler
    faulthandler.enable()
    ap = argparse.ArgumentParser()
    ap.add_argument("--fs", type=int, default=100_000_000)
    ap.add_argument("--height", type=int, default=1125)
    ap.add_argument("--fv", type=float, default=60.0)
    ap.add_argument("--seconds", type=float, default=4.0)
    ap.add_argument("--file-seconds", type=float, default=0.3)
    ap.add_argument("--reference", action="store_true", help="also run the reference's CPU library on the same file")
    ap.add_argument("--only", default="", help="comma-separated leg names (default: all)")
    ap.add_argument("--variants", action="store_true", help="A/B legs of the engine's switches (round 4)")
    args = ap.parse_args()
    rawfile = os.path.join(ROOT, "oracle", "_ref", "libTSDRPlugin_RawFile_bench.so")
    reflib = os.path.join(ROOT, "oracle", "_ref", "libtsdr_ref.so")
    path = "/tmp/e2e_iq.f32"
    block = 524288
    n = (int(args.file_seconds * args.fs) // (block // 2)) * (block // 2)
    with open(path, "wb") as f:
        step = 1 << 22
        for s in range(0, n, step):
            synth.synth_iq(args.fs, MODES[args.height], args.fv, min(step, n - s), start=s).tofile(f)
    S = args.fs / args.fv
    out = {"config": f"{args.fs/1e6:g} MS/s float32 IQ recording of {n/args.fs:.3f} s replayed free-running, h={args.height}, fv={args.fv}",
           "host_cores": os.cpu_count()}

    def leg(name, lib, plugin, params, env=None, free=True, set_int=(), rgb=False):
        # every leg in a process of its own, like a host application (and so that no leg inherits another's runtime state)
        e = {"TSDR_GPU_STATS": "1"}
        e.update(env or {})
        if args.only and name not in args.only.split(","):
            return
        r = tsdrlib.throughput_subprocess(lib, plugin, params, args.height, args.fv, args.seconds, env=e, set_int=set_int, free=free, rgb=rgb)
        print(r.pop("stderr_tail", ""), file=sys.stderr)
        r["effective_Msps"] = r["frames_per_s"] * S / 1e6
        out[name] = r
        print(name, json.dumps(r), file=sys.stderr, flush=True)

    leg("mi355x_mem_plugin", tsdrlib.LIB, tsdrlib.MEM_PLUGIN, f"{path} {args.fs} {block} 0 0")
    leg("mi355x_mem_plugin_fast_modes", tsdrlib.LIB, tsdrlib.MEM_PLUGIN, f"{path} {args.fs} {block} 0 0", {"TSDR_GPU_EXACT": "0"})
    leg("mi355x_mem_plugin_bounce_buffers", tsdrlib.LIB, tsdrlib.MEM_PLUGIN, f"{path} {args.fs} {block} 0 0", {"TSDR_GPU_ZEROCOPY": "0"})
    # PARAM_INT_FRAMERATE_PLL (1): one frame per launch group and one host round trip per frame (the nudge feeds back)
    leg("mi355x_mem_plugin_pll_on", tsdrlib.LIB, tsdrlib.MEM_PLUGIN, f"{path} {args.fs} {block} 0 0",
        set_int=[(1, 1)])
    # int16 recording of the same stream: half the bytes in (tsdrplugin_readasync_raw, decoded on the device)
    path16 = "/tmp/e2e_iq.s16"
    import numpy as np
    with open(path, "rb") as fin, open(path16, "wb") as fout:
        while True:
            a = np.fromfile(fin, np.float32, 1 << 22)
            if a.size == 0:
                break
            np.clip(np.round(a * 20000.0), -32768, 32767).astype(np.int16).tofile(fout)
    leg("mi355x_mem_plugin_int16_raw", tsdrlib.LIB, tsdrlib.MEM_PLUGIN, f"{path16} {args.fs} {block} 0 0 int16")
    path8 = "/tmp/e2e_iq.s8"
    with open(path, "rb") as fin, open(path8, "wb") as fout:
        while True:
            a = np.fromfile(fin, np.float32, 1 << 22)
            if a.size == 0:
                break
            np.clip(np.round(a * 100.0), -128, 127).astype(np.int8).tofile(fout)
    # the narrowest way in and the viewer's format out: int8 IQ (2 bytes per sample in), packed RGB frames (8 bytes per sample out)
    leg("mi355x_mem_plugin_int8_raw_rgb_out", tsdrlib.LIB, tsdrlib.MEM_PLUGIN, f"{path8} {args.fs} {block} 0 0 int8", rgb=True)
    leg("mi355x_mem_plugin_rgb_out", tsdrlib.LIB, tsdrlib.MEM_PLUGIN, f"{path} {args.fs} {block} 0 0", rgb=True)
    if args.variants:
        base = (tsdrlib.LIB, tsdrlib.MEM_PLUGIN, f"{path} {args.fs} {block} 0 0")
        leg("variant_two_uploads_in_flight", *base, {"TSDR_GPU_ASYNC_UPLOAD": "1"})
        leg("variant_ring_1GiB", *base, {"TSDR_GPU_AUTOCORR_RETAIN_MB": "1024"})
        leg("variant_no_premise_check", *base, {"TSDRGPU_AC_CHECK_EVERY": "0"})
        leg("variant_plots_off", *base, set_int=[(3, 1)])
        leg("variant_block_8MiB", tsdrlib.LIB, tsdrlib.MEM_PLUGIN, f"{path} {args.fs} {4 * block} 0 0")
        leg("variant_exact_detector", *base, {"TSDR_GPU_AUTOCORR": "exact"})
        if os.path.exists(rawfile):
            leg("variant_rawfile_no_copy_thread", tsdrlib.LIB, rawfile, f"{path} {args.fs} float", {"TSDR_GPU_COPY_THREAD": "0"})
    if os.path.exists(rawfile):
        leg("mi355x_rawfile_plugin", tsdrlib.LIB, rawfile, f"{path} {args.fs} float")
        if args.reference and os.path.exists(reflib):
            try:  # the reference's own teardown is not safe (use-after-free, TSDRLibrary.c:523-527): a crash there is its, not ours
                leg("reference_cpu_rawfile_plugin", reflib, rawfile, f"{path} {args.fs} float", free=False)
            except RuntimeError as ex:
                out["reference_cpu_rawfile_plugin"] = {"error": str(ex)[:300]}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
