#!/usr/bin/env python3
"""Turns rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter CSVs into per-kernel HBM bytes per launch.

Units and corrections (MI355X_MICROARCH.md §HBM): the counters are in KiB; on gfx950 FETCH_SIZE reports half
of the bytes of a wide coalesced streaming read, so it is doubled; WRITE_SIZE is uncalibrated by the guide, so
both are calibrated here against k_demod_vec4, whose traffic is known exactly (8 B read + 4 B written per
sample, float4 accesses): the calibration factors are printed and applied.

usage: pmc_summarize.py <dir with FETCH_SIZE/ and WRITE_SIZE/ sub-directories> [out.json [flat_for_bench.json]]
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def load(dirpath, counter):
    files = glob.glob(os.path.join(dirpath, counter, "**", "*counter_collection.csv"), recursive=True)
    per = defaultdict(list)
    for f in files:
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            name = row["Kernel_Name"]
            per[name].append(float(row["Counter_Value"]))
    return per


def short(name):
    m = re.match(r"(?:void )?(k_[A-Za-z0-9_]+)", name)
    if not m:
        return None
    k = m.group(1)
    if k == "k_rs_area_up":  # the sample-parallel form of the same stage (bench.py's stage name is k_rs_area)
        k = "k_rs_area"
    if k == "k_frame_stats" and re.search(r"k_frame_stats<\s*true\s*>", name):  # the fused run's statistics + store trip
        k = "k_frame_stats_store"
    if k == "k_ac_cols":  # one kernel template, two trips: <log2 N1, input mode, LAST>
        k += "_trip3" if re.search(r"k_ac_cols<[^>]*true>", name) else "_trip1"
    if k == "k_sb_cols":  # the stitch's first trips: <log2 N1, 5> abs-diff of the hops (alignment), <log2 N1, 6> the rotated hops
        k += "_absdiff" if re.search(r"k_sb_cols<\d+, 5>", name) else "_rotated"
    if k == "k_sb_rows":
        k += "_xcorr" if re.search(r"k_sb_rows<0>", name) else "_stitch"
    return k


def main():
    d = sys.argv[1]
    fetch = load(d, "FETCH_SIZE")
    write = load(d, "WRITE_SIZE")
    names = sorted(set(fetch) | set(write))
    rows = {}
    for n in names:
        k = short(n)
        if not k:
            continue
        f = fetch.get(n, [])
        w = write.get(n, [])
        r = rows.setdefault(k, {"launches": 0, "fetch_kib": 0.0, "write_kib": 0.0})
        r["launches"] += max(len(f), len(w))
        r["fetch_kib"] += sum(f)
        r["write_kib"] += sum(w)
    # calibration on the demod kernel of bench.py --pmc-calibrate: nsamples = 99 999 600
    nsamp = 99_999_600  # bench.py --pmc-calibrate demodulates exactly this many, whatever the batch length
    cal_f = cal_w = None
    if "k_demod_vec4" in rows and rows["k_demod_vec4"]["launches"]:
        r = rows["k_demod_vec4"]
        n_l = r["launches"]
        cal_f = (8.0 * nsamp) / (r["fetch_kib"] / n_l * 1024.0) if r["fetch_kib"] else None
        cal_w = (4.0 * nsamp) / (r["write_kib"] / n_l * 1024.0) if r["write_kib"] else None
    print(f"calibration on k_demod_vec4 (8 B read + 4 B written per sample): FETCH_SIZE x{cal_f}, WRITE_SIZE x{cal_w}")
    ff = cal_f if cal_f else 2.0  # the guide's gfx950 correction
    wf = cal_w if cal_w else 1.0
    out = {}
    print(f"{'kernel':24s} {'launches':>8s} {'fetch MB/launch':>16s} {'write MB/launch':>16s} {'total MB/launch':>16s}")
    for k, r in sorted(rows.items()):
        n_l = max(r["launches"], 1)
        fb = r["fetch_kib"] / n_l * 1024.0 * ff
        wb = r["write_kib"] / n_l * 1024.0 * wf
        out[k] = {"launches": r["launches"], "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb,
                  "hbm_bytes_per_launch": fb + wb}
        print(f"{k:24s} {r['launches']:8d} {fb / 1e6:16.1f} {wb / 1e6:16.1f} {(fb + wb) / 1e6:16.1f}")
    if "k_ac_cols_trip1" in out and "k_ac_cols_trip3" in out:  # the profiler stage of bench.py covers both trips
        a, b = out["k_ac_cols_trip1"], out["k_ac_cols_trip3"]
        n_l = a["launches"] + b["launches"]
        out["k_ac_cols"] = {"launches": n_l,
                            "fetch_bytes_per_launch": (a["fetch_bytes_per_launch"] * a["launches"] + b["fetch_bytes_per_launch"] * b["launches"]) / n_l,
                            "write_bytes_per_launch": (a["write_bytes_per_launch"] * a["launches"] + b["write_bytes_per_launch"] * b["launches"]) / n_l}
        out["k_ac_cols"]["hbm_bytes_per_launch"] = out["k_ac_cols"]["fetch_bytes_per_launch"] + out["k_ac_cols"]["write_bytes_per_launch"]
    res = {"fetch_factor": ff, "write_factor": wf, "unit": "bytes per launch", "kernels": out}
    if len(sys.argv) > 2:
        json.dump(res, open(sys.argv[2], "w"), indent=1)
    if len(sys.argv) > 3:
        # the flat form bench.py reads for roofline.traffic: kernel -> HBM bytes per launch
        flat = {k: int(round(v["hbm_bytes_per_launch"])) for k, v in out.items()}
        flat["_source"] = os.environ.get("PMC_SOURCE", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (scripts/pmc_collect.sh), not collected live")
        flat["_note"] = (f"HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, KiB units), "
                         f"FETCH_SIZE x{ff:.3f} and WRITE_SIZE x{wf:.3f} calibrated on k_demod_vec4 (known 8 B read + 4 B "
                         "written per sample); bench workload: 1 s of 100 MS/s IQ per pass (60 frames; 17 windows transformed "
                         "9+8 per launch); k_ac_cols = launch-weighted mean of its two trips; scripts/pmc_collect.sh")
        json.dump(flat, open(sys.argv[3], "w"), indent=1)
    return res


if __name__ == "__main__":
    main()
