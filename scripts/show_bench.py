#!/usr/bin/env python3
"""Prints the headline fields of bench.py JSON lines (files given as arguments, or stdin)."""
import json
import sys


def show(tag, line):
    d = json.loads(line)
    print(tag, "value", d["value"], d["unit"], "| ms/pass", d.get("ms_per_pass"), "| step_ms", d.get("step_ms"))
    print("   roofline", d.get("roofline"))
    for k in ("kernels", "frame_path", "autocorrelation", "whole_pass", "stage_ms_per_pass", "exact_autocorr", "e2e", "detected", "cpu_baseline"):
        if d.get(k) is not None:
            print("  ", k, d[k])


files = sys.argv[1:]
if not files:
    for line in sys.stdin:
        if line.startswith("{"):
            show("", line)
for f in files:
    for line in open(f):
        if line.startswith("{"):
            show(f, line)
