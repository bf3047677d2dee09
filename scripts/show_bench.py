#!/usr/bin/env python3
"""Prints the headline fields of a bench.py JSON line read from stdin (helper for A/B runs)."""
import json
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else ""
for line in sys.stdin:
    if line.startswith("{"):
        d = json.loads(line)
        print(tag, d["value"], d["ms_per_step"], d.get("stage_ms_per_step", {}))
