#!/bin/bash
set -u
O=gpurun_out/r2p; mkdir -p $O
for v in "" "--overlap" "--overlap --no-split" "--frames-per-launch 20" "--fuse"; do
  tag=$(echo "base$v" | tr -d ' -')
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e $v > $O/$tag.json 2> $O/$tag.err
  python - <<PY
import json
d=json.loads([l for l in open("$O/$tag.json") if l.startswith("{")][0])
print("$tag", d["value"], d["ms_per_pass"], {k:v for k,v in d["stage_ms_per_pass"].items() if v>0.02})
PY
done
