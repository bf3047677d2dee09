#!/bin/bash
set -u
O=gpurun_out/r2q; mkdir -p $O
run() { tag=$1; shift; timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e "$@" > $O/$tag.json 2> $O/$tag.err; python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/$tag.json") if l.startswith("{")][0])
    print("$tag", d["value"], d["ms_per_pass"], {k:v for k,v in d["stage_ms_per_pass"].items() if v>0.02})
except Exception as e:
    print("$tag failed", e); print(open("$O/$tag.err").read()[-1500:])
PY
}
run fence1; TSDRGPU_EVENT_NOFENCE=1 run nofence1; run fence2; TSDRGPU_EVENT_NOFENCE=1 run nofence2
TSDRGPU_EVENT_NOFENCE=1 timeout 1200 python -m pytest tests/test_gpu_autocorr.py tests/test_gpu_postproc.py tests/test_gpu_demod_resample.py -x -q -m gpu 2>&1 | grep -E "passed|failed|rror" | tail -3
