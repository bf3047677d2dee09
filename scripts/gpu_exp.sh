#!/bin/bash
set -u
O=gpurun_out/r2q; mkdir -p $O
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e > $O/$tag.json 2> $O/$tag.err; python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/$tag.json") if l.startswith("{")][0])
    print("$tag", d["value"], d["ms_per_pass"], {k:v for k,v in d["stage_ms_per_pass"].items() if v>0.02})
except Exception as e:
    print("$tag failed", e); print(open("$O/$tag.err").read()[-1500:])
PY
}
run pad0 A=1
run pad16 TSDRGPU_AC_PAD=16
run pad32 TSDRGPU_AC_PAD=32
run pad64 TSDRGPU_AC_PAD=64
run pad8 TSDRGPU_AC_PAD=8
run pad0b A=1
TSDRGPU_AC_PAD=32 timeout 900 python -m pytest tests/test_gpu_autocorr.py -x -q -m gpu > $O/pytest.log 2>&1; grep -E "passed|failed|error" $O/pytest.log | tail -3
