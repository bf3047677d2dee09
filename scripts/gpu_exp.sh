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
run a; run b
timeout 1200 python -m pytest tests/test_gpu_demod_resample.py tests/test_gpu_host_pipeline.py -x -q -m gpu 2>&1 | grep -E "passed|failed|rror" | tail -3
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/$O/gapprof
cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/gapprof -o trace -- python $R/bench.py --steps 2 --warmup 1 --passes 10 --no-cpu-baseline --no-e2e --no-profile > $R/$O/gapprof.log 2>&1
cd $R
python scripts/bench_gaps.py $(find $O/gapprof -name "*kernel_trace.csv" | head -1) | head -9
find $O/gapprof -type f -size +1M -delete
