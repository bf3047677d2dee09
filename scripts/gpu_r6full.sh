#!/bin/bash
# round 6: the whole GPU suite, smoke, the driver's bench command, then same-box A/B pairs of one environment switch
# usage: gpu_r6full.sh TAG [ENVVAR=val-for-B]   (A = default build, B = the switch set)
set -u
T=${1:-r6full}
AB=${2:-}
O=gpurun_out/$T; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu --maxfail=15 --durations=6 -rs -p no:cacheprovider > $O/tests.log 2>&1; echo "tests rc=$?" | tee $O/summary.txt; grep -E "passed|failed" $O/tests.log | tail -2; grep -E "^FAILED|^ERROR|SKIPPED" $O/tests.log | cut -c1-220 | head -20
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt
S=$(date +%s.%N)
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err; RC=$?
E_=$(date +%s.%N)
echo "bench rc=$RC wall_s=$(python -c "print(round($E_-$S,1))") last_line_bytes=$(tail -1 $O/bench_line.json | wc -c)" | tee -a $O/summary.txt
cp gpurun_out/bench_detail.json $O/bench_detail.json 2>/dev/null
tail -1 $O/bench_line.json | cut -c1-1800
if [ -n "$AB" ]; then
  for i in 1 2; do
    timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e > $O/A$i.json 2> $O/A$i.err
    env $AB timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e > $O/B$i.json 2> $O/B$i.err
  done
  python - <<PY
import json
for t in ("A1","B1","A2","B2"):
    try:
        d=json.loads(open("$O/%s.json"%t).read().strip().splitlines()[-1])
        r=d.get("roofline",{})
        print(t, d["value"], d["ms_per_step"], r.get("frac"), r.get("frac_rocprof"), r.get("avg_launch_ms"))
    except Exception as e:
        print(t,"failed",e)
PY
fi
