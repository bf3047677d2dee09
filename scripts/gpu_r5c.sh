#!/bin/bash
set -u
T=${1:-r5c}
O=gpurun_out/$T; mkdir -p $O
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
ab() {  # $1 = tag, rest = env assignments
  tag=$1; shift
  env "$@" timeout 300 python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-legs > $O/ab_$tag.json 2> $O/ab_$tag.err
  python - $O/ab_$tag.json $tag <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "value", d["value"], "ms/pass", d.get("ms_per_pass"), "ac", (d.get("autocorrelation") or {}).get("group_ms_per_pass"), "frac", (d.get("autocorrelation") or {}).get("frac"),
      "stages", {n: v for n, v in (d.get("stage_ms_per_pass") or {}).items() if "ac" in n or "accum" in n}, "steady", (d.get("steady_state") or {}).get("ms_per_window"), (d.get("steady_state") or {}).get("frac"))
PY
}
ab rows512_1 TSDRGPU_ROWS256=0
ab rows256_1 TSDRGPU_ROWS256=1
ab rows512_2 TSDRGPU_ROWS256=0
ab rows256_2 TSDRGPU_ROWS256=1
ab retaincopy TSDRGPU_RETAIN_COPY=1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee $O/summary.txt
python - $O/bench.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "ms/pass", d.get("ms_per_pass"), "roofline", {k: d["roofline"].get(k) for k in ("frac", "frac_rocprof", "frac_moved", "achieved")})
print("whole", d["whole_pass"]); print("frame", {k: d["frame_path"].get(k) for k in ("frac", "frac_moved", "kernels_ms_per_pass")})
print("kernels", d["kernels"])
print("superbandwidth", {k: d["superbandwidth"].get(k) for k in ("ms_per_stitch", "frac", "frac_moved")} if d.get("superbandwidth") else None)
print("steady_state", {k: d["steady_state"].get(k) for k in ("ms_per_window", "frac", "plot_updates_uncertified", "epochs_replayed_exact")} if d.get("steady_state") else None)
for k, v in (d.get("configs") or {}).items():
    if not v: print(k, v); continue
    print(k, {kk: v.get(kk) for kk in ("value_Msps", "whole_pass_frac", "error")}, "cpu", (v.get("cpu_baseline") or {}).get("value"), ((v.get("cpu_baseline") or {}).get("pipeline") or {}).get("value"))
print("configs[0]", json.dumps((d.get("configs") or {}).get("configs[0]"))[:1500])
print("cpu_baseline", {k: d["cpu_baseline"].get(k) for k in ("value", "cores")} if d.get("cpu_baseline") else None)
PY
timeout 1800 python -m pytest tests -q -m gpu --maxfail=10 --durations=10 > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt; grep -E "passed|failed" $O/tests.log | tail -2; grep -E "^FAILED|^ERROR" $O/tests.log | head
