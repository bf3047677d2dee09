#!/bin/bash
set -u
T=${1:-r5d}
O=gpurun_out/$T; mkdir -p $O
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1200 python -m pytest -q -m gpu tests/test_gpu_autocorr.py tests/test_gpu_certify.py tests/test_gpu_distributed.py tests/test_gpu_sweep_tool.py tests/test_gpu_dryrun.py --maxfail=8 --durations=5 -rs > $O/tests_new.log 2>&1; echo "new tests rc=$?" | tee $O/summary.txt; tail -12 $O/tests_new.log
ab() {
  tag=$1; shift
  env "$@" timeout 300 python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-legs > $O/ab_$tag.json 2> $O/ab_$tag.err
  python - $O/ab_$tag.json $tag <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "value", d["value"], "ms/pass", d.get("ms_per_pass"), "ac", (d.get("autocorrelation") or {}).get("group_ms_per_pass"), "frac", (d.get("autocorrelation") or {}).get("frac"),
      "stages", {n: v for n, v in (d.get("stage_ms_per_pass") or {}).items() if "ac" in n or "accum" in n}, "steady", (d.get("steady_state") or {}).get("ms_per_window"), (d.get("steady_state") or {}).get("frac"))
PY
}
ab fold_1 TSDRGPU_FOLD_ACC=1
ab nofold_1 TSDRGPU_FOLD_ACC=0
ab fold_2 TSDRGPU_FOLD_ACC=1
ab nofold_2 TSDRGPU_FOLD_ACC=0
for v in 1 0; do
  env TSDRGPU_FOLD_ACC=$v timeout 300 python bench.py --leg --gpus 1 --warmup 2 --config 4 --steps 4 --passes 25 > $O/cfg4_fold$v.json 2> $O/cfg4_fold$v.err
  python - $O/cfg4_fold$v.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("config4", sys.argv[1][-10:], "value", d["value"], "ac", (d.get("autocorrelation") or {}).get("group_ms_per_pass"), (d.get("autocorrelation") or {}).get("frac"), {n: v for n, v in (d.get("stage_ms_per_pass") or {}).items() if "ac" in n or "accum" in n})
PY
done
