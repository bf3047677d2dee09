#!/bin/bash
set -u
T=${1:-r5b}
O=gpurun_out/$T; mkdir -p $O
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest -q -m gpu tests/test_gpu_autocorr.py tests/test_gpu_certify.py --maxfail=5 --durations=5 > $O/tests_new.log 2>&1; echo "new tests rc=$?" | tee $O/summary.txt; tail -12 $O/tests_new.log
for v in new c4; do
  if [ $v = new ]; then unset TSDRGPU_LIB; else export TSDRGPU_LIB=$R/tempestsdr_amd/ab/$v.so; fi
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stitch_prof_$v -o t -- python $R/scripts/exp_stitch_prof.py > $R/$O/stitch_$v.log 2>&1)
  f=$(find $O/stitch_prof_$v -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/stitch_kernel_stats_$v.csv
  find $O/stitch_prof_$v -type f -size +1M -delete
  echo "== $v"; grep "^stitch" $O/stitch_$v.log | tail -3; cut -c1-150 $O/stitch_kernel_stats_$v.csv | head -10
done
unset TSDRGPU_LIB
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-legs --no-e2e --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/summary.txt
python - $O/bench.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "ms/pass", d.get("ms_per_pass"))
print("superbandwidth", d.get("superbandwidth"))
print("steady_state", d.get("steady_state"))
PY
