#!/bin/bash
# round 5, first GPU call: the new tests first, then the whole suite, the stitch's kernel trace, the bench line, and the same-box A/B
# of the refactored column kernels against round 4's text (tempestsdr_amd/ab/r4cols.so)
set -u
T=${1:-r5a}
O=gpurun_out/$T; mkdir -p $O
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest -q -x -m gpu tests/test_gpu_extras.py tests/test_gpu_autocorr.py tests/test_gpu_certify.py "tests/test_gpu_host_pipeline.py::test_dropped_samples_keep_alignment" "tests/test_gpu_host_pipeline.py::test_superresolution_mode" --durations=8 > $O/tests_new.log 2>&1; echo "new tests rc=$?" | tee $O/summary.txt; tail -15 $O/tests_new.log
scripts/micro/arith_check > $O/arith_check.txt 2>&1; echo "arith rc=$?" | tee -a $O/summary.txt; tail -4 $O/arith_check.txt
scripts/micro/wave_reduce_check > $O/wave_reduce_check.txt 2>&1; echo "wave rc=$?" | tee -a $O/summary.txt; cat $O/wave_reduce_check.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stitch_prof -o t -- python $R/scripts/exp_stitch_prof.py > $R/$O/stitch.log 2>&1)
f=$(find $O/stitch_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/stitch_kernel_stats.csv
find $O/stitch_prof -type f -size +1M -delete
cat $O/stitch.log | tail -8; cut -c1-160 $O/stitch_kernel_stats.csv | head -14
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/summary.txt
python scripts/show_bench.py $O/bench.json | cut -c1-2500
for i in 1 2; do
  for v in new r4cols; do
    if [ $v = new ]; then unset TSDRGPU_LIB; else export TSDRGPU_LIB=$R/tempestsdr_amd/ab/$v.so; fi
    timeout 300 python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-legs > $O/ab_${v}_$i.json 2> $O/ab_${v}_$i.err
    python - $O/ab_${v}_$i.json $v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d.get("kernels", {})
print(sys.argv[2], "value", d["value"], "ms/pass", d.get("ms_per_pass"), "ac", (d.get("autocorrelation") or {}).get("group_ms_per_pass"), {n: v.get("avg_launch_ms") for n, v in k.items()}, "stages", {n: v for n, v in (d.get("stage_ms_per_pass") or {}).items() if "ac" in n or "accum" in n})
PY
  done
done
unset TSDRGPU_LIB
timeout 1500 python -m pytest tests -q -m gpu --maxfail=10 --durations=10 > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt; grep -E "passed|failed" $O/tests.log | tail -2
