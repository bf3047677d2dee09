#!/bin/bash
# quick A/B: bench (exact + fast) and optionally a test subset; usage: gpu_quick.sh <tag> [pytest args...]
set -u
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
if [ $# -gt 0 ]; then timeout 900 python -m pytest "$@" -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt; tail -n 4 $O/tests.log; fi
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --fast-sync > $O/bench_fast.json 2> $O/bench_fast.err
python scripts/show_bench.py $O/bench.json $O/bench_fast.json 2>&1 | grep -v "detected\|exact_autocorr\|whole_pass\|frame_path\|roofline\|kernels" | cut -c1-900
grep -o '"sync_redo_last_batch": {[^}]*}' $O/bench.json
