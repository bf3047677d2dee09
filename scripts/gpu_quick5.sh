#!/bin/bash
# a short confirmation call: GPU suite, smoke, the driver's bench command
set -u
T=${1:-r5q}
O=gpurun_out/$T; mkdir -p $O
timeout 1800 python -m pytest tests -q -m gpu --maxfail=10 --durations=6 -rs > $O/tests.log 2>&1; echo "tests rc=$?" | tee $O/summary.txt; grep -E "passed|failed" $O/tests.log | tail -2; grep -E "^FAILED|^ERROR|SKIPPED" $O/tests.log | head
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/summary.txt
python scripts/show_bench.py $O/bench.json | cut -c1-600
