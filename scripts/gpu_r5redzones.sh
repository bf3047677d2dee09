#!/bin/bash
# the GPU suite and a differential soak with red zones around every device allocation the LIBRARY makes (TSDRGPU_REDZONES=2:
# report and go on; reports of all processes collected in one file) on top of the ones tests/conftest.py puts around the tests' buffers
set -u
T=${1:-r5rz}
O=gpurun_out/$T; mkdir -p $O
export TSDRGPU_REDZONES=2 TSDRGPU_REDZONE_LOG=$PWD/$O/redzone_reports.txt
: > $TSDRGPU_REDZONE_LOG
timeout 2400 python -m pytest tests -q -m gpu --maxfail=20 --durations=4 -rs > $O/tests.log 2>&1; echo "tests rc=$?" | tee $O/summary.txt; grep -E "passed|failed" $O/tests.log | tail -2; grep -E "^FAILED|^ERROR" $O/tests.log | head -20
timeout 600 python scripts/fuzz_parity.py 600 77 > $O/fuzz.txt 2>&1; echo "fuzz rc=$?" | tee -a $O/summary.txt; tail -3 $O/fuzz.txt
echo "red-zone reports: $(wc -l < $TSDRGPU_REDZONE_LOG)" | tee -a $O/summary.txt
sort $TSDRGPU_REDZONE_LOG | uniq -c | sort -rn | head -20
