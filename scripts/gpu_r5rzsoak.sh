#!/bin/bash
# differential soaks with BOTH red-zone layers on: the library's own allocations (TSDRGPU_REDZONES=2, reports collected in one
# file) and the scripts' buffers (TSDR_TEST_REDZONES=1: tests/conftest.py's zones); random shapes are where an overrun would hide
set -u
T=${1:-r5rzsoak}
O=gpurun_out/$T; mkdir -p $O
export TSDRGPU_REDZONES=2 TSDRGPU_REDZONE_LOG=$PWD/$O/redzone_reports.txt TSDR_TEST_REDZONES=1
: > $TSDRGPU_REDZONE_LOG
for seed in ${SEEDS:-81 82}; do timeout 900 python scripts/fuzz_parity.py 2500 $seed > $O/fuzz_parity_$seed.txt 2>&1; tail -1 $O/fuzz_parity_$seed.txt; done
timeout 600 python scripts/fuzz_engine.py ${ENGINE_SESSIONS:-120} ${ENGINE_SEED:-91} > $O/fuzz_engine.txt 2>&1; tail -1 $O/fuzz_engine.txt
echo "red-zone reports of the library: $(wc -l < $TSDRGPU_REDZONE_LOG)" | tee $O/summary.txt
sort $TSDRGPU_REDZONE_LOG | uniq -c | sort -rn | head
