#!/bin/bash
# the host library after its move to acquire/release atomics: the host-pipeline tests, the stress driver against the real
# libtsdrgpu.so plain and under the sanitizers (ThreadSanitizer with the HIP runtime in the process is an experiment: its
# outcome, whatever it is, is kept), and the whole-library throughput legs for a before/after
set -u
T=${1:-r5host}
O=gpurun_out/$T; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_host_pipeline.py tests/test_gpu_extras.py -q -m gpu --maxfail=10 --durations=5 -rs > $O/tests.log 2>&1; echo "tests rc=$?" | tee $O/summary.txt; grep -E "passed|failed" $O/tests.log | tail -2; grep -E "^FAILED|^ERROR" $O/tests.log | head
python - <<'PY'
import numpy as np
np.random.default_rng(1).standard_normal(8_000_000).astype(np.float32).tofile('/tmp/iq.f32')
PY
export GPU_MAX_HW_QUEUES=2 TSDR_GPU_STATS=1
for v in plain asan tsan; do
  timeout 200 tests/sanitize/host_stress_$v tempestsdr_amd/libTSDRPlugin_Mem.so "/tmp/iq.f32 8000000 524288 0 2000" 525 60 4 2 > $O/stress_$v.log 2>&1
  echo "stress $v rc=$?" | tee -a $O/summary.txt
  grep -c "WARNING: ThreadSanitizer" $O/stress_$v.log | sed "s/^/  tsan warnings: /"
  grep "SUMMARY" $O/stress_$v.log | sort | uniq -c | sort -rn | head -12
  grep "^host_stress\|^tsdr stats: [0-9]" $O/stress_$v.log | tail -3
done
unset TSDR_GPU_STATS
timeout 600 python scripts/e2e_bench.py --seconds 3 > $O/e2e.txt 2> $O/e2e.err; echo "e2e rc=$?" | tee -a $O/summary.txt
grep -E "MS/s|GS/s" $O/e2e.txt | head -12
