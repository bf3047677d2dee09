#!/bin/bash
# NOTE: the persistent kernel and its TSDRGPU_AC_ROWS switch were measured from the working tree of that day and dropped without ever being
# committed (what it was and what it measured: the note above ac4_rows_pair in fft4step.h, profiles/round6_ab_runs.txt); kept as the record of the command.
# round 6: trip 2 of the autocorrelation as a persistent grid (TSDRGPU_AC_ROWS=1: rows addressed as buffers, 2: plain pointers) against the
# one-pair-per-workgroup kernel (0, the default): parity of the three-trip plan first, then same-box A/B/C timing of the driver's command
set -u
T=${1:-r6rows}
O=gpurun_out/$T; mkdir -p $O
for m in 0 1 2; do
  TSDRGPU_AC_ROWS=$m timeout 600 python -m pytest tests/test_gpu_autocorr.py -q -m gpu -k "three_trip" -p no:cacheprovider --tb=line > $O/parity_$m.log 2>&1; echo "mode $m parity rc=$? $(tail -1 $O/parity_$m.log)" | tee -a $O/summary.txt
done
for i in 1 2; do
  for m in 0 1 2; do
    TSDRGPU_AC_ROWS=$m timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e > $O/m${m}_$i.json 2> $O/m${m}_$i.err
  done
done
python - <<PY | tee -a $O/summary.txt
import json
for i in (1,2):
  for m in (0,1,2):
    try:
        d=json.loads(open("$O/m%d_%d.json"%(m,i)).read().strip().splitlines()[-1])
        r=d.get("roofline",{})
        print("mode",m,"run",i, d["value"], d["ms_per_pass"], r.get("kernel"), r.get("frac"), r.get("avg_launch_ms"))
    except Exception as e:
        print(m,i,"failed",e)
PY
timeout 1200 python -m pytest tests/test_gpu_autocorr.py tests/test_gpu_distributed.py tests/test_gpu_dryrun.py -q -m gpu -x -p no:cacheprovider --tb=short > $O/tests.log 2>&1; echo "default-mode tests rc=$? $(tail -1 $O/tests.log)" | tee -a $O/summary.txt
