#!/bin/bash
# round 6: the speculated band run — parity (band tests, the 8-process oracle tests, the dry runs), then configs[4] on one rank:
# the plain leg, the band path speculated (default) and literal (TSDRGPU_BAND_SPECULATE=0), twice each
set -u
T=${1:-r6bands}
O=gpurun_out/$T; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_bands.py tests/test_gpu_distributed.py tests/test_gpu_dryrun.py tests/test_gpu_edges.py -q -m gpu --maxfail=10 -p no:cacheprovider --tb=short > $O/tests.log 2>&1; echo "band tests rc=$? $(grep -E 'passed|failed' $O/tests.log | tail -1)" | tee $O/summary.txt
grep -E "^FAILED|^ERROR" $O/tests.log | cut -c1-220 | head
B="--config 4 --seconds 0.5 --steps 10 --warmup 3 --no-cpu-baseline --no-e2e"
for i in 1 2; do
  timeout 600 python bench.py $B > $O/plain_$i.json 2> $O/plain_$i.err
  timeout 600 python bench.py $B --bands --force-dist > $O/spec_$i.json 2> $O/spec_$i.err
  TSDRGPU_BAND_SPECULATE=0 timeout 600 python bench.py $B --bands --force-dist > $O/literal_$i.json 2> $O/literal_$i.err
done
python - <<PY | tee -a $O/summary.txt
import json
for i in (1,2):
  for t in ("plain","spec","literal"):
    try:
        d=json.loads(open("$O/%s_%d.json"%(t,i)).read().strip().splitlines()[-1])
        print(t, i, d["value"], d["ms_per_pass"], d["config"].get("row_bands"))
    except Exception as e:
        print(t,i,"failed",e, open("$O/%s_%d.err"%(t,i)).read()[-600:])
PY
