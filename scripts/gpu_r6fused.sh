#!/bin/bash
# round 6: the FUSED band run — parity (in-process band tests, the process-per-rank oracle tests), then configs[4] on one rank:
# plain leg, fused band run (default), two-trip band run (--no-band-fuse), twice each; and the headline stream in bands (blur 0: lines)
set -u
T=${1:-r6fused}
O=gpurun_out/$T; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_bands.py tests/test_gpu_distributed.py tests/test_gpu_dryrun.py tests/test_gpu_demod_resample.py -q -m gpu --maxfail=12 -p no:cacheprovider --tb=short > $O/tests.log 2>&1; echo "band tests rc=$? $(grep -E 'passed|failed' $O/tests.log | tail -1)" | tee $O/summary.txt
grep -E "^FAILED|^ERROR" $O/tests.log | cut -c1-220 | head -20
B="--config 4 --seconds 0.5 --steps 10 --warmup 3 --no-cpu-baseline --no-e2e"
for i in 1 2; do
  timeout 600 python bench.py $B > $O/plain_$i.json 2> $O/plain_$i.err
  timeout 600 python bench.py $B --bands --force-dist > $O/fused_$i.json 2> $O/fused_$i.err
  timeout 600 python bench.py $B --bands --force-dist --no-band-fuse > $O/twotrip_$i.json 2> $O/twotrip_$i.err
  timeout 600 python bench.py $B --bands --force-dist --no-band-prefetch > $O/fusednopf_$i.json 2> $O/fusednopf_$i.err
done
H="--steps 10 --warmup 3 --no-cpu-baseline --no-e2e"
timeout 600 python bench.py $H --bands --force-dist > $O/h_fused.json 2> $O/h_fused.err
timeout 600 python bench.py $H --bands --force-dist --no-band-fuse > $O/h_twotrip.json 2> $O/h_twotrip.err
python - <<PY | tee -a $O/summary.txt
import json
def rd(p):
    l=[x for x in open(p) if x.startswith("{")]
    return json.loads(l[-1])
for i in (1,2):
  for t in ("plain","fused","twotrip","fusednopf"):
    try:
        d=rd("$O/%s_%d.json"%(t,i)); print(t, i, d["value"], d["ms_per_pass"], d.get("frame_path"), d["config"].get("row_bands"))
    except Exception as e: print(t,i,"failed",e, open("$O/%s_%d.err"%(t,i)).read()[-800:])
for t in ("h_fused","h_twotrip"):
    try:
        d=rd("$O/%s.json"%t); print(t, d["value"], d["ms_per_pass"], d.get("frame_path"), d["config"].get("row_bands"))
    except Exception as e: print(t,"failed",e, open("$O/%s.err"%t).read()[-800:])
PY
