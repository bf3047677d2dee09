#!/bin/bash
# NOTE: the TSDR_NT macros (non-temporal loads / stores behind -DTSDR_NT=<bits>) were measured from the working tree of that day and removed
# again without being committed (profiles/round6_ab_runs.txt has the numbers; the three places are marked in tsdrgpu_frame.hip); kept as the record.
# round 6: (1) the band path with the chain speculated (default) against the literal run, configs[4] on one rank; (2) streaming hints on the
# frame path's big streams: A/B builds tempestsdr_amd/ab/nt<bits>.so (scripts/build_ab2.sh, -DTSDR_NT=<bits>) against the product
set -u
T=${1:-r6nt}
O=gpurun_out/$T; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_bands.py tests/test_gpu_distributed.py -q -m gpu --maxfail=10 -p no:cacheprovider --tb=short > $O/tests.log 2>&1; echo "band tests rc=$? $(grep -E 'passed|failed' $O/tests.log | tail -1)" | tee $O/summary.txt
grep -E "^FAILED|^ERROR" $O/tests.log | cut -c1-220 | head
B="--config 4 --seconds 0.5 --steps 10 --warmup 3 --no-cpu-baseline --no-e2e"
for i in 1 2; do
  timeout 600 python bench.py $B --bands --force-dist > $O/spec_$i.json 2> $O/spec_$i.err
  TSDRGPU_BAND_SPECULATE=0 timeout 600 python bench.py $B --bands --force-dist > $O/literal_$i.json 2> $O/literal_$i.err
done
H="--steps 20 --warmup 5 --no-cpu-baseline --no-e2e"
for i in 1 2; do
  timeout 600 python bench.py $H > $O/nt0_$i.json 2> $O/nt0_$i.err
  for v in 1 2 4 6 7; do
    TSDRGPU_LIB=tempestsdr_amd/ab/nt$v.so timeout 600 python bench.py $H > $O/nt${v}_$i.json 2> $O/nt${v}_$i.err
  done
done
python - <<PY | tee -a $O/summary.txt
import json,glob,os
def rd(p):
    l=[x for x in open(p) if x.startswith("{")]
    return json.loads(l[-1])
for i in (1,2):
  for t in ("spec","literal"):
    try:
        d=rd("$O/%s_%d.json"%(t,i)); print(t, i, d["value"], d["ms_per_pass"], d["config"].get("row_bands"))
    except Exception as e: print(t,i,"failed",e)
for i in (1,2):
  for v in (0,1,2,4,6,7):
    try:
        d=rd("$O/nt%d_%d.json"%(v,i)); print("nt",v,"run",i, d["value"], d["ms_per_pass"], d["frame_path"], d["roofline"]["frac"])
    except Exception as e: print("nt",v,i,"failed",e)
PY
