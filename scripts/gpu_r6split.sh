#!/bin/bash
# round 6: how a pass's 17 capture windows are cut into launches (TSDRGPU_AC_SPLIT): whole residency rounds (two windows fill the
# 512 resident workgroups of a trip) against the product's 9 + 8
set -u
T=${1:-r6split}
O=gpurun_out/$T; mkdir -p $O
H="--steps 20 --warmup 5 --no-cpu-baseline --no-e2e"
for i in 1 2; do
  for sp in default 8,8,1 16,1 1,16 1,8,8 10,7 6,6,5 4,4,4,4,1; do
    tag=$(echo $sp | tr ',' '_')
    if [ $sp = default ]; then timeout 600 python bench.py $H > $O/${tag}_$i.json 2> $O/${tag}_$i.err
    else TSDRGPU_AC_SPLIT=$sp timeout 600 python bench.py $H > $O/${tag}_$i.json 2> $O/${tag}_$i.err; fi
  done
done
python - <<PY | tee $O/summary.txt
import json,glob
for i in (1,2):
  for sp in "default 8_8_1 16_1 1_16 1_8_8 10_7 6_6_5 4_4_4_4_1".split():
    try:
        d=json.loads([x for x in open("$O/%s_%d.json"%(sp,i)) if x.startswith("{")][-1])
        r=d["roofline"]; print("%-10s run %d  %9.2f GS/s  %.4f ms/pass  group %.4f ms  frac %.4f  detected %s"%(sp,i,d["value"],d["ms_per_pass"],r["avg_launch_ms"],r["frac"],d["detected"]["frame_lag"]))
    except Exception as e: print(sp,i,"failed",e)
PY
