#!/bin/bash
# round 6: the fused frame path in sub-batches small enough for the Infinity Cache (bench.py --fuse-sub N) against whole batches
set -u
T=${1:-r6sub}
O=gpurun_out/$T; mkdir -p $O
H="--steps 20 --warmup 5 --no-cpu-baseline --no-e2e"
for i in 1 2; do
  for n in 0 8 10 12 16 20 30; do
    timeout 600 python bench.py $H --fuse-sub $n > $O/sub${n}_$i.json 2> $O/sub${n}_$i.err
  done
done
python - <<PY | tee $O/summary.txt
import json
for i in (1,2):
  for n in (0,8,10,12,16,20,30):
    try:
        d=json.loads([x for x in open("$O/sub%d_%d.json"%(n,i)) if x.startswith("{")][-1])
        print("fuse-sub %2d run %d  %9.2f GS/s  %.4f ms/pass  frame_path %s  autocorr %s"%(n,i,d["value"],d["ms_per_pass"],d.get("frame_path"),d.get("autocorrelation")))
    except Exception as e: print(n,i,"failed",e, open("$O/sub%d_%d.err"%(n,i)).read()[-500:])
PY
