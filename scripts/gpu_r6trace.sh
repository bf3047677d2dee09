#!/bin/bash
# per-launch durations of the autocorrelation's trips by window count, for two orders of the same cut (rocprofv3 kernel trace, one lane)
set -u
T=${1:-r6trace}
O=gpurun_out/$T; mkdir -p $O
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
for sp in 9,8 8,9 8,8,1 1,8,8; do
  tag=$(echo $sp | tr ',' '_')
  (cd /tmp && TSDRGPU_AC_SPLIT=$sp timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/$tag -o t -- python $R/bench.py --steps 2 --warmup 1 --passes 10 --no-cpu-baseline --no-e2e --no-legs --serial --no-profile > $R/$O/$tag.log 2>&1)
  python - <<PY
import csv,glob,collections
fs=glob.glob("$O/$tag/**/*kernel_trace.csv",recursive=True)
d=collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    k=r['Kernel_Name']
    if k.startswith('void k_ac_cols') or k.startswith('k_ac_rows'):
        d[(k.split('(')[0][-22:], r['Grid_Size_Y'])].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000)
print("split $sp")
for k,v in sorted(d.items()):
    v.sort(); print("   %-24s windows %2s  launches %3d  min %6.1f  med %6.1f  max %6.1f us"%(k[0],k[1],len(v),v[0],v[len(v)//2],v[-1]))
PY
  find $O/$tag -name "*.csv" -size +1M -delete
done 2>&1 | tee $O/summary.txt
