#!/bin/bash
# kernel trace of the bench; usage: gpu_prof.sh <tag> [bench args]
set -u
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o trace -- python $R/bench.py --steps 2 --warmup 1 --passes 10 --no-cpu-baseline --no-e2e --no-legs --serial --no-profile "$@" > $R/$O/prof.log 2>&1
cd $R
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats.csv
find $O/prof -type f -size +1M -delete
grep -v "at::native" $O/kernel_stats.csv | cut -c1-170 | head -32
