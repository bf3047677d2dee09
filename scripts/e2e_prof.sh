#!/bin/bash
set -u
O=gpurun_out/r2q; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/$O/e2eprof
cd /tmp && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/$O/e2eprof -o trace -- python $R/scripts/e2e_exp.py '[["first","f32",{}],["second","f32",{}]]' > $R/$O/e2eprof.log 2>&1
cd $R
grep -E "MS/s" $O/e2eprof.log
python scripts/e2e_trace_analyze.py $(find $O/e2eprof -name "*kernel_trace.csv" | head -1) $(find $O/e2eprof -name "*memory_copy_trace.csv" | head -1) > $O/e2e_trace_analysis.txt 2>&1
find $O/e2eprof -type f -size +1M -delete
cat $O/e2e_trace_analysis.txt | head -60
