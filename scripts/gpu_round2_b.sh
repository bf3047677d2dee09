#!/bin/bash
set -u
O=gpurun_out/r2b; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_autocorr.py -x -q -m gpu -k "three_trip or config3 or golden" > $O/t_autocorr8.log 2>&1; echo "autocorr rows8 rc=$?" | tee -a $O/summary.txt
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_rows8.json 2> $O/bench_rows8.err
TSDRGPU_ROWS16=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_rows16.json 2> $O/bench_rows16.err
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o trace -- python $R/bench.py --steps 2 --warmup 1 --passes 10 --no-cpu-baseline --no-profile > $R/$O/prof.log 2>&1
cd $R
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats.csv
find $O/prof -type f -size +1M -delete
python scripts/show_bench.py $O/bench_rows8.json $O/bench_rows16.json 2>&1 | grep -v "detected\|exact_autocorr\|whole_pass\|frame_path\|roofline" | cut -c1-900
cut -c1-150 $O/kernel_stats.csv | head -40
