#!/bin/bash
set -u
O=gpurun_out/r2c; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_autocorr.py tests/test_gpu_postproc.py tests/test_gpu_soak.py tests/test_gpu_edges.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --fast-sync > $O/bench_fast.json 2> $O/bench_fast.err
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o trace -- python $R/bench.py --steps 2 --warmup 1 --passes 10 --no-cpu-baseline --no-profile > $R/$O/prof.log 2>&1
cd $R
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats.csv
find $O/prof -type f -size +1M -delete
tail -n 5 $O/tests.log
python scripts/show_bench.py $O/bench.json $O/bench_fast.json 2>&1 | grep -v "detected\|exact_autocorr\|whole_pass\|frame_path\|roofline\|kernels" | cut -c1-900
grep -v "at::native" $O/kernel_stats.csv | cut -c1-160 | head -30
