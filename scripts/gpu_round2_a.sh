#!/bin/bash
# round 2, GPU call A: new FFT plan + exact-ties default: tests, bench variants, kernel trace
set -u
O=gpurun_out/r2a; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_autocorr.py -x -q -m gpu > $O/t_autocorr.log 2>&1; echo "autocorr rc=$?" | tee -a $O/summary.txt
timeout 900 python -m pytest tests/test_gpu_postproc.py tests/test_gpu_soak.py -x -q -m gpu > $O/t_postproc.log 2>&1; echo "postproc rc=$?" | tee -a $O/summary.txt
timeout 900 python -m pytest tests/test_gpu_host_pipeline.py -x -q -m gpu > $O/t_host.log 2>&1; echo "host rc=$?" | tee -a $O/summary.txt
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/summary.txt
timeout 300 python bench.py --steps 10 --warmup 3 --plan 5 --no-cpu-baseline > $O/bench_plan5.json 2> $O/bench_plan5.err
timeout 300 python bench.py --steps 10 --warmup 3 --fast-sync --no-cpu-baseline > $O/bench_fastsync.json 2> $O/bench_fastsync.err
timeout 300 python bench.py --steps 10 --warmup 3 --no-split --no-cpu-baseline > $O/bench_nosplit.json 2> $O/bench_nosplit.err
timeout 300 python bench.py --steps 5 --warmup 2 --config 1 --seconds 2 --no-cpu-baseline > $O/bench_cfg1.json 2> $O/bench_cfg1.err
timeout 300 python bench.py --steps 5 --warmup 2 --config 4 --seconds 0.5 --no-cpu-baseline > $O/bench_cfg4.json 2> $O/bench_cfg4.err
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --passes 10 --no-cpu-baseline --no-profile > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
find $O/prof -name "*kernel_stats*" | head -3 | while read f; do cp $f $O/kernel_stats.csv; done
find $O/prof -type f ! -name "*stats*" -size +2M -delete
tail -3 $O/t_*.log
python scripts/show_bench.py $O/bench.json $O/bench_plan5.json $O/bench_fastsync.json $O/bench_nosplit.json 2>&1 | cut -c1-1500
