#!/bin/bash
# full evidence run of round 5: GPU suite (with the device self-checks' output kept), smoke, kernel trace, PMC traffic (bench and
# stitch), driver-style bench, e2e legs, bounded differential soaks.  Run through gpurun; everything lands under gpurun_out/<tag>.
set -u
T=${1:-r5full}
O=gpurun_out/$T; mkdir -p $O
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -q -m gpu --maxfail=10 --durations=10 -rs > $O/tests.log 2>&1; echo "tests rc=$?" | tee $O/summary.txt; grep -E "passed|failed" $O/tests.log | tail -2
scripts/micro/arith_check > $O/arith_check.txt 2>&1; echo "arith_check rc=$?" | tee -a $O/summary.txt
scripts/micro/wave_reduce_check > $O/wave_reduce_check.txt 2>&1; echo "wave_reduce_check rc=$?" | tee -a $O/summary.txt
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt
# kernel trace of the bench, one lane
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o trace -- python $R/bench.py --steps 2 --warmup 1 --passes 10 --no-cpu-baseline --no-e2e --no-legs --serial --no-profile > $R/$O/prof.log 2>&1)
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats.csv
find $O/prof -type f -size +1M -delete
grep -v "at::native" $O/kernel_stats.csv | cut -c1-150 | head -24
# the stitch: kernel trace and counters (calibrated on a demodulation of known traffic in the same run)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stitch_prof -o t -- python $R/scripts/exp_stitch_prof.py > $R/$O/stitch.log 2>&1)
f=$(find $O/stitch_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/stitch_kernel_stats.csv
find $O/stitch_prof -type f -size +1M -delete
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/$O/stitch_pmc/$C -o pmc -- python $R/scripts/exp_stitch_prof.py --pmc-calibrate > $R/$O/stitch_pmc_$C.log 2>&1)
done
python scripts/pmc_summarize.py $O/stitch_pmc $O/stitch_pmc_traffic_full.json > $O/stitch_pmc_traffic.txt 2>&1; cat $O/stitch_pmc_traffic.txt
find $O/stitch_pmc -name "*.csv" -size +1M -delete
grep "^stitch" $O/stitch.log | tail -3; cut -c1-150 $O/stitch_kernel_stats.csv | head -10
bash scripts/pmc_collect.sh $T/pmc > $O/pmc.log 2>&1
cat $O/pmc/summary.txt | head -30
# the bench line's roofline.rocprof and roofline.traffic are read from profiles/: install THIS call's trace and counters first
[ -s $O/kernel_stats.csv ] && cp $O/kernel_stats.csv profiles/kernel_stats.csv
[ -s $O/pmc/pmc_traffic.json ] && cp $O/pmc/pmc_traffic.json profiles/pmc_traffic.json
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/summary.txt
python scripts/show_bench.py $O/bench.json | cut -c1-2500
timeout 700 python scripts/e2e_bench.py --reference --variants > $O/e2e.json 2> $O/e2e.txt; echo "e2e rc=$?" | tee -a $O/summary.txt
grep -E "^(mi355x|reference|variant)" $O/e2e.txt | cut -c1-200
timeout 600 python scripts/fuzz_parity.py 2000 51 > $O/fuzz_parity.txt 2>&1; tail -3 $O/fuzz_parity.txt
timeout 600 python scripts/fuzz_engine.py 100 52 > $O/fuzz_engine.txt 2>&1; tail -3 $O/fuzz_engine.txt
