#!/bin/bash
# full evidence run of round 6: GPU suite (with the device self-checks' output kept), smoke, kernel trace, PMC traffic (bench and
# stitch), driver-style bench, e2e legs, bounded differential soaks.  Run through gpurun; everything lands under gpurun_out/<tag>.
set -u
T=${1:-r6final}
O=gpurun_out/$T; mkdir -p $O
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu --maxfail=10 --durations=10 -rs -p no:cacheprovider > $O/tests.log 2>&1; echo "tests rc=$?" | tee $O/summary.txt; grep -E "passed|failed" $O/tests.log | tail -2
scripts/micro/arith_check > $O/arith_check.txt 2>&1; echo "arith_check rc=$?" | tee -a $O/summary.txt
scripts/micro/wave_reduce_check > $O/wave_reduce_check.txt 2>&1; echo "wave_reduce_check rc=$?" | tee -a $O/summary.txt
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt
# kernel trace of the bench, one lane
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o trace -- python $R/bench.py --steps 2 --warmup 1 --passes 10 --no-cpu-baseline --no-e2e --no-legs --serial --no-profile > $R/$O/prof.log 2>&1)
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats.csv
find $O/prof -type f -size +1M -delete
grep -v "at::native" $O/kernel_stats.csv | cut -c1-150 | head -24
bash scripts/pmc_collect.sh $T/pmc > $O/pmc.log 2>&1
cat $O/pmc/summary.txt | head -30
# the bench line's roofline.rocprof and roofline.traffic are read from profiles/: install THIS call's trace and counters first
[ -s $O/kernel_stats.csv ] && cp $O/kernel_stats.csv profiles/kernel_stats.csv
[ -s $O/pmc/pmc_traffic.json ] && cp $O/pmc/pmc_traffic.json profiles/pmc_traffic.json
S=$(date +%s.%N); timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; RC=$?; E_=$(date +%s.%N); echo "bench rc=$RC wall_s=$(python -c "print(round($E_-$S,1))") stdout_lines=$(wc -l < $O/bench.json) last_line_bytes=$(tail -1 $O/bench.json | wc -c)" | tee -a $O/summary.txt; cp gpurun_out/bench_detail.json $O/bench_detail.json
python scripts/show_bench.py $O/bench.json | cut -c1-2500
timeout 700 python scripts/e2e_bench.py --reference --variants > $O/e2e.json 2> $O/e2e.txt; echo "e2e rc=$?" | tee -a $O/summary.txt
grep -E "^(mi355x|reference|variant)" $O/e2e.txt | cut -c1-200
timeout 600 python scripts/fuzz_parity.py 2000 51 > $O/fuzz_parity.txt 2>&1; tail -3 $O/fuzz_parity.txt
timeout 600 python scripts/fuzz_engine.py 100 52 > $O/fuzz_engine.txt 2>&1; tail -3 $O/fuzz_engine.txt
