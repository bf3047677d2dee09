#!/bin/bash
# full evidence run of round 3: GPU suite, smoke, driver-style bench, kernel trace, PMC traffic, SQ counters, e2e legs,
# the exact transform's profile, the Infinity-Cache micro-benchmark, a bounded differential soak.  Run through gpurun.
set -u
O=gpurun_out/${1:-r3full}; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" | tee $O/summary.txt; tail -n 3 $O/tests.log
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/summary.txt
bash scripts/gpu_prof.sh ${1:-r3full}/trace > /dev/null 2>&1
cp $O/trace/kernel_stats.csv $O/kernel_stats.csv 2>/dev/null
bash scripts/pmc_collect.sh ${1:-r3full}/pmc > $O/pmc.log 2>&1
bash scripts/pmc_sq.sh ${1:-r3full}/sq > $O/sq.txt 2>&1
python scripts/show_bench.py $O/bench.json | cut -c1-1500
cat $O/pmc/summary.txt | head -40
cat $O/sq.txt | tail -25
timeout 600 python scripts/e2e_bench.py --reference > $O/e2e.json 2> $O/e2e.txt; echo "e2e rc=$?" | tee -a $O/summary.txt
grep -E "^(mi355x|reference)" $O/e2e.txt | cut -c1-200
bash scripts/gpu_exact_prof.sh > $O/exact_prof.txt 2>&1; grep "us/window" $O/exact_prof.txt
(cd scripts/micro && hipcc --offload-arch=gfx950 -O2 -o mall_bench mall_bench.hip 2>/dev/null && ./mall_bench) > $O/mall_bench.txt 2>&1
(cd scripts/micro && hipcc --offload-arch=gfx950 -O2 -w -o tile_read_bench tile_read_bench.hip 2>/dev/null && ./tile_read_bench) > $O/tile_read_bench.txt 2>&1
timeout 400 python scripts/fuzz_parity.py 400 > $O/fuzz_parity.txt 2>&1; tail -3 $O/fuzz_parity.txt
timeout 400 python scripts/fuzz_engine.py 40 > $O/fuzz_engine.txt 2>&1; tail -3 $O/fuzz_engine.txt
