#!/bin/bash
# SQ counters of the stitch and of the bench on the final build, then the long differential soaks
set -u
T=${1:-r5soak}
O=gpurun_out/$T; mkdir -p $O
R=$GRAFT_REPO_ROOT
PMC_SQ_CMD="python $R/scripts/exp_stitch_prof.py" bash scripts/pmc_sq.sh $T/sq_stitch > $O/sq_stitch.txt 2>&1; cat $O/sq_stitch.txt | cut -c1-260
bash scripts/pmc_sq.sh $T/sq_bench > $O/sq_bench.txt 2>&1; grep -E "k_ac_|k_rs_area|k_frame_stats" $O/sq_bench.txt | cut -c1-260
for seed in 61 62 63 64; do timeout 900 python scripts/fuzz_parity.py 3000 $seed > $O/fuzz_parity_$seed.txt 2>&1; tail -1 $O/fuzz_parity_$seed.txt; done
for seed in 71 72; do timeout 900 python scripts/fuzz_engine.py 300 $seed > $O/fuzz_engine_$seed.txt 2>&1; tail -1 $O/fuzz_engine_$seed.txt; done
