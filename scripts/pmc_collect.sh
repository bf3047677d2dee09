#!/bin/bash
# Collects HBM traffic counters for the bench workload on the GPU box (run through gpurun).
# FETCH_SIZE needs 3 of the 4 TCC counter slots and WRITE_SIZE 2, so they are separate passes
# (MI355X_MICROARCH.md, "rocprofv3 PMC slots"); counters are collected with --kernel-trace only.
# usage: pmc_collect.sh [out_dir_under_gpurun_out]
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/${1:-pmc}
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$C -o pmc -- \
      python $R/bench.py --steps 1 --warmup 1 --passes 4 --no-cpu-baseline --no-e2e --no-legs --serial --no-profile --pmc-calibrate > $OUT/$C.log 2>&1
done
cd $R
PMC_SOURCE="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of 'bench.py --steps 1 --warmup 1 --passes 4 --pmc-calibrate' at commit $(cat $R/.commit_id 2>/dev/null || echo unknown), scripts/pmc_collect.sh (not collected live)" \
  python scripts/pmc_summarize.py $OUT $OUT/pmc_traffic_full.json $OUT/pmc_traffic.json > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
find $OUT -name "*.csv" -size +1M -delete
