#!/bin/bash
# Collects HBM traffic counters for the bench workload on the GPU box (run through gpurun).
# FETCH_SIZE needs 3 of the 4 TCC counter slots and WRITE_SIZE 2, so they are separate passes
# (MI355X_MICROARCH.md, "rocprofv3 PMC slots"); counters are collected with --kernel-trace only.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/pmc
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$C -o pmc -- \
      python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --pmc-calibrate > $OUT/$C.log 2>&1
done
cd $R
python scripts/pmc_summarize.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
