#!/bin/bash
# SQ counters for the bench kernels (instruction mix / stall picture).  Run through gpurun.
# usage: pmc_sq.sh [out_dir_under_gpurun_out]      PMC_SQ_CMD="python scripts/exp_stitch_prof.py" profiles another workload (the stitch)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/${1:-pmc_sq}
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES \
   --output-format csv -d $OUT -o sq -- ${PMC_SQ_CMD:-python $R/bench.py --steps 1 --warmup 1 --passes 4 --no-cpu-baseline --no-e2e --no-legs --serial --no-profile} > $OUT/run.log 2>&1
cd $R
python - <<PY
import csv, glob, collections, re
f = glob.glob("$OUT/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if not k.startswith("k_"): continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES": cnt[k] += 1
for k, c in sorted(agg.items()):
    n = max(cnt[k], 1); w = max(c["SQ_WAVES"], 1)
    print(f"{k[:34]:34s} launches {n:3d} waves/launch {c['SQ_WAVES']/n:10.0f} VALU/wave {c['SQ_INSTS_VALU']/w:8.0f} SALU/wave {c['SQ_INSTS_SALU']/w:7.0f} VMEM_RD/wave {c['SQ_INSTS_VMEM_RD']/w:6.1f} VMEM_WR/wave {c['SQ_INSTS_VMEM_WR']/w:6.1f} wave_cycles/wave {c['SQ_WAVE_CYCLES']/w*4:9.0f} wait_inst% {100*c['SQ_WAIT_INST_ANY']/max(c['SQ_WAVE_CYCLES'],1):5.1f} active_valu% {100*c['SQ_ACTIVE_INST_VALU']/max(c['SQ_WAVE_CYCLES'],1):5.1f}")
PY
find $OUT -name "*.csv" -size +1M -delete
