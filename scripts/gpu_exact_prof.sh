#!/bin/bash
# A/B of the exact transform's variants + a kernel trace and SQ counters of one of them.  Run through gpurun.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/exact_prof; mkdir -p $O
cd $R
for U in 1 2; do for XB in 4 8; do TSDRGPU_FFTX_U=$U TSDRGPU_XBATCH=$XB python scripts/exp_exact.py 2>/dev/null | tee -a $O/ab.txt; done; done
TSDRGPU_FFTX_GENERIC=1 python scripts/exp_exact.py 2>/dev/null | tee -a $O/ab.txt
TSDRGPU_FFTX_U=1 TSDRGPU_XBATCH=8 python scripts/exp_exact.py 200000000 8 3 2>/dev/null | tee -a $O/ab.txt
TSDRGPU_FFTX_U=1 TSDRGPU_XBATCH=8 python scripts/exp_exact.py 25000000 16 5 2>/dev/null | tee -a $O/ab.txt
cd /tmp; export TMPDIR=/tmp
for U in 1 2; do
TSDRGPU_FFTX_U=$U TSDRGPU_XBATCH=8 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace$U -o t -- python $R/scripts/exp_exact.py > $O/trace$U.log 2>&1
f=$(find $O/trace$U -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -c1-200 $f | head -12 | tee $O/kernel_stats_U$U.csv
done
TSDRGPU_FFTX_U=1 TSDRGPU_XBATCH=8 timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY \
   --output-format csv -d $O/sq -o sq -- python $R/scripts/exp_exact.py 100000000 16 2 > $O/sq.log 2>&1
cd $R
python - <<PY | tee $O/sq_summary.txt
import csv, glob, collections
fs = glob.glob("$O/sq/**/*counter_collection.csv", recursive=True)
if fs:
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].replace("void ", "")[:40]
        if not k.startswith("k_fftx"): continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES": cnt[k] += 1
    for k, c in sorted(agg.items()):
        n = max(cnt[k], 1); w = max(c["SQ_WAVES"], 1); wc = max(c["SQ_WAVE_CYCLES"], 1)
        print(f"{k:40s} launches {n:3d} VALU/wave {c['SQ_INSTS_VALU']/w:7.0f} LDS/wave {c['SQ_INSTS_LDS']/w:6.0f} VMEM_RD/wave {c['SQ_INSTS_VMEM_RD']/w:6.1f} "
              f"cycles/wave {c['SQ_WAVE_CYCLES']/w*4:8.0f} wait_any% {100*c['SQ_WAIT_ANY']/wc:5.1f} wait_inst% {100*c['SQ_WAIT_INST_ANY']/wc:5.1f} active_valu% {100*c['SQ_ACTIVE_INST_VALU']/wc:5.1f}")
PY
find $O -name "*.csv" -size +1M -delete; find $O -name "*.db" -delete 2>/dev/null
