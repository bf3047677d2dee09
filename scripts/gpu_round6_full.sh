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
# SQ counters of the bench kernels, and the side legs (configs[0], [1], [4], 4 s batches, blur 0.5, unfused) in a run of their own
bash scripts/pmc_sq.sh $T/sq > $O/sq_counters.txt 2>&1; head -40 $O/sq_counters.txt | cut -c1-220
timeout 1800 python bench.py --gpus 1 --steps 20 --warmup 5 --legs --detail gpurun_out/$T/bench_legs_detail.json > $O/bench_legs.json 2> $O/bench_legs.err; echo "bench --legs rc=$?" | tee -a $O/summary.txt
# the band path on one rank (configs[4]) beside its plain leg, and the bands soak
B="--config 4 --seconds 0.5 --steps 10 --warmup 3 --no-cpu-baseline --no-e2e"
timeout 600 python bench.py $B > $O/c4_plain.json 2> $O/c4_plain.err
timeout 600 python bench.py $B --bands --force-dist > $O/c4_bands.json 2> $O/c4_bands.err
python - <<PY | tee -a $O/summary.txt
import json
for t in ("c4_plain","c4_bands"):
    try:
        d=json.loads([x for x in open("$O/%s.json"%t) if x.startswith("{")][-1]); print(t, d["value"], d["ms_per_pass"], d["config"].get("row_bands"))
    except Exception as e: print(t,"failed",e)
PY
FUZZ_ONLY=bands timeout 900 python scripts/fuzz_parity.py 600 63 > $O/fuzz_bands.txt 2>&1; tail -2 $O/fuzz_bands.txt
