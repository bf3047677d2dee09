#!/bin/bash
# full evidence run of round 4: GPU suite, smoke, kernel trace (+ every launch of k_sync_chain), PMC traffic, driver-style bench,
# e2e legs with the engine's switches, a bounded differential soak.  Run through gpurun; everything lands under gpurun_out/<tag>.
set -u
T=${1:-r4full}
O=gpurun_out/$T; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu --maxfail=10 --durations=10 > $O/tests.log 2>&1; echo "tests rc=$?" | tee $O/summary.txt; grep -E "passed|failed" $O/tests.log | tail -2
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt
# kernel trace; the per-launch rows of the sync chain are kept (a 1.27 ms outlier among 41 us launches)
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o trace -- python $R/bench.py --steps 2 --warmup 1 --passes 10 --no-cpu-baseline --no-e2e --no-legs --serial --no-profile > $R/$O/prof.log 2>&1)
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats.csv
t=$(find $O/prof -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python - "$t" > $O/sync_chain_launches.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
for i, r in enumerate(rows):
    if "k_sync_chain" in r["Kernel_Name"]:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        prev = names[i - 1][:40] if i else ""
        print(f"launch {i:6d}  {d:9.1f} us  stream/queue {r.get('Queue_Id', r.get('Stream_Id', '?'))}  after {prev}")
PY
find $O/prof -type f -size +1M -delete
grep -v "at::native" $O/kernel_stats.csv | cut -c1-150 | head -24
sort -k3 -n -r $O/sync_chain_launches.txt | head -5
bash scripts/pmc_collect.sh $T/pmc > $O/pmc.log 2>&1
cat $O/pmc/summary.txt | head -30
# the bench line's roofline.rocprof and roofline.traffic are read from profiles/: install THIS call's trace and counters first, so
# that the line, the kernel stats and the traffic are one build on one box
[ -s $O/kernel_stats.csv ] && cp $O/kernel_stats.csv profiles/kernel_stats.csv
[ -s $O/pmc/pmc_traffic.json ] && cp $O/pmc/pmc_traffic.json profiles/pmc_traffic.json
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/summary.txt
python scripts/show_bench.py $O/bench.json | cut -c1-1800
timeout 700 python scripts/e2e_bench.py --reference --variants > $O/e2e.json 2> $O/e2e.txt; echo "e2e rc=$?" | tee -a $O/summary.txt
grep -E "^(mi355x|reference|variant)" $O/e2e.txt | cut -c1-200
timeout 300 python scripts/fuzz_parity.py 300 > $O/fuzz_parity.txt 2>&1; tail -3 $O/fuzz_parity.txt
timeout 300 python scripts/fuzz_engine.py 30 > $O/fuzz_engine.txt 2>&1; tail -3 $O/fuzz_engine.txt
