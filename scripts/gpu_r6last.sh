#!/bin/bash
# round 6, last call: the GPU suite and the driver's bench on the final tree, then longer soaks
set -u
T=${1:-r6last}
O=gpurun_out/$T; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/tests.log 2>&1; echo "tests (-x, the driver's command) rc=$? $(grep -E 'passed|failed' $O/tests.log | tail -1)" | tee $O/summary.txt
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt
S=$(date +%s.%N); timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; RC=$?; E_=$(date +%s.%N)
echo "bench rc=$RC wall_s=$(python -c "print(round($E_-$S,1))") stdout_lines=$(wc -l < $O/bench.json) last_line_bytes=$(tail -1 $O/bench.json | wc -c)" | tee -a $O/summary.txt
tail -1 $O/bench.json | cut -c1-700
timeout 1500 python scripts/fuzz_parity.py 6000 71 > $O/fuzz_parity.txt 2>&1; tail -2 $O/fuzz_parity.txt | tee -a $O/summary.txt
timeout 1500 python scripts/fuzz_engine.py 300 72 > $O/fuzz_engine.txt 2>&1; tail -1 $O/fuzz_engine.txt | tee -a $O/summary.txt
FUZZ_ONLY=bands timeout 900 python scripts/fuzz_parity.py 3000 73 > $O/fuzz_bands.txt 2>&1; tail -2 $O/fuzz_bands.txt | tee -a $O/summary.txt
