#!/bin/bash
# round 6: the driver's bench command as the driver runs it — the ONE stdout line kept, its size and the wall clock beside it
set -u
T=${1:-r6bench}
O=gpurun_out/$T; mkdir -p $O
S=$(date +%s.%N)
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err; RC=$?
E_=$(date +%s.%N)
echo "bench rc=$RC wall_s=$(python -c "print(round($E_-$S,1))") stdout_lines=$(wc -l < $O/bench_line.json) last_line_bytes=$(tail -1 $O/bench_line.json | wc -c)" | tee $O/summary.txt
cp gpurun_out/bench_detail.json $O/bench_detail.json 2>/dev/null
tail -1 $O/bench_line.json
tail -5 $O/bench.err
