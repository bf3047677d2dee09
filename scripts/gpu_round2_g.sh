#!/bin/bash
set -u
O=gpurun_out/r2o; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_autocorr.py tests/test_gpu_host_pipeline.py -x -q -m gpu -k "superb or superres or super" > $O/t_superb.log 2>&1; echo "superb tests rc=$?"; tail -n 4 $O/t_superb.log
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-e2e > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
grep -o '"superbandwidth": {[^}]*}' $O/bench.json
TSDR_GPU_STATS=1 timeout 300 python scripts/e2e_bench.py --seconds 3 > $O/e2e.json 2> $O/e2e.err; echo "e2e rc=$?"
grep -v "^$" $O/e2e.err | cut -c1-420 | tail -n 24
