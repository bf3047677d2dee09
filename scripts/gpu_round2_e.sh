#!/bin/bash
set -u
O=gpurun_out/r2m; mkdir -p $O
TSDR_GPU_STATS=1 timeout 300 python scripts/e2e_bench.py --seconds 3 > $O/e2e.json 2> $O/e2e.err; echo "e2e rc=$?"
grep -v "^$" $O/e2e.err | cut -c1-600 | tail -n 30
