#!/bin/bash
set -u
O=gpurun_out/r2n; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_distributed.py -x -q -m gpu > $O/t_dist.log 2>&1; echo "dist rc=$?"; tail -n 12 $O/t_dist.log
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --force-dist > $O/bench_dist.json 2> $O/bench_dist.err; echo "bench force-dist rc=$?"; tail -n 3 $O/bench_dist.err
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --force-dist --scaling strong > $O/bench_dist_strong.json 2> $O/bench_dist_strong.err; echo "bench strong rc=$?"; tail -n 3 $O/bench_dist_strong.err
python scripts/show_bench.py $O/bench_dist.json $O/bench_dist_strong.json 2>&1 | grep "value\|detected"
