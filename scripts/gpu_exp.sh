#!/bin/bash
set -u
timeout 900 python -m pytest tests/test_gpu_host_pipeline.py -x -q -m gpu 2>&1 | tail -2
timeout 300 python scripts/e2e_exp.py '[["pll_on","f32",{"PARAM_ID":"1"}]]' 2>&1 | grep -E "MS/s|tsdr stats"
