#!/bin/bash
run() { env $2 timeout 300 python scripts/e2e_exp.py "$1" 2>&1 | grep -E "MS/s" | tr '\n' ' '; echo " [$2]"; }
run '[["first","f32",{}],["second","f32",{}]]' A=1
run '[["first","f32",{}],["second","f32",{}],["third_i16","i16",{}]]' TSDRGPU_EVENT_NOFENCE=1
run '[["first","f32",{}],["second","f32",{}]]' "TSDRGPU_EVENT_NOFENCE=1 GPU_MAX_HW_QUEUES=3"
run '[["first","f32",{}],["second","f32",{}]]' "TSDRGPU_EVENT_NOFENCE=1 GPU_MAX_HW_QUEUES=8"
run '[["first_bg","f32",{"TSDR_GPU_DETECTOR_LANE":"background"}],["second_bg","f32",{"TSDR_GPU_DETECTOR_LANE":"background"}]]' "TSDRGPU_EVENT_NOFENCE=1"
TSDRGPU_EVENT_NOFENCE=1 timeout 900 python -m pytest tests/test_gpu_host_pipeline.py -x -q -m gpu 2>&1 | tail -2
