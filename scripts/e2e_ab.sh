#!/bin/bash
run() { env $2 timeout 300 python scripts/e2e_exp.py "$1" 2>&1 | grep -E "MS/s" | tr '\n' ' '; echo " [$2]"; }
run '[["first","f32",{}],["second","f32",{}],["third_i16","i16",{}]]' A=1
run '[["first","f32",{}],["second_fast","f32",{"TSDR_GPU_EXACT":"0"}]]' A=2
run '[["first","f32",{}]]' GPU_MAX_HW_QUEUES=4
