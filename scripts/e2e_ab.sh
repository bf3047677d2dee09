#!/bin/bash
# Whole-library throughput (tsdr_* API, in-memory plugin) against the number of hardware queues the HIP runtime may
# use, first and second engine of a process.  Run through gpurun; prints one line per process.
run() { env $2 timeout 300 python scripts/e2e_exp.py "$1" 2>&1 | grep -E "MS/s" | tr '\n' ' '; echo " [$2]"; }
LEGS='[["first_engine","f32",{}],["second_engine","f32",{}]]'
run "$LEGS" LIBRARY_DEFAULT=2
for q in 1 2 3 4 8; do run "$LEGS" GPU_MAX_HW_QUEUES=$q; done
run '[["detector_on_background_lane","f32",{"TSDR_GPU_DETECTOR_LANE":"background"}],["second_engine","f32",{"TSDR_GPU_DETECTOR_LANE":"background"}]]' LIBRARY_DEFAULT=2
run '[["int16_recording","i16",{}],["plots_off","f32",{"PARAM_ID":"3"}]]' LIBRARY_DEFAULT=2
cat /proc/loadavg
