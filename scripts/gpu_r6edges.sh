#!/bin/bash
# round 6: the edge-parity tests (non-finite input, one-row / one-column frames, strips longer than the LDS) and the reference's JNI
# shim on top of the drop-in, each file in a process of its own
set -u
T=${1:-r6edges}
O=gpurun_out/$T; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_edges.py -q -m gpu --maxfail=30 --tb=short -p no:cacheprovider > $O/edges.log 2>&1; echo "edges rc=$?" | tee $O/summary.txt; tail -3 $O/edges.log
timeout 1500 python -m pytest tests/test_gpu_nonfinite.py -q -m gpu --maxfail=60 --tb=line -p no:cacheprovider > $O/nonfinite.log 2>&1; echo "nonfinite rc=$?" | tee -a $O/summary.txt; tail -3 $O/nonfinite.log
timeout 600 python -m pytest tests/test_jni_shim.py -q --maxfail=10 --tb=short -rs -p no:cacheprovider > $O/jni.log 2>&1; echo "jni rc=$?" | tee -a $O/summary.txt; tail -30 $O/jni.log | cut -c1-250
grep -E "^FAILED|^ERROR|Error|assert" $O/nonfinite.log | cut -c1-260 | head -70
grep -E "^FAILED|^ERROR" $O/edges.log | cut -c1-260 | head -30
