#!/bin/bash
# A/B build of libtsdrgpu.so: the same tree with extra flags for tsdrgpu_fft.hip (e.g. -DSOME_EXPERIMENT), into tempestsdr_amd/ab/<name>.so.
# Use with TSDRGPU_LIB=tempestsdr_amd/ab/<name>.so (tempestsdr_amd/gpu.py).  usage: scripts/build_ab.sh <name> <flags...>
set -e
name=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
B=$R/tempestsdr_amd/build
mkdir -p $R/tempestsdr_amd/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -fhip-fp32-correctly-rounded-divide-sqrt -I$R/include -fno-slp-vectorize "$@" \
    -c $R/tempestsdr_amd/csrc/tsdrgpu_fft.hip -o $B/ab_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tempestsdr_amd/ab/$name.so $B/tsdrgpu_core.o $B/tsdrgpu_frame.o $B/ab_$name.o $B/tsdrgpu_fftx.o $B/tsdrgpu_extras.o $B/tsdrgpu_rccl.o -ldl
echo built tempestsdr_amd/ab/$name.so
