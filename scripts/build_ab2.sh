#!/bin/bash
# A/B build of libtsdrgpu.so with extra flags for tsdrgpu_core.hip and tsdrgpu_frame.hip (e.g. -DTSDR_NT=7), into tempestsdr_amd/ab/<name>.so.
# Use with TSDRGPU_LIB=tempestsdr_amd/ab/<name>.so.  usage: scripts/build_ab2.sh <name> <flags...>
set -e
name=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
B=$R/tempestsdr_amd/build
mkdir -p $R/tempestsdr_amd/ab
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -fhip-fp32-correctly-rounded-divide-sqrt -I$R/include -ffp-contract=off"
/opt/rocm/bin/hipcc $F "$@" -c $R/tempestsdr_amd/csrc/tsdrgpu_core.hip -o $B/ab2_core_$name.o &
/opt/rocm/bin/hipcc $F "$@" -c $R/tempestsdr_amd/csrc/tsdrgpu_frame.hip -o $B/ab2_frame_$name.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tempestsdr_amd/ab/$name.so $B/ab2_core_$name.o $B/ab2_frame_$name.o $B/tsdrgpu_fft.o $B/tsdrgpu_fftx.o $B/tsdrgpu_extras.o $B/tsdrgpu_rccl.o -ldl
echo built tempestsdr_amd/ab/$name.so
