#!/bin/bash
# Builds tests/sanitize/host_stress_{tsan,asan}: the host library's three C files compiled WITH the driver under a sanitizer
# and linked against the in-tree libtsdrgpu.so (uninstrumented).  Test infrastructure only; outputs are git-ignored.
#   bash scripts/build_sanitized.sh            both
set -e
cd "$(dirname "$0")/.."
H=tempestsdr_amd/csrc/host
OUT=tests/sanitize
COMMON="-g -O1 -fno-omit-frame-pointer -std=gnu11 -Wall -Wextra -Wno-unused-parameter -Iinclude -I$H"
SRCS="$OUT/host_stress.c $H/tsdr_api.c $H/plugin_host.c $H/engine.c"
LINK="-Ltempestsdr_amd -ltsdrgpu -Wl,-rpath,\$ORIGIN/../../tempestsdr_amd -lpthread -ldl -lm"
gcc -fsanitize=thread $COMMON -o $OUT/host_stress_tsan $SRCS $LINK
gcc -fsanitize=address,undefined -fno-sanitize-recover=undefined $COMMON -o $OUT/host_stress_asan $SRCS $LINK
gcc $COMMON -O2 -o $OUT/host_stress_plain $SRCS $LINK
# the same against tests/sanitize/stub_tsdrgpu.c (host memory, no signal processing): runs without a GPU.  The stub itself
# is compiled without instrumentation (its fill loops would take the whole run under TSAN); its memcpy calls are still
# seen through the sanitizers' interceptors.
gcc -O2 -g -std=gnu11 -Iinclude -c -o $OUT/stub_tsdrgpu.o $OUT/stub_tsdrgpu.c
STUB="$OUT/stub_tsdrgpu.o -lpthread -ldl -lm"
gcc -fsanitize=thread $COMMON -o $OUT/host_stress_tsan_stub $SRCS $STUB
gcc -fsanitize=address,undefined -fno-sanitize-recover=undefined $COMMON -o $OUT/host_stress_asan_stub $SRCS $STUB
gcc $COMMON -O2 -o $OUT/host_stress_plain_stub $SRCS $STUB
# the multi-GPU sweep's C host (one thread and one rank per device) on the stand-in: its meeting points with more than one rank
gcc -fsanitize=thread $COMMON -o $OUT/host_stress_sweep_tsan_stub $H/tsdr_sweep.c $STUB
ls -la $OUT/host_stress_*
