#!/bin/bash
# Builds tests/sanitize/host_stress_*: the host library's three C files compiled WITH the stress driver — plain, under
# ThreadSanitizer and under AddressSanitizer + UBSan — linked against the in-tree libtsdrgpu.so (uninstrumented) or against
# tests/sanitize/stub_tsdrgpu.c (host memory, no signal processing: runs without a GPU).  Test infrastructure only; outputs are
# git-ignored.  The plain variants come first: a host without libtsan / libasan still gets them (exit status 1 then).
cd "$(dirname "$0")/.."
H=tempestsdr_amd/csrc/host
OUT=tests/sanitize
COMMON="-g -O1 -fno-omit-frame-pointer -std=gnu11 -Wall -Wextra -Wno-unused-parameter -Iinclude -I$H"
SRCS="$OUT/host_stress.c $H/tsdr_api.c $H/plugin_host.c $H/engine.c"
LINK="-Ltempestsdr_amd -ltsdrgpu -Wl,-rpath,\$ORIGIN/../../tempestsdr_amd -lpthread -ldl -lm"
# the stand-in itself is compiled without instrumentation (its fill loops would take the whole run under TSAN); its memcpy calls
# are still seen through the sanitizers' interceptors
gcc -O2 -g -std=gnu11 -Iinclude -c -o $OUT/stub_tsdrgpu.o $OUT/stub_tsdrgpu.c || exit 2
STUB="$OUT/stub_tsdrgpu.o -lpthread -ldl -lm"
gcc $COMMON -O2 -o $OUT/host_stress_plain $SRCS $LINK || exit 2
gcc $COMMON -O2 -o $OUT/host_stress_plain_stub $SRCS $STUB || exit 2
rc=0
gcc -fsanitize=thread $COMMON -o $OUT/host_stress_tsan $SRCS $LINK || rc=1
gcc -fsanitize=address,undefined -fno-sanitize-recover=undefined $COMMON -o $OUT/host_stress_asan $SRCS $LINK || rc=1
gcc -fsanitize=thread $COMMON -o $OUT/host_stress_tsan_stub $SRCS $STUB || rc=1
gcc -fsanitize=address,undefined -fno-sanitize-recover=undefined $COMMON -o $OUT/host_stress_asan_stub $SRCS $STUB || rc=1
# the multi-GPU sweep's C host (one thread and one rank per device) on the stand-in: its meeting points with more than one rank
gcc -fsanitize=thread $COMMON -o $OUT/host_stress_sweep_tsan_stub $H/tsdr_sweep.c $STUB || rc=1
ls -la $OUT/host_stress_*
exit $rc
