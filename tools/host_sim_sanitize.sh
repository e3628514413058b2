#!/bin/bash
# host_sim_sanitize.sh — the kernels' decode logic compiled for the host (tests/host_sim/lane_sim.cpp: alac_wave.h, alac_regular.h,
# alac_duo.h, alac_split.h with a one-lane wave policy) under AddressSanitizer + UndefinedBehaviorSanitizer, driven by the CPU
# suite's lane-logic tests (intact, truncated and mutated packets, dense blobs against a guard page, every cookie byte).
# GPU AddressSanitizer is not available on the pool; this is the sanitizer run of the same source. signed-integer-overflow is
# off: the build defines it (-fwrapv, as the GPU build does). usage: tools/host_sim_sanitize.sh   (from the repo root, CPU only)
set -e
so=tests/host_sim/liblane_sim.so
g++ -O1 -g -fwrapv -fPIC -std=c++17 -Wno-unknown-pragmas -fsanitize=address,undefined -fno-sanitize-recover=undefined \
    -fno-sanitize=signed-integer-overflow -shared -o $so tests/host_sim/lane_sim.cpp
touch $so
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1 \
    python -m pytest tests/test_lane_logic.py -x -q -m "not gpu"
rm -f $so   # the next test run rebuilds the plain one
