#!/bin/bash
# Round 5, GPU call 5: two tiles' folded sums per register set in the slice-pair kernels (gate/up 222 -> 192 VGPRs: fits beside the rider attention;
# room for eight weight requests in flight, tools key 29 = 8): bit-exactness suites, then the A/B per kernel and per step.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_call5
mkdir -p $OUT
python3 -m dropoutdecoding_amd.build > $OUT/build.log 2>&1 || { echo "build failed"; tail -5 $OUT/build.log; exit 1; }
timeout 1800 python3 -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$? $(tail -3 $OUT/pytest_gpu.log | tr '\n' ' ')"
for a in "rows=72 29=4" "rows=72 29=8" "rows=64"; do echo "== gemv_times $a"; timeout 300 python3 tools/gemv_times.py $a 2>&1 | tail -1; done > $OUT/gemv_times.log 2>&1
cat $OUT/gemv_times.log
timeout 900 python3 tools/rider_ab.py 64 "29=4" "29=8" "29=8,47=1" > $OUT/rider_ab.log 2>&1; tail -7 $OUT/rider_ab.log
DD_AB_K=4 timeout 600 python3 tools/rider_ab.py 56 "29=4" "29=8" > $OUT/rider_ab_k4.log 2>&1; tail -5 $OUT/rider_ab_k4.log
