#!/bin/bash
# Round 5, GPU call 3: the GPU suite on the tree with the progressive stage-in (default on), then its A/B (tools key 49) per kernel and per step.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_call3
mkdir -p $OUT
python3 -m dropoutdecoding_amd.build > $OUT/build.log 2>&1 || { echo "build failed"; tail -5 $OUT/build.log; exit 1; }
timeout 1800 python3 -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$? $(tail -3 $OUT/pytest_gpu.log | tr '\n' ' ')"
grep -n "^\[\|FAILED\|passed\|failed" $OUT/pytest_gpu.log | tail -15
for a in "rows=72 49=0" "rows=72 49=1" "rows=64 49=0" "rows=64 49=1" "rows=32 49=0" "rows=32 49=1" "rows=16 49=0" "rows=16 49=1"; do echo "== gemv_times $a"; timeout 300 python3 tools/gemv_times.py $a 2>&1 | tail -1; done > $OUT/gemv_times.log 2>&1
cat $OUT/gemv_times.log
timeout 900 python3 tools/rider_ab.py 64 "49=0" "49=1" > $OUT/rider_ab.log 2>&1; tail -5 $OUT/rider_ab.log
DD_AB_K=4 timeout 600 python3 tools/rider_ab.py 56 "49=0" "49=1" > $OUT/rider_ab_k4.log 2>&1; tail -5 $OUT/rider_ab_k4.log
