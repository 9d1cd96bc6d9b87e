#!/bin/bash
# Round 5, GPU call 2: the one-wave sampler through the GPU suite, in the failing company, the barrier probe, and what the blocking stage-in of the
# operand planes costs (tools key 36 = 2: staging skipped, results garbage, timing only).
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_call2
mkdir -p $OUT
python3 -m dropoutdecoding_amd.build > $OUT/build.log 2>&1 || { echo "build failed"; tail -5 $OUT/build.log; exit 1; }
timeout 1500 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$? $(tail -3 $OUT/pytest_gpu.log | tr '\n' ' ')"
for a in "rows=72" "rows=72 36=2" "rows=64" "rows=64 36=2" "rows=32" "rows=32 36=2"; do echo "== gemv_times $a"; timeout 300 python3 tools/gemv_times.py $a 2>&1 | tail -2; done > $OUT/gemv_times.log 2>&1
tail -12 $OUT/gemv_times.log
timeout 600 python3 tools/rider_ab.py 64 "36=0" "36=2" "34=3" > $OUT/rider_ab.log 2>&1; tail -8 $OUT/rider_ab.log
tools/r05_fault_legs.sh 45 wave barrier1024 barrier256 pvprobe wave4 barrier1024_alone
