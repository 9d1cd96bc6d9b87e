#!/bin/bash
# Round 5, GPU call 7: the GPU suite on the tree with the bit plane in LDS (one-wave sampler) and K <= 64, then the round's profiles and bench lines.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_call7
mkdir -p $OUT
python3 -m dropoutdecoding_amd.build > $OUT/build.log 2>&1 || { echo "build failed"; tail -5 $OUT/build.log; exit 1; }
timeout 1800 python3 -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc $(tail -3 $OUT/pytest_gpu.log | tr '\n' ' ')"
grep -n "^\[\|FAILED" $OUT/pytest_gpu.log | tail -12
[ $rc -ne 0 ] && exit 1
tools/r05_final_measure.sh profiles benches
