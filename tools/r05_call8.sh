#!/bin/bash
# Round 5, GPU call 8: the batched one-wave sampler against its suites; what the partial sums and the stage-in of the 72-row kernels cost
# (tools key 36: 2 = no stage-in, 4 = no partial sums written, 6 = neither: timing only, results garbage).
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_call8
mkdir -p $OUT
python3 -m dropoutdecoding_amd.build > $OUT/build.log 2>&1 || { echo "build failed"; tail -5 $OUT/build.log; exit 1; }
timeout 1500 python3 -m pytest tests/test_gpu_dropout_ops.py tests/test_gpu_engine.py tests/test_gpu_sampler_repro.py tests/test_gpu_rider.py tests/test_gpu_half_planes.py tests/test_gpu_speculative_step.py tests/test_gpu_eos_stream.py tests/test_gpu_wrappers.py -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$? $(tail -3 $OUT/pytest_gpu.log | tr '\n' ' ')"
grep -n "^\[\|FAILED" $OUT/pytest_gpu.log | tail -12
for a in "rows=72" "rows=72 36=4" "rows=72 36=6" "rows=64 36=4"; do echo "== gemv_times $a"; timeout 300 python3 tools/gemv_times.py $a 2>&1 | tail -1; done > $OUT/gemv_times.log 2>&1
cat $OUT/gemv_times.log
timeout 600 python3 tools/rider_ab.py 64 "33=1" "33=0" > $OUT/rider_ab.log 2>&1; tail -4 $OUT/rider_ab.log
