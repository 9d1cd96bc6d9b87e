#!/bin/bash
# Round 5, GPU call 9: the rows' rstd in a workgroup of its own + the slice-pair kernels' partial sums written at the end of the stream (tools/seq_lab.hip
# measured gate/up at 72 rows 48.1 -> 40.1 us alone): the slice / rider parity tests, the isolated GEMVs and the 64-lane / 56-lane steps, old vs new
# (tools key 50 = 0 and key 36 bit 8 restore the old placements; same bits either way).
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_call9
mkdir -p $OUT
python3 -m dropoutdecoding_amd.build > $OUT/build.log 2>&1 || { echo "build failed"; tail -5 $OUT/build.log; exit 1; }
timeout 120 tools/seq_lab > $OUT/seq_lab.log 2>&1; head -6 $OUT/seq_lab.log
timeout 900 python3 -m pytest tests/test_gpu_gemv_slices.py tests/test_gpu_rider.py tests/test_gpu_half_planes.py -m gpu -q -x > $OUT/pytest_slices.log 2>&1; echo "pytest slices+rider+half planes rc=$? $(tail -2 $OUT/pytest_slices.log | tr '\n' ' ')"
for a in "rows=72" "rows=72 50=0 36=8" "rows=32" "rows=32 50=0" "rows=16" "rows=16 50=0"; do echo "== gemv_times $a"; timeout 300 python3 tools/gemv_times.py $a 2>&1 | tail -1; done > $OUT/gemv_times.log 2>&1
cat $OUT/gemv_times.log
timeout 900 python3 tools/rider_ab.py 64 "50=1,36=0" "50=0,36=8" "50=1,36=8" "50=0,36=0" > $OUT/rider_ab.log 2>&1; grep "ms per" $OUT/rider_ab.log
DD_AB_K=4 timeout 600 python3 tools/rider_ab.py 56 "50=1,36=0" "50=0,36=8" > $OUT/rider_ab_k4.log 2>&1; grep "ms per" $OUT/rider_ab_k4.log
