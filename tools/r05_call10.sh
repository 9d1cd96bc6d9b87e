#!/bin/bash
# Round 5, GPU call 10 (after the final sequence): the placement A/B test added to tests/test_gpu_rider.py, and the headline bench once more on another
# box (the final sequence's box ran every unchanged kernel ~2.5 % slower than the box of the previous final sequence).
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r05_call10
mkdir -p $O
python3 -m dropoutdecoding_amd.build > $O/build.log 2>&1 || { echo "build failed"; tail -5 $O/build.log; exit 1; }
timeout 600 python3 -m pytest tests/test_gpu_rider.py -m gpu -q -k "rstd_workgroup" > $O/pytest_placement.log 2>&1; echo "pytest placement rc=$? $(tail -2 $O/pytest_placement.log | tr '\n' ' ')"
timeout 900 python3 bench.py > $O/bench_line.json 2> $O/bench_line.err; echo "bench c3 rc=$? $(cut -c1-140 $O/bench_line.json)"
timeout 300 python3 tools/rider_ab.py 64 "50=1,36=0" "50=0,36=8" > $O/rider_ab.log 2>&1; grep "ms per" $O/rider_ab.log
