#!/bin/bash
# Round 5: grid / branch knobs of the rider step re-measured with every attention / finishing workgroup fitting beside every GEMV (tools keys 17-19:
# workgroups per K slice of the nine-plane qkv / o_proj / gate-up kernels; 28: branches; 21: attention key tiles per workgroup).
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r05_knobs
mkdir -p $O
python3 -m dropoutdecoding_amd.build > $O/build.log 2>&1 || { echo "build failed"; exit 1; }
timeout 1500 python3 tools/rider_ab.py 64 "28=4" "18=32" "17=64" "19=64" "28=3" "21=2" "21=1" "18=32,19=64" > $O/rider_ab.log 2>&1; cat $O/rider_ab.log | grep "ms per"
