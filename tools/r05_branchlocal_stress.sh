#!/bin/bash
# Round 5: the engine-level form in which rounds 3 and 4 SAW the difference (rider, K = 8, 64 lanes, every ring's masks sampled on its own branch:
# tools key 33 = 0) — with the product's one-wave sampler, and, as the positive control, with round 4's 1,024-thread kernel un-fenced (34 = 3, 48 = 0).
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r05_branchlocal
mkdir -p $O
python3 -m dropoutdecoding_amd.build > $O/build.log 2>&1 || { echo "build failed"; exit 2; }
export DD_STRESS_LOG=$O/stress.jsonl
DD_STRESS_TRACE=0 timeout 1500 python3 tools/stress_lanes.py 64 ${1:-100} 100 "33=0" > $O/wave_branchlocal.log 2>&1; echo "wave sampler, branch-local rc=$? $(tail -n 2 $O/wave_branchlocal.log | head -n 1 | cut -c1-200)"
DD_STRESS_TRACE=0 timeout 1500 python3 tools/stress_lanes.py 64 ${2:-100} 100 "33=0,34=3,48=0" > $O/block_unfenced_branchlocal.log 2>&1; echo "block sampler un-fenced, branch-local rc=$? $(tail -n 2 $O/block_unfenced_branchlocal.log | head -n 1 | cut -c1-200)"
