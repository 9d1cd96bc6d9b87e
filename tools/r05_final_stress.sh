#!/bin/bash
# Round 5's repetition stress of the forms that ship, on libdropdec.so.  Every leg's exit code is kept: the script
# ends non-zero when any leg saw a differing repetition.  Summaries: gpurun_out/r05_stress/stress.jsonl (copied to profiles/r05_stress.jsonl).
#   usage: tools/r05_final_stress.sh [pipeline reps (config 3)] [config 5 reps] [config 2 reps] [lanes reps]
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r05_stress
mkdir -p $O
python3 -m dropoutdecoding_amd.build > $O/build.log 2>&1 || { echo "build failed"; exit 2; }
export DD_STRESS_LOG=$O/stress.jsonl
rc=0
run() { name=$1; shift; "$@" > $O/$name.log 2>&1; r=$?; echo "$name rc=$r: $(tail -n 2 $O/$name.log | head -n 1 | cut -c1-300)"; [ $r -ne 0 ] && rc=1; }
# GroupPipeline as bench.py drives it: config 3 (K = 8, 64 images per batch, overlapped prefill), 4 batches = 512 group steps per repetition
run pipeline_c3 timeout 1500 python3 tools/stress_pipeline.py 3 ${1:-20} 4
# config 5: fp8 nine-plane rider step, 2928 visual tokens, 64 images per batch
run pipeline_c5 timeout 2400 python3 tools/stress_pipeline.py 5 ${2:-20} 4
# config 2: K = 4 half planes, groups of fourteen, 56 images per batch
run pipeline_c2 timeout 1200 python3 tools/stress_pipeline.py 2 ${3:-20} 4
# the lane-level stress of round 4 on the SHIPPED library (default staged sampling, no tuning keys)
run lanes_rider_k8_64 timeout 1500 python3 tools/stress_lanes.py 64 ${4:-100} 100
exit $rc
