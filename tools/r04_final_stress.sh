#!/bin/bash
# The round's LAST GPU action: repetition stress of the final tree, every repetition against the first and one lane
# per repetition against its solo run.  Summaries -> profiles/r04_stress.jsonl (copied from gpurun_out by the caller).
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04_stress
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
export DD_STRESS_LOG=$O/stress.jsonl
# rider form, K = 8, 64 lanes (the bench's configuration; staged sampling = the default)
timeout 2400 python tools/stress_lanes.py 64 ${RIDER_REPS:-200} 100 > $O/rider_k8_64.log 2>&1
# the same with the masks sampled on the branches (key 33 = 0): the form in which round 3 saw its differences
timeout 1500 python tools/stress_lanes.py 64 ${BRANCH_REPS:-100} 100 "33=0" > $O/rider_k8_64_branchlocal.log 2>&1
# half planes, K = 4, 56 lanes (groups of fourteen)
DD_STRESS_K=4 timeout 900 python tools/stress_lanes.py 56 ${HP_REPS:-30} 100 > $O/halfplanes_k4_56.log 2>&1
# classic form, 32 lanes
timeout 900 python tools/stress_lanes.py 32 ${CLASSIC_REPS:-30} 100 "26=0" > $O/classic_k8_32.log 2>&1
# InstructBLIP (quantile masks, vote on the hidden state), 64 lanes
DD_STRESS_FAMILY=iblip timeout 900 python tools/stress_lanes.py 64 ${IBLIP_REPS:-30} 100 > $O/iblip_k8_64.log 2>&1
# fp32 cache, classic form on two branches, 16 lanes (round 3 kept these on one branch)
DD_STRESS_KV=fp32 timeout 900 python tools/stress_lanes.py 16 ${FP32_REPS:-30} 100 > $O/classic_fp32_16.log 2>&1
tail -n 2 $O/*.log
