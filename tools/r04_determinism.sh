#!/bin/bash
# Round-4 determinism experiments, one GPU session.  Logs under gpurun_out/r04_det/.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04_det
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
# unit level: the sampler alone / beside 72-row GEMVs, both forms; the scratch probe
DD_REPRO_LOG=$O/sampler_repro.json timeout 900 python tools/sampler_repro.py ${REPRO_ROUNDS:-60} > $O/sampler_repro.log 2>&1
# fp32 cache, two branches vs solo (the second open case)
DD_AB_STEPS=6 timeout 600 python tools/lanes_mixed_ab.py fp32 11 > $O/fp32_ab.log 2>&1
DD_AB_STEPS=6 DD_AB_POISON=40 timeout 600 python tools/lanes_mixed_ab.py fp32 11 > $O/fp32_ab_poison.log 2>&1
# A: the round-3 sampler (scratch), branch-local sampling, traced
DD_STRESS_TRACE=1 DD_STRESS_STOP=4 DD_STRESS_LOG=$O/stress.jsonl timeout 1500 python tools/stress_lanes.py 64 ${A_REPS:-100} 100 "33=0,34=1" > $O/stress_A_scratch_branchlocal.log 2>&1
# B: the new sampler (no scratch), branch-local sampling, traced
DD_STRESS_TRACE=1 DD_STRESS_STOP=4 DD_STRESS_LOG=$O/stress.jsonl timeout 2400 python tools/stress_lanes.py 64 ${B_REPS:-200} 100 "33=0" > $O/stress_B_noscratch_branchlocal.log 2>&1
tail -n 3 $O/*.log
