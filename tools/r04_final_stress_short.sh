#!/bin/bash
# After the sampler's LDS padding (the round's last code change): the sampler / rider tests, then the two rider forms of the repetition stress —
# masks sampled on the branches (the form that differed) and the staged default — 10,000 steps each.  Budgeted for the GPU minutes that were left.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04_stress3
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/build_smoke.log 2>&1; echo "build+smoke rc=$?"
timeout 420 python -m pytest tests/test_gpu_dropout_ops.py tests/test_gpu_sampler_repro.py tests/test_gpu_rider.py tests/test_gpu_speculative_step.py -x -q > $O/pytest.log 2>&1; tail -n 2 $O/pytest.log
export DD_STRESS_LOG=$O/stress.jsonl
timeout 720 python tools/stress_lanes.py 64 ${BRANCH_REPS:-100} 100 "33=0" > $O/rider_k8_64_branchlocal.log 2>&1
timeout 720 python tools/stress_lanes.py 64 ${RIDER_REPS:-100} 100 > $O/rider_k8_64.log 2>&1
tail -n 2 $O/rider_*.log | cut -c1-500
