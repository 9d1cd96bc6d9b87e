#!/bin/bash
# Round 5, the final GPU call on the final tree: GPU suite, profiles + bench lines, the 32-layer checkpoint load, then — the round's last GPU
# action — the repetition stress of the forms that ship.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O
python3 -m dropoutdecoding_amd.build > $O/r05_final_build.log 2>&1 || { echo "build failed"; tail -5 $O/r05_final_build.log; exit 1; }
timeout 1800 python3 -m pytest tests -m gpu -q > $O/r05_pytest_gpu_final.log 2>&1; rc=$?; echo "pytest rc=$rc $(tail -3 $O/r05_pytest_gpu_final.log | tr '\n' ' ')"
grep -n "^\[\|FAILED" $O/r05_pytest_gpu_final.log | tail -12
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/r05_smoke.log 2>&1; echo "smoke rc=$? $(tail -2 $O/r05_smoke.log | tr '\n' ' ' | cut -c1-200)"
tools/r05_final_measure.sh profiles benches
DD_CKPT_LAYERS=32 timeout 900 python3 -m pytest tests/test_gpu_checkpoint_load.py -m gpu -q -s > $O/r05_checkpoint_load_32_layers.log 2>&1; echo "ckpt32 rc=$? $(grep -a 'checkpoint:' $O/r05_checkpoint_load_32_layers.log | cut -c1-200)"
tools/r05_final_stress.sh ${1:-24} ${2:-20} ${3:-24} ${4:-100}
echo "stress rc=$?"
cp $O/r05_stress/stress.jsonl $O/r05_stress.jsonl
