#!/bin/bash
# Round 5: the fourth shipped form through the same stress — config 4 (InstructBLIP-Vicuna-7B: EVA ViT-g + Q-Former front-end, quantile masks, vote on
# the hidden state), GroupPipeline with its overlapped front-end + prefill, on libdropdec.so.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r05_stress_c4
mkdir -p $O
python3 -m dropoutdecoding_amd.build > $O/build.log 2>&1 || { echo "build failed"; exit 2; }
DD_STRESS_LOG=$O/stress.jsonl timeout 1500 python3 tools/stress_pipeline.py 4 ${1:-24} 4 > $O/pipeline_c4.log 2>&1; echo "pipeline_c4 rc=$? $(tail -n 2 $O/pipeline_c4.log | head -n 1 | cut -c1-300)"
