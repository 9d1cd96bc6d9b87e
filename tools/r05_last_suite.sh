#!/bin/bash
# The round's very last GPU call: the GPU suite + smoke() on the committed final tree (after the stress only comments and documents changed).
set -u
cd "$(dirname "$0")/.."
O=gpurun_out
python3 -m dropoutdecoding_amd.build > $O/r05_last_build.log 2>&1 || { echo "build failed"; exit 1; }
timeout 1800 python3 -m pytest tests -m gpu -q > $O/r05_pytest_gpu_last.log 2>&1; echo "pytest rc=$? $(tail -3 $O/r05_pytest_gpu_last.log | tr '\n' ' ' | cut -c1-300)"
grep -n "sampler beside rider steps" $O/r05_pytest_gpu_last.log | head -3
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/r05_smoke_last.log 2>&1; echo "smoke rc=$? $(tail -1 $O/r05_smoke_last.log | cut -c1-200)"
