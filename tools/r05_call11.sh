#!/bin/bash
# Round 5, GPU call 11: the headline with the old and the new placements (rstd workgroup, partial-sum stores) on ONE box — bench.py on
# libdropdec_tools.so (the product's kernels + the experiment switches), DD_TOOLS_TUNE sets the keys.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r05_call11
mkdir -p $O
python3 -m dropoutdecoding_amd.build > $O/build.log 2>&1 || { echo "build failed"; tail -5 $O/build.log; exit 1; }
for t in "50=1,36=0" "50=0,36=8" "50=1,36=0" "50=0,36=8"; do
  DD_USE_TOOLS_LIB=1 DD_TOOLS_TUNE="$t" timeout 600 python3 bench.py --no-cpu-baseline --no-roofline --single-images 0 --no-determinism-check > $O/bench_$t.json 2> $O/bench_$t.err
  echo "$t rc=$? $(python3 -c "import json,sys; d=json.loads(open('$O/bench_$t.json').read()); print(d['value'], d['ms_per_step'])")"
done
