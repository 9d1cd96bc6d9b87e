#!/bin/bash
# Round 5, GPU call 4: the GPU suite on the tree with the fused scorer statistics, the write-after-read probe with hipcc's own instruction group,
# and the default bench line.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_call4
mkdir -p $OUT
python3 -m dropoutdecoding_amd.build > $OUT/build.log 2>&1 || { echo "build failed"; tail -5 $OUT/build.log; exit 1; }
timeout 1800 python3 -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$? $(tail -3 $OUT/pytest_gpu.log | tr '\n' ' ')"
grep -n "^\[\|FAILED\|passed\|failed" $OUT/pytest_gpu.log | tail -15
tools/r05_fault_legs.sh 20 pkwar standalone
timeout 900 python3 bench.py > $OUT/bench_line.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-600 $OUT/bench_line.json
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r05_call4/bench_line.json"))
print({k:d[k] for k in ("value","ms_per_step")}, "single", d["single_stream"]["value"], "two-sweep", d["single_stream_two_sweep"]["value"], "det", d["determinism_check"])
r=d["roofline"]; print({k:r.get(k) for k in ("frac","frac_isolated","frac_source","traffic")}); print(r.get("finishing_share"))
PY
