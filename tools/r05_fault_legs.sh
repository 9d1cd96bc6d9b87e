#!/bin/bash
# Round 5: the sampler fault's experiments, one process per leg (tools/r05_sampler_fault.py); logs under gpurun_out/r05_fault/.
# usage: tools/r05_fault_legs.sh <seconds per leg> <leg> [<leg> ...]
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_fault
mkdir -p $OUT
SEC=$1; shift
python3 -m dropoutdecoding_amd.build > $OUT/build.log 2>&1 || { echo "build failed"; tail -5 $OUT/build.log; exit 1; }
for leg in "$@"; do
  echo "== leg $leg"
  if [ "$leg" = standalone ]; then
    hipcc --offload-arch=gfx950 -O3 -o $OUT/pkfma_war_repro tools/pkfma_war_repro.hip 2> $OUT/standalone_build.log && timeout 300 $OUT/pkfma_war_repro 40 | tee -a $OUT/legs.jsonl
    rm -f $OUT/pkfma_war_repro
    continue
  fi
  rounds=1000000
  [ "$leg" = pkwar ] && rounds=40
  timeout $((SEC + 240)) python3 tools/r05_sampler_fault.py $leg $rounds $OUT/legs.jsonl $SEC > $OUT/leg_$leg.out 2> $OUT/leg_$leg.err
  echo "rc=$? $(tail -c 600 $OUT/leg_$leg.out | cut -c1-600)"
done
