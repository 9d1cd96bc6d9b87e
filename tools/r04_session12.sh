#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04_s12
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
DD_REPRO_LOG=$O/probes.json timeout 900 python tools/sampler_repro.py 12 > $O/probes.log 2>&1
grep -E "lds_full" $O/probes.log | cut -c1-330
