#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04_s15
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
DD_REPRO_LOG=$O/probes.json timeout 900 python tools/sampler_repro.py 10 > $O/probes.log 2>&1
grep -E "pv_step|packed_fp32" $O/probes.log | cut -c1-330
