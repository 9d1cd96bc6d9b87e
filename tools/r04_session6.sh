#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04_s6
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 500 python tools/race_bisect.py fp32 12 3 > $O/bisect_attn.log 2>&1
timeout 500 python tools/race_bisect.py fp32 12 3 "13=0" > $O/bisect_attn_noslices.log 2>&1
DD_REPRO_LOG=$O/sampler_repro.json timeout 600 python tools/sampler_repro.py 20 > $O/sampler_repro.log 2>&1
tail -n 12 $O/bisect*.log | cut -c1-600; tail -n 6 $O/sampler_repro.log | cut -c1-300
