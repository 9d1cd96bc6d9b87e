#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04_s7
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
DD_OVERLAP_LOG=$O/lds_overlap.json timeout 600 python tools/lds_overlap.py > $O/lds_overlap.log 2>&1
DD_BISECT_ATTN=0 timeout 400 python tools/race_bisect.py fp32 12 3 > $O/bisect_noattntrace.log 2>&1
tail -n 14 $O/lds_overlap.log; tail -n 5 $O/bisect_noattntrace.log | cut -c1-500
