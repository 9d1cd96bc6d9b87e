#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04_s9
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
DD_BISECT_ATTN=0 timeout 400 python tools/race_bisect.py fp32 12 3 "41=1" > $O/bisect_attn_masked_disjoint.log 2>&1
DD_BISECT_ATTN=0 timeout 400 python tools/race_bisect.py fp32 12 3 "41=2" > $O/bisect_attn_masked_same_half.log 2>&1
tail -n 6 $O/bisect*.log | cut -c1-400
