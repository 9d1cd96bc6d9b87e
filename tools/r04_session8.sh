#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04_s8
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
DD_BISECT_ATTN=0 timeout 400 python tools/race_bisect.py fp32 12 3 > $O/bisect_plain.log 2>&1
DD_BISECT_ATTN=0 timeout 400 python tools/race_bisect.py fp32 12 3 "40=1" > $O/bisect_cu_masked.log 2>&1
DD_BISECT_ATTN=0 timeout 400 python tools/race_bisect.py fp32 12 3 "39=20480" > $O/bisect_ldspad20k.log 2>&1
DD_BISECT_ATTN=0 timeout 400 python tools/race_bisect.py fp32 12 3 "39=35840" > $O/bisect_ldspad35k.log 2>&1
DD_BISECT_ATTN=1 timeout 400 python tools/race_bisect.py fp32 12 3 > $O/bisect_traced.log 2>&1
tail -n 6 $O/bisect*.log | cut -c1-600
