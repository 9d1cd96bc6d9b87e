#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04_s10
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
for m in 128 15 48 1 2 4 8; do
  DD_BISECT_ATTN=0 timeout 300 python tools/race_bisect.py fp32 12 3 "40=1,42=$m" > $O/bisect_unmask_$m.log 2>&1
done
for f in $O/bisect_unmask_*.log; do echo "== $f"; grep -c "trace cells differ" $f; grep "rep" $f | cut -c1-330 | head -4; done
