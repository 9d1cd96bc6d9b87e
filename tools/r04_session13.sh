#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04_s13
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
DD_BISECT_ATTN=0 DD_BISECT_REPLAY=1 timeout 400 python tools/race_bisect.py fp32 12 3 > $O/bisect_replay.log 2>&1
grep -v amdgpu $O/bisect_replay.log | cut -c1-420
