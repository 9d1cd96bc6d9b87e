#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04_s14
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
DD_BISECT_ATTN=0 timeout 400 python tools/race_bisect.py fp32 12 3 "43=1" > $O/bisect_nopk.log 2>&1
DD_BISECT_ATTN=0 timeout 400 python tools/race_bisect.py fp32 12 3 > $O/bisect_plain.log 2>&1
DD_REPRO_LOG=$O/probes.json timeout 900 python tools/sampler_repro.py 10 > $O/probes.log 2>&1
grep -v amdgpu $O/bisect_nopk.log | cut -c1-330; grep -v amdgpu $O/bisect_plain.log | cut -c1-200 | tail -4; grep packed_fp32 $O/probes.log | cut -c1-330
