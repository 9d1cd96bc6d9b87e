#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04_s19
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 900 python tools/prefill_ab.py > $O/prefill_ab.log 2>&1
timeout 1200 python -m pytest tests/test_gpu_prefill_kernels.py -x -q > $O/pytest_prefill_kernels.log 2>&1
grep -v amdgpu $O/prefill_ab.log | cut -c1-250; tail -n 3 $O/pytest_prefill_kernels.log
