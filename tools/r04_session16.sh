#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04_s16
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
for n in 12 16 10; do timeout 300 python tools/lanes_mixed_ab.py fp32 $n > $O/fp32_ab_$n.log 2>&1; done
DD_BISECT_ATTN=0 timeout 400 python tools/race_bisect.py fp32 12 3 > $O/bisect_product.log 2>&1
timeout 3000 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1
timeout 900 python tools/rider_ab.py 64 "26=1" "18=-2" > $O/rider_ab_64.log 2>&1
timeout 900 python bench.py --config 5 --steps 2 --warmup 1 --no-roofline --no-cpu-baseline --single-images 0 > $O/bench_c5.json 2> $O/bench_c5.err
tail -n 3 $O/fp32_ab_*.log $O/bisect_product.log | cut -c1-300; tail -n 6 $O/pytest_gpu.log | cut -c1-300; grep -v amdgpu $O/rider_ab_64.log; tail -c 400 $O/bench_c5.json
