#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04_s4
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 400 python tools/race_bisect.py fp32 12 3 > $O/bisect_fp32_12.log 2>&1
timeout 400 python tools/race_bisect.py fp32 12 3 "13=0" > $O/bisect_fp32_12_noslices.log 2>&1
timeout 400 python tools/race_bisect.py fp32 12 3 "10=2" > $O/bisect_fp32_12_attnsplit2.log 2>&1
timeout 400 python tools/race_bisect.py fp32 12 3 "9=4" > $O/bisect_fp32_12_pairs4.log 2>&1
timeout 400 python tools/race_bisect.py fp32 10 3 > $O/bisect_fp32_10.log 2>&1
timeout 400 python tools/race_bisect.py fp16 12 3 > $O/bisect_fp16_12.log 2>&1
GPU_MAX_HW_QUEUES=1 timeout 400 python tools/race_bisect.py fp32 12 3 > $O/bisect_fp32_12_one_hw_queue.log 2>&1
DD_AB_POISON=40 timeout 300 python tools/lanes_mixed_ab.py fp32 12 "37=1" > $O/fp32_fork_12_poison.log 2>&1
# config 5: GQA attention with all heads of a kv group per workgroup (key 38), 16 and 32 images per step
for t in "0" "38=1"; do
  DD_USE_TOOLS_LIB=1 DD_TOOLS_TUNE="$t" timeout 900 python bench.py --config 5 --images-per-gpu 16 --steps 2 --warmup 1 --no-roofline --no-cpu-baseline --single-images 0 > $O/bench_c5_b16_tune_$t.json 2> $O/bench_c5_b16_tune_$t.err
done
tail -n 8 $O/bisect*.log $O/fp32*.log | cut -c1-400; tail -c 300 $O/bench_c5_b16_tune_*.json
