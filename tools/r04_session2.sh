#!/bin/bash
# Round-4 GPU session 2: fp32 two-branch case at the lane counts that failed in round 3, the round-3 stress configuration without trace / solo,
# the new tests, config 5 bench at several batch sizes.  Logs under gpurun_out/r04_s2/.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04_s2
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
for n in 12 16 10; do
  timeout 300 python tools/lanes_mixed_ab.py fp32 $n > $O/fp32_ab_$n.log 2>&1
done
timeout 300 python tools/lanes_mixed_ab.py fp32 12 "13=0" > $O/fp32_ab_12_noslices.log 2>&1
timeout 300 python tools/lanes_mixed_ab.py fp32 12 "8=0" > $O/fp32_ab_12_nograph.log 2>&1
timeout 300 python tools/lanes_mixed_ab.py fp32 12 "10=2" > $O/fp32_ab_12_attnsplit2.log 2>&1
DD_AB_POISON=40 timeout 300 python tools/lanes_mixed_ab.py fp32 12 > $O/fp32_ab_12_poison.log 2>&1
timeout 2400 python -m pytest tests/test_gpu_rider.py tests/test_gpu_gemv_slices.py tests/test_gpu_checkpoint_load.py tests/test_gpu_dist_nccl.py tests/test_gpu_sampler_repro.py -x -q -m gpu -s > $O/pytest_new.log 2>&1
for b in 8 16 32; do
  timeout 900 python bench.py --config 5 --images-per-gpu $b --steps 2 --warmup 1 --no-roofline --no-cpu-baseline --single-images 0 > $O/bench_c5_b$b.json 2> $O/bench_c5_b$b.err
done
DD_STRESS_TRACE=0 DD_STRESS_SOLO=0 DD_STRESS_STOP=3 DD_STRESS_LOG=$O/stress.jsonl timeout 1200 python tools/stress_lanes.py 64 100 100 "33=0,34=1" > $O/stress_A2_scratch_branchlocal_plain.log 2>&1
tail -n 4 $O/*.log; tail -c 600 $O/bench_c5_b*.json
