#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04_s3
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
for n in 12 16 10; do
  timeout 300 python tools/lanes_mixed_ab.py fp32 $n "37=1" > $O/fp32_fork_$n.log 2>&1
done
DD_AB_STEPS=8 timeout 300 python tools/lanes_mixed_ab.py fp32 12 "37=1" > $O/fp32_fork_12_8steps.log 2>&1
timeout 1800 python -m pytest tests/test_gpu_checkpoint_load.py tests/test_gpu_dist_nccl.py tests/test_gpu_sampler_repro.py -x -q -m gpu -s > $O/pytest_new.log 2>&1
# config 5 kernel trace (16 images per step: one rider ring)
rm -rf /tmp/c5_stats
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c5_stats -- python3 bench.py --config 5 --images-per-gpu 16 --steps 1 --warmup 0 --n-new 24 --no-cpu-baseline --no-roofline --single-images 0 > $O/c5_under_rocprof.log 2>&1
f=$(find /tmp/c5_stats -name "*kernel_stats.csv" | head -1)
head -60 "$f" > $O/r04_c5_kernel_stats.csv
rm -rf /tmp/c5_stats
# headline A/B knobs
timeout 900 python tools/rider_ab.py 64 "26=1" "18=-2" "36=1" "18=-2,36=1" > $O/rider_ab_64.log 2>&1
tail -n 6 $O/*.log; head -30 $O/r04_c5_kernel_stats.csv | cut -c1-170
