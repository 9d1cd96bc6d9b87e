#!/bin/bash
# After the round's final run: the bench line once more with the round's own profiles committed (roofline.frac cites profiles/r05_*), and the
# two-ranks-on-one-device run of bench.py (world > 1 code paths over gloo; not a scaling number).
set -u
cd "$(dirname "$0")/.."
O=gpurun_out
python3 -m dropoutdecoding_amd.build > $O/r05_post_build.log 2>&1 || { echo "build failed"; exit 1; }
timeout 900 python3 bench.py > $O/r05_bench_line.json 2> $O/r05_bench_line.err; echo "bench rc=$? $(cut -c1-120 $O/r05_bench_line.json)"
DD_BENCH_SHARE_DEVICE=1 timeout 900 python3 bench.py --gpus 2 --steps 1 --warmup 1 --images-per-gpu 16 --no-cpu-baseline --no-roofline --single-images 0 > $O/r05_bench_2_ranks_one_device.json 2> $O/r05_bench_2_ranks.err; echo "2 ranks rc=$? $(cut -c1-160 $O/r05_bench_2_ranks_one_device.json)"
