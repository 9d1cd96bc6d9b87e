#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04_s17
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
DD_REPRO_LOG=$O/probes.json timeout 1500 python tools/sampler_repro.py 10 > $O/probes.log 2>&1
timeout 1500 python -m pytest tests/test_gpu_gemv_slices.py tests/test_gpu_rider.py tests/test_gpu_full_size_configs.py -x -q -m gpu -k "fp8 or mistral" > $O/pytest_fp8.log 2>&1
for v in 1 0; do
DD_USE_TOOLS_LIB=1 DD_TOOLS_TUNE="44=$v" timeout 900 python bench.py --config 5 --steps 2 --warmup 1 --no-roofline --no-cpu-baseline --single-images 0 > $O/bench_c5_pairs$v.json 2> $O/bench_c5_pairs$v.err
done
DD_CKPT_LAYERS=32 timeout 1500 python -m pytest tests/test_gpu_checkpoint_load.py -x -q -s > $O/checkpoint_load_32_layers.log 2>&1
grep -o '"what[^}]*' $O/probes.log | cut -c1-400 | grep -i "pack\|gload" ; tail -n 4 $O/pytest_fp8.log | cut -c1-300; tail -c 300 $O/bench_c5_pairs1.json; tail -c 300 $O/bench_c5_pairs0.json; tail -n 5 $O/checkpoint_load_32_layers.log | cut -c1-400
