#!/bin/bash
# Round 6, the final sequence on the final tree (one gpurun call): GPU suite, smoke, the full-depth oracle run, profiles (kernel trace + PMC passes,
# configs 3 and 5), the bench lines of configs 3 / 2 / 4 / 5 and two ranks on one device, then the repetition stress of the overlapped pipeline.
# usage: tools/r06_final.sh [suite] [oracle] [profiles] [benches] [stress]
set -u
cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O
python3 -m dropoutdecoding_amd.build > $O/r06_final_build.log 2>&1 || { echo "build failed"; tail -5 $O/r06_final_build.log; exit 1; }
for what in "$@"; do
  if [ "$what" = suite ]; then
    timeout 1800 python3 -m pytest tests -m gpu -q --durations=8 > $O/r06_pytest_gpu_final.log 2>&1; rc=$?; echo "pytest rc=$rc $(tail -3 $O/r06_pytest_gpu_final.log | tr '\n' ' ')"
    grep -n "FAILED" $O/r06_pytest_gpu_final.log | tail -12
    python3 -c "import __graft_entry__ as g; g.smoke()" > $O/r06_smoke.log 2>&1; echo "smoke rc=$? $(tail -2 $O/r06_smoke.log | tr '\n' ' ' | cut -c1-200)"
  elif [ "$what" = oracle ]; then
    DD_FULL_DEPTH=1 timeout 1500 python3 -m pytest tests/test_gpu_full_depth_oracle.py -x -q -s > $O/r06_full_size_oracle.log 2>&1; echo "full-depth oracle rc=$? $(tail -1 $O/r06_full_size_oracle.log)"
  elif [ "$what" = profiles ]; then
    bash tools/collect_profiles.sh r06 stats
    bash tools/collect_profiles.sh r06 pmc
    bash tools/collect_profiles.sh r06 stats5
    bash tools/collect_profiles.sh r06 pmc5
    # the bench lines below read the newest profile under profiles/ (roofline.kernel_stats_file, traffic, profile_matches_tree): on this box
    # that is what was just collected on this tree (the same files are committed from gpurun_out/ afterwards)
    for f in r06_kernel_stats.csv r06_pmc_summary.json r06_sources.json r06_c5_kernel_stats.csv r06_c5_pmc_summary.json; do
      [ -s $O/$f ] && cp $O/$f profiles/$f
    done
  elif [ "$what" = benches ]; then
    timeout 900 python3 bench.py > $O/r06_bench_line.json 2> $O/r06_bench_line.err; echo "bench c3 rc=$? $(cut -c1-140 $O/r06_bench_line.json)"
    timeout 900 python3 bench.py --config 2 --no-cpu-baseline > $O/r06_bench_config2.json 2> $O/r06_bench_config2.err; echo "bench c2 rc=$? $(cut -c1-140 $O/r06_bench_config2.json)"
    timeout 900 python3 bench.py --config 4 --no-cpu-baseline > $O/r06_bench_config4.json 2> $O/r06_bench_config4.err; echo "bench c4 rc=$? $(cut -c1-140 $O/r06_bench_config4.json)"
    timeout 1200 python3 bench.py --config 5 --no-cpu-baseline > $O/r06_bench_config5.json 2> $O/r06_bench_config5.err; echo "bench c5 rc=$? $(cut -c1-140 $O/r06_bench_config5.json)"
    DD_BENCH_SHARE_DEVICE=1 timeout 900 python3 bench.py --gpus 2 --images-per-gpu 24 --no-cpu-baseline --no-roofline > $O/r06_bench_2_ranks_one_device.json 2> $O/r06_bench_2_ranks.err; echo "bench 2 ranks rc=$? $(cut -c1-140 $O/r06_bench_2_ranks_one_device.json)"
  elif [ "$what" = stress ]; then
    # GroupPipeline as bench.py drives it (configs 3 and 5), R repetitions of the same batches from the same seeds: exit code 1 on a difference
    timeout 1500 python3 tools/stress_pipeline.py 3 ${R06_STRESS_REPS:-10} > $O/r06_stress_c3.log 2>&1; echo "stress c3 rc=$? $(tail -1 $O/r06_stress_c3.log | cut -c1-200)"
    timeout 1500 python3 tools/stress_pipeline.py 5 ${R06_STRESS_REPS5:-4} > $O/r06_stress_c5.log 2>&1; echo "stress c5 rc=$? $(tail -1 $O/r06_stress_c5.log | cut -c1-200)"
  fi
done
