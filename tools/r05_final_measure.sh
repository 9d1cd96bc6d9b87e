#!/bin/bash
# Round 5, final measurements on the final tree: profiles (kernel trace + PMC passes, configs 3 and 5), the bench lines of configs 3 / 2 / 4 / 5.
# usage: tools/r05_final_measure.sh [profiles] [benches]
set -u
cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O
python3 -m dropoutdecoding_amd.build > $O/r05_final_build.log 2>&1 || { echo "build failed"; tail -5 $O/r05_final_build.log; exit 1; }
for what in "$@"; do
  if [ "$what" = profiles ]; then
    bash tools/collect_profiles.sh r05 stats
    bash tools/collect_profiles.sh r05 pmc
    bash tools/collect_profiles.sh r05 stats5
    bash tools/collect_profiles.sh r05 pmc5
  elif [ "$what" = benches ]; then
    timeout 900 python3 bench.py > $O/r05_bench_line.json 2> $O/r05_bench_line.err; echo "bench c3 rc=$? $(cut -c1-140 $O/r05_bench_line.json)"
    timeout 900 python3 bench.py --config 2 --no-cpu-baseline > $O/r05_bench_config2.json 2> $O/r05_bench_config2.err; echo "bench c2 rc=$? $(cut -c1-140 $O/r05_bench_config2.json)"
    timeout 900 python3 bench.py --config 4 --no-cpu-baseline > $O/r05_bench_config4.json 2> $O/r05_bench_config4.err; echo "bench c4 rc=$? $(cut -c1-140 $O/r05_bench_config4.json)"
    timeout 1200 python3 bench.py --config 5 --no-cpu-baseline > $O/r05_bench_config5.json 2> $O/r05_bench_config5.err; echo "bench c5 rc=$? $(cut -c1-140 $O/r05_bench_config5.json)"
  fi
done
