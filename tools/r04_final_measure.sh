#!/bin/bash
# Round-4 measurement session on the final tree: GPU suite, the bench lines of configs 3 (default) / 2 / 4 / 5, kernel-trace and PMC summaries.
# Outputs under gpurun_out/ (the caller copies what is judged into profiles/).
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04_final
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/build_smoke.log 2>&1; echo "build+smoke rc=$?"
timeout 3000 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1
tail -n 3 $O/pytest_gpu.log | cut -c1-300
timeout 1500 python bench.py > $O/r04_bench_line.json 2> $O/bench.err
for c in 2 4 5; do
timeout 1500 python bench.py --config $c --steps 2 --warmup 1 > $O/r04_bench_config$c.json 2> $O/bench_c$c.err
done
for f in $O/r04_bench_line.json $O/r04_bench_config2.json $O/r04_bench_config4.json $O/r04_bench_config5.json; do python - $f <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); print(sys.argv[1], j['value'], j['ms_per_step'], (j.get('single_stream') or {}).get('value'), (j.get('single_stream_two_sweep') or {}).get('value'))
except Exception as e: print('ERR', e)
PY
done
DD_BENCH_SHARE_DEVICE=1 timeout 900 python bench.py --gpus 2 --steps 1 --warmup 1 --images-per-gpu 16 --no-roofline --no-cpu-baseline --single-images 0 > $O/r04_bench_2_ranks_one_device.json 2> $O/bench_2ranks.err
tail -c 700 $O/r04_bench_2_ranks_one_device.json
bash tools/collect_profiles.sh r04 stats > $O/collect_stats.log 2>&1
bash tools/collect_profiles.sh r04 pmc > $O/collect_pmc.log 2>&1
bash tools/collect_profiles.sh r04 stats5 > $O/collect_stats5.log 2>&1
bash tools/collect_profiles.sh r04 pmc5 > $O/collect_pmc5.log 2>&1
timeout 600 python tools/rider_ab.py 64 "26=1" "33=0" "47=1" > $O/rider_ab_64.log 2>&1; grep -v amdgpu $O/rider_ab_64.log | cut -c1-200
timeout 600 python tools/prefill_ab.py > $O/prefill_ab.log 2>&1; grep -v amdgpu $O/prefill_ab.log | cut -c1-200
