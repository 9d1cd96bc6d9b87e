#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04_s21
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 3000 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1
tail -n 3 $O/pytest_gpu.log | cut -c1-300
timeout 1500 python bench.py > $O/r04_bench_line.json 2> $O/bench.err
tail -c 600 $O/r04_bench_line.json
for c in 2 4 5; do
timeout 1200 python bench.py --config $c --steps 2 --warmup 1 > $O/r04_bench_config$c.json 2> $O/bench_c$c.err
python - $O/r04_bench_config$c.json <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); print(sys.argv[1], j['value'], j['ms_per_step'], (j.get('single_stream') or {}).get('value'), (j.get('single_stream_two_sweep') or {}).get('value'))
except Exception as e: print('ERR', e)
PY
done
for n in 48 64; do
timeout 900 python bench.py --config 5 --images-per-gpu $n --steps 2 --warmup 1 --no-roofline --no-cpu-baseline --single-images 0 > $O/bench_c5_$n.json 2> $O/bench_c5_$n.err
python - $O/bench_c5_$n.json <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); print(sys.argv[1], j['value'], j['ms_per_step'])
except Exception as e: print('ERR', e)
PY
done
bash tools/collect_profiles.sh r04 stats > $O/collect_stats.log 2>&1
bash tools/collect_profiles.sh r04 pmc > $O/collect_pmc.log 2>&1
bash tools/collect_profiles.sh r04 stats5 > $O/collect_stats5.log 2>&1
bash tools/collect_profiles.sh r04 pmc5 > $O/collect_pmc5.log 2>&1
tail -n 4 $O/collect_*.log | cut -c1-200
