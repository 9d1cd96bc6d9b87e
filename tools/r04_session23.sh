#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04_s23
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
for c in 16 32 22 20 16 32; do
timeout 900 python bench.py --prefill-chunk $c --steps 2 --warmup 1 --no-roofline --no-cpu-baseline --single-images 0 > $O/bench_c3_chunk$c.json 2> $O/bench_c3_chunk$c.err
python - $O/bench_c3_chunk$c.json <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); print(sys.argv[1], j['value'], j['ms_per_step'])
except Exception as e: print('ERR', e)
PY
done
for c in 3 5 6; do
timeout 900 python bench.py --config 5 --prefill-chunk $c --steps 2 --warmup 1 --no-roofline --no-cpu-baseline --single-images 0 > $O/bench_c5_chunk$c.json 2> $O/bench_c5_chunk$c.err
python - $O/bench_c5_chunk$c.json <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); print(sys.argv[1], j['value'], j['ms_per_step'])
except Exception as e: print('ERR', e)
PY
done
