#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04_s22
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 1200 python -m pytest tests/test_gpu_prefill_kernels.py tests/test_gpu_engine.py -x -q -k "prefill or fp8" > $O/pytest_prefill.log 2>&1
tail -n 3 $O/pytest_prefill.log | cut -c1-300
for c in 2 4 8; do
timeout 900 python bench.py --config 5 --prefill-chunk $c --steps 2 --warmup 1 --no-roofline --no-cpu-baseline --single-images 0 > $O/bench_c5_chunk$c.json 2> $O/bench_c5_chunk$c.err
python - $O/bench_c5_chunk$c.json <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); print(sys.argv[1], j['value'], j['ms_per_step'])
except Exception as e: print('ERR', e)
PY
done
