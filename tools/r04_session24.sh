#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04_s24
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 1800 python -m pytest tests/test_gpu_prefill_kernels.py tests/test_gpu_vision.py tests/test_gpu_fp32_branches.py tests/test_gpu_wrappers.py -x -q > $O/pytest.log 2>&1
tail -n 5 $O/pytest.log | cut -c1-400
for c in 3 5; do
timeout 900 python bench.py --config $c --steps 2 --warmup 1 --no-roofline --no-cpu-baseline --single-images 0 > $O/bench_c$c.json 2> $O/bench_c$c.err
python - $O/bench_c$c.json <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); print(sys.argv[1], j['value'], j['ms_per_step'])
except Exception as e: print('ERR', e)
PY
done
timeout 600 python tools/tower_time.py > $O/tower_time.log 2>&1; grep -v amdgpu $O/tower_time.log | tail -n 6 | cut -c1-200
