#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04_s18
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 1200 python -m pytest tests/test_gpu_prefill_kernels.py -x -q > $O/pytest_prefill_kernels.log 2>&1
timeout 2400 python -m pytest tests/test_gpu_engine.py tests/test_gpu_kv_fp16.py tests/test_gpu_fp16_weights.py tests/test_gpu_wrappers.py -x -q -m gpu > $O/pytest_engine.log 2>&1
for v in "45=1,46=2" "45=1,46=1" "45=0,46=0" "45=1,46=0"; do
n=$(echo $v | tr ',=' '__')
DD_USE_TOOLS_LIB=1 DD_TOOLS_TUNE="$v" timeout 900 python bench.py --config 5 --steps 2 --warmup 1 --no-roofline --no-cpu-baseline --single-images 0 > $O/bench_c5_$n.json 2> $O/bench_c5_$n.err
done
for v in "45=1,46=2" "45=0,46=0"; do
n=$(echo $v | tr ',=' '__')
DD_USE_TOOLS_LIB=1 DD_TOOLS_TUNE="$v" timeout 900 python bench.py --steps 3 --warmup 1 --no-roofline --no-cpu-baseline --single-images 0 > $O/bench_c3_$n.json 2> $O/bench_c3_$n.err
done
tail -n 5 $O/pytest_prefill_kernels.log | cut -c1-400; tail -n 5 $O/pytest_engine.log | cut -c1-400
for f in $O/bench_c*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); print(j['value'], j['ms_per_step'])
except Exception as e: print('ERR', e)
PY
done
