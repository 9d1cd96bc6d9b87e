#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04_s5
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 400 python tools/race_bisect.py fp32 12 3 > $O/bisect_base.log 2>&1
timeout 400 python tools/race_bisect.py fp32 12 3 "39=102400" > $O/bisect_ldspad100k.log 2>&1
HIP_FORCE_DEV_KERNARG=0 timeout 400 python tools/race_bisect.py fp32 12 3 > $O/bisect_kernarg_host.log 2>&1
HIP_FORCE_DEV_KERNARG=1 timeout 400 python tools/race_bisect.py fp32 12 3 > $O/bisect_kernarg_dev.log 2>&1
DEBUG_CLR_KERNARG_HDP_FLUSH_WA=1 timeout 400 python tools/race_bisect.py fp32 12 3 > $O/bisect_hdp_flush_wa.log 2>&1
DEBUG_HIP_KERNARG_COPY_OPT=0 timeout 400 python tools/race_bisect.py fp32 12 3 > $O/bisect_kernarg_copy_opt0.log 2>&1
AMD_SERIALIZE_KERNEL=3 timeout 400 python tools/race_bisect.py fp32 12 3 > $O/bisect_serialize.log 2>&1
timeout 400 python tools/race_bisect.py fp32 12 3 "10=4" > $O/bisect_attnsplit4.log 2>&1
tail -n 5 $O/bisect*.log | cut -c1-700
