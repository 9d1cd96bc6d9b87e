#!/bin/bash
# Round 5: counter passes over tools/fuse_lab's three variants (FETCH_SIZE, WRITE_SIZE: one counter per pass, kernel trace only — DESIGN.md 6).
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
O=$ROOT/gpurun_out/r05_lab
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for v in ${@:-A B C D E}; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/fl_${v}_${c}
    timeout 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/fl_${v}_${c} -- $ROOT/tools/fuse_lab $v > $O/fuse_lab_pmc_${v}_${c}.log 2>&1
    echo "$v $c rc=$?"
  done
  python3 $ROOT/tools/pmc_summary.py /tmp/fl_${v}_FETCH_SIZE /tmp/fl_${v}_WRITE_SIZE > $O/fuse_lab_pmc_${v}.json 2> $O/fuse_lab_pmc_${v}.err
  python3 - <<PY
import json
d=json.load(open("$O/fuse_lab_pmc_${v}.json"))["kernels"]
for k,e in d.items():
    if "fill" in k: continue
    print("$v", k[:60], "launches", e["launches"], "read MB", round(e["hbm_read_bytes_per_launch"]/1e6,1), "written MB", round(e.get("hbm_write_bytes_per_launch",0)/1e6,1))
PY
done
