#!/bin/bash
# Round 5: the 8-row decode GEMVs (the reference's own pass width, two sweeps per token) under one SQ counter pass + FETCH_SIZE: parked / issue-stalled /
# issuing wave cycles per kernel (tools/gemv_times.py rows=8 and rows=16).
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
O=$ROOT/gpurun_out/r05_lab
mkdir -p $O
python3 -m dropoutdecoding_amd.build > $O/gemv8_build.log 2>&1 || { echo "build failed"; exit 1; }
export TMPDIR=/tmp DD_NO_BUILD=1
cd /tmp
for rows in 8 16; do
  rm -rf /tmp/g8_sq_$rows
  timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d /tmp/g8_sq_$rows -- python3 $ROOT/tools/gemv_times.py rows=$rows > $O/gemv${rows}_pmc.log 2>&1
  echo "rows=$rows rocprof rc=$? $(tail -1 $O/gemv${rows}_pmc.log | cut -c1-200)"
done
python3 - <<PY > $O/gemv8_sq_counters.json
import csv, glob, json
from collections import defaultdict
out = {}
for rows in (8, 16):
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for f in glob.glob(f"/tmp/g8_sq_{rows}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            a = acc[r["Kernel_Name"]][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    dur = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(f"/tmp/g8_sq_{rows}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            d = dur[r["Kernel_Name"]]
            d[0] += 1
            d[1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    for k, cs in acc.items():
        if "k_gemv" not in k:
            continue
        n = max(v[0] for v in cs.values())
        if n < 50:
            continue
        e = {c: round(v[1] / v[0], 1) for c, v in cs.items()}
        e["launches"] = n
        if k in dur:
            e["duration_us_in_pass"] = round(dur[k][1] / dur[k][0] / 1e3, 1)
        w = e.get("SQ_WAVE_CYCLES")
        if w:
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                if c in e:
                    e[c + "_share_of_wave_cycles"] = round(e[c] / w, 3)
        out[f"rows={rows} " + k.split("(")[0]] = e
print(json.dumps(out, indent=1))
PY
python3 - <<PY
import json
d = json.load(open("$O/gemv8_sq_counters.json"))
for k, e in d.items():
    print(k[:64], {c: e[c] for c in e if c.endswith("share_of_wave_cycles") or c in ("duration_us_in_pass", "launches", "SQ_WAVES")})
PY
