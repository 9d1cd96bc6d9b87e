#!/bin/bash
# Round 5: where the prefill GEMM's cycles go — one SQ counter pass over a 2,960-row prefill (k_gemm_big<EPI>: 128 x 512 blocks, three-stage LDS ring).
# WAIT_ANY (parked on s_waitcnt / barrier) + WAIT_INST_ANY (issue stall; WAIT_INST_LDS is its LDS part) + ACTIVE_INST_ANY ~ WAVE_CYCLES (quad-cycles).
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
O=$ROOT/gpurun_out/r05_lab
mkdir -p $O
python3 -m dropoutdecoding_amd.build > $O/gemm_big_build.log 2>&1 || { echo "build failed"; exit 1; }
export TMPDIR=/tmp DD_NO_BUILD=1
cd /tmp
rm -rf /tmp/gb_sq
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/gb_sq -- python3 $ROOT/tools/prefill_time.py 2960 1,1024 > $O/gemm_big_pmc.log 2>&1
echo "rocprof rc=$?"
python3 - <<PY > $O/gemm_big_sq_counters.json
import csv, glob, json, os
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for f in glob.glob("/tmp/gb_sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        a = acc[r["Kernel_Name"]][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
dur = defaultdict(lambda: [0, 0.0])
for f in glob.glob("/tmp/gb_sq/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d = dur[r["Kernel_Name"]]
        d[0] += 1
        d[1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
out = {}
for k, cs in acc.items():
    if not any(t in k for t in ("k_gemm", "k_attn_prefill", "k_rmsnorm_split")):
        continue
    e = {c: round(v[1] / v[0], 1) for c, v in cs.items()}
    e["launches"] = max(v[0] for v in cs.values())
    if k in dur:
        e["duration_us_in_pass"] = round(dur[k][1] / dur[k][0] / 1e3, 1)
    w = e.get("SQ_WAVE_CYCLES")
    if w:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY"):
            if c in e:
                e[c + "_share_of_wave_cycles"] = round(e[c] / w, 3)
    if e.get("SQ_LDS_IDX_ACTIVE"):
        e["lds_conflict_share"] = round(e.get("SQ_LDS_BANK_CONFLICT", 0) / e["SQ_LDS_IDX_ACTIVE"], 3)
    out[k.split("(")[0]] = e
print(json.dumps(out, indent=1))
PY
python3 - <<PY
import json
d = json.load(open("$O/gemm_big_sq_counters.json"))
for k, e in d.items():
    print(k[:44], {c: e[c] for c in e if c.endswith("share_of_wave_cycles") or c in ("lds_conflict_share", "duration_us_in_pass", "launches")})
PY
