#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04_s20
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
for w in mistral llava; do
rm -rf /tmp/pab_$w
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pab_$w -- python3 tools/prefill_ab.py $w > $O/prefill_ab_$w.log 2>&1
f=$(find /tmp/pab_$w -name "*kernel_stats.csv" | head -1)
python3 - "$f" > $O/prefill_ab_${w}_stats.csv <<'PY'
import csv, sys, re
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    print(re.sub(r"\(.*", "", r["Name"])[:60], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), r["Percentage"], sep=",")
PY
done
cat $O/prefill_ab_mistral_stats.csv $O/prefill_ab_llava_stats.csv
