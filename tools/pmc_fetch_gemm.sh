export TMPDIR=/tmp
for v in "2960 0,0" "2960 1,0" "2960 0,1024" "2960 1,1024" "608 0,0" "608 1,0"; do
  set -- $v
  rm -rf /tmp/pm
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pm -- python3 tools/prefill_time.py $1 $2 > /tmp/pm.log 2>&1
  python3 - "$v" <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(lambda: [0, 0.0])
for f in glob.glob("/tmp/pm/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if row.get("Counter_Name") == "FETCH_SIZE" and "k_gemm" in row["Kernel_Name"]:
            a = acc[row["Kernel_Name"][:32]]; a[0] += 1; a[1] += float(row["Counter_Value"])
print(sys.argv[1], {k: f"{2 * v[1] / v[0] * 1024 / 1e6:.0f} MB" for k, v in sorted(acc.items())})
PY
done
