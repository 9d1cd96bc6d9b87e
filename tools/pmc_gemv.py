"""Summarise per-kernel SQ counters of a rocprofv3 --pmc run (mean per launch)."""
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "gemv" not in k:
            continue
        a = acc[k][row["Counter_Name"]]
        a[0] += 1
        a[1] += float(row["Counter_Value"])
for k, cs in sorted(acc.items()):
    print(k[:60], {c: round(v[1] / v[0]) for c, v in sorted(cs.items())})
