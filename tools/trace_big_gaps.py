"""GPU idle gaps in a kernel trace under /tmp/pf from the first group step on, grouped by the kernels on either side
(all streams merged)."""
import collections, csv, glob, sys
thr = float(sys.argv[1]) * 1e3 if len(sys.argv) > 1 else 50_000       # ns
rows = []
for f in glob.glob("/tmp/pf/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:44]))
rows.sort()
i0 = next(i for i, r in enumerate(rows) if r[2].startswith("k_step_begin_lanes"))
rows = rows[i0:]
end, t0 = rows[0][1], rows[0][0]
acc, cnt = collections.defaultdict(float), collections.Counter()
tot = small = 0
for i in range(1, len(rows)):
    s, e, n = rows[i]
    g = s - end
    if g > thr:
        acc[(rows[i - 1][2], n)] += g
        cnt[(rows[i - 1][2], n)] += 1
        tot += g
    elif g > 0:
        small += g
    end = max(end, e)
print(f"from the first group step: {(end - t0) / 1e9:.2f} s, {len(rows)} kernels; idle in gaps > {thr / 1e3:.0f} us: {tot / 1e9:.3f} s; in smaller gaps: {small / 1e9:.3f} s")
for k, v in sorted(acc.items(), key=lambda x: -x[1])[:22]:
    print(f"  {v / 1e6:9.2f} ms in {cnt[k]:5d} gaps   after {k[0]}   before {k[1]}")
