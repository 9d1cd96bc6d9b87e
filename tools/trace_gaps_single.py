"""Busy time, gaps and kernels per step of the one-sequence step from a rocprofv3 kernel trace under /tmp/pf
(rocprofv3 --kernel-trace --output-format csv -d /tmp/pf -- python3 tools/single_step_split.py)."""
import collections, csv, glob
rows = []
for f in glob.glob("/tmp/pf/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2].startswith("k_step_begin(")]
for name, (a, b) in (("speculative, host-decided (generate)", (150, 250)), ("two sweeps (generate)", (len(starts) - 110, len(starts) - 10))):
    i0, i1 = starts[a], starts[b]
    seg = rows[i0:i1]
    n = b - a
    wall = seg[-1][1] - seg[0][0]
    busy = sum(e - s for s, e, _ in seg)
    gaps = [seg[i + 1][0] - seg[i][1] for i in range(len(seg) - 1)]
    big = sorted(gaps)[-n:]
    print(f"{name}: {n} steps: wall {wall / 1e6 / n:.3f} ms/step, kernels {len(seg) / n:.0f}/step, busy {busy / 1e6 / n:.3f}, gaps {sum(g for g in gaps if g > 0) / 1e6 / n:.3f} "
          f"(largest gap per step ~{sum(big) / len(big) / 1e3:.1f} us)")
    acc = collections.defaultdict(float)
    cnt = collections.Counter()
    for s, e, k in seg:
        acc[k.split("(")[0][:70]] += e - s
        cnt[k.split("(")[0][:70]] += 1
    for k, v in sorted(acc.items(), key=lambda x: -x[1])[:16]:
        print(f"   {v / 1e3 / n:8.1f} us/step {cnt[k] / n:6.1f}x  {k}")
# host-decided step: the two places the GPU waits for the host
chk = [i for i, r in enumerate(rows) if r[2].startswith("k_spec_check(")]
g1, g2 = [], []
for i in chk[150:250]:
    g1.append(rows[i + 1][0] - rows[i][1])                 # check -> first kernel the host launched after reading the note
for i in starts[150:250]:
    g2.append(rows[i][0] - rows[i - 1][1])                 # last kernel of the previous step -> k_step_begin
import statistics
print(f"after the check: median {statistics.median(g1) / 1e3:.1f} us; before k_step_begin: median {statistics.median(g2) / 1e3:.1f} us")
