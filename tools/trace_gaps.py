import csv, glob, sys
rows=[]
for f in glob.glob("/tmp/pf/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
starts=[i for i,r in enumerate(rows) if "k_step_begin_lanes" in r[2]]
i0, i1 = starts[-11], starts[-1]
seg=rows[i0:i1]
wall=(seg[-1][1]-seg[0][0])
busy=sum(e-s for s,e,_ in seg)
gaps=[seg[i+1][0]-seg[i][1] for i in range(len(seg)-1)]
print(f"10 steps: wall {wall/1e6/10:.2f} ms/step, kernels {len(seg)/10:.0f}/step, busy {busy/1e6/10:.2f} ms/step, gaps {sum(g for g in gaps if g>0)/1e6/10:.2f} ms/step, mean gap {sum(gaps)/len(gaps)/1e3:.2f} us, overlapped(neg) {sum(1 for g in gaps if g<0)}")
import collections
acc=collections.defaultdict(float)
for s,e,n in seg: acc[n.split('(')[0][:60]]+=e-s
for n,v in sorted(acc.items(), key=lambda x:-x[1])[:14]: print(f"  {v/1e6/10:7.3f} ms/step  {n}")
