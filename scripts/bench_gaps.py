#!/usr/bin/env python3
"""Where the main queue of a bench pass is idle: gaps between consecutive kernels of the busiest queue in a rocprofv3
kernel trace (usage: bench_gaps.py <kernel_trace.csv>).  Prints the largest recurring gaps by (previous kernel -> next)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
K = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"].split("(")[0].replace("void ", "")[:28]) for r in rows)
byq = collections.defaultdict(list)
for k in K: byq[k[2]].append(k)
for q, ks in sorted(byq.items(), key=lambda kv: -len(kv[1]))[:3]:
    ks = ks[len(ks) // 3:]  # steady state
    span = (ks[-1][1] - ks[0][0]) / 1e6
    busy = sum(e - s for s, e, *_ in ks) / 1e6
    print(f"queue {q}: {len(ks)} kernels over {span:.2f} ms, busy {busy:.2f} ms ({100*busy/span:.0f}%)")
    gaps = collections.defaultdict(list)
    for a, b in zip(ks, ks[1:]):
        gaps[(a[3], b[3])].append(b[0] - a[1])
    for (a, b), g in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:14]:
        print(f"   {a:28s} -> {b:28s} n={len(g):4d} mean gap {sum(g)/len(g)/1e3:7.1f} us  total {sum(g)/1e6:7.2f} ms")
