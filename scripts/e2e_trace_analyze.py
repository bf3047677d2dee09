#!/usr/bin/env python3
"""Timeline summary of a rocprofv3 kernel trace of the streaming engine: per run (split at the longest gaps) and per
hardware queue: kernel count, busy time, median kernel duration; plus the copies per direction."""
import csv, sys, collections, statistics
rows = list(csv.DictReader(open(sys.argv[1])))
K = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r.get("Stream_Id", "?"), r["Kernel_Name"].split("(")[0][:36]) for r in rows)
# split into runs at gaps > 0.3 s
runs, cur = [], [K[0]]
for a, b in zip(K, K[1:]):
    if b[0] - a[1] > 3e8: runs.append(cur); cur = []
    cur.append(b)
runs.append(cur)
for i, R in enumerate(runs):
    span = (R[-1][1] - R[0][0]) / 1e9
    if span < 0.5: continue
    print(f"== run {i}: {span:.2f} s, {len(R)} kernels")
    byq = collections.defaultdict(list)
    for k in R: byq[(k[2], k[3])].append(k)
    for q, ks in sorted(byq.items(), key=lambda kv: -len(kv[1])):
        d = sorted(e - s for s, e, *_ in ks)
        names = [n for n, _ in collections.Counter(k[4] for k in ks).most_common(3)]
        print(f"   queue {q[0]} stream {q[1]}: {len(ks):6d} kernels, busy {sum(d)/1e9:.3f} s, median {d[len(d)//2]/1e3:.1f} us, p90 {d[int(len(d)*.9)]/1e3:.1f} us  {names}")
if len(sys.argv) > 2 and sys.argv[2]:
    M = list(csv.DictReader(open(sys.argv[2])))
    tot = collections.defaultdict(lambda: [0, 0])
    for r in M:
        tot[(r.get("Direction", "?"), r.get("Stream_Id", "?"))][0] += 1
        tot[(r.get("Direction", "?"), r.get("Stream_Id", "?"))][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for k, (n, d) in sorted(tot.items()):
        print(f"copies {k}: {n} calls, busy {d/1e9:.3f} s, mean {d/max(n,1)/1e3:.1f} us")
