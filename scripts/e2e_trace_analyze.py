#!/usr/bin/env python3
"""Timeline summary of a rocprofv3 kernel trace of the streaming engine: per queue busy time, duration inflation and
the gaps between consecutive kernels of the busiest queue (the COMPUTE lane)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
print("columns:", list(rows[0].keys()))
K = []
for r in rows:
    K.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", r.get("Stream_Id", "?")), r["Kernel_Name"].split("(")[0][:40]))
K.sort()
t0, t1 = K[0][0], max(k[1] for k in K)
# steady-state window: the last 2 s
lo = t1 - int(2.0e9)
K = [k for k in K if k[0] >= lo]
span = (t1 - lo) / 1e9
print(f"window {span:.2f} s, kernels {len(K)}")
byq = collections.defaultdict(list)
for k in K: byq[k[2]].append(k)
for q, ks in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    busy = sum(e - s for s, e, _, _ in ks) / 1e9
    names = collections.Counter(n for _, _, _, n in ks).most_common(4)
    print(f"queue {q}: {len(ks)} kernels, sum of durations {busy:.3f} s ({100*busy/span:.0f}% of window); {names}")
# union busy over all queues
ev = sorted([(s, 1) for s, e, _, _ in K] + [(e, -1) for s, e, _, _ in K])
act = 0; last = lo; idle = 0
for t, d in ev:
    if act == 0: idle += t - last
    act += d; last = t
print(f"GPU idle (no kernel of ours running) {idle/1e9:.3f} s = {100*idle/1e9/span:.0f}% of window")
q = max(byq, key=lambda q: len(byq[q]))
ks = byq[q]
gaps = [ks[i + 1][0] - ks[i][1] for i in range(len(ks) - 1)]
import statistics
print(f"busiest queue {q}: median gap {statistics.median(gaps)/1e3:.1f} us, mean gap {statistics.mean(gaps)/1e3:.1f} us, sum of gaps {sum(g for g in gaps if g>0)/1e9:.3f} s")
per = collections.defaultdict(list)
for s, e, _, n in ks: per[n].append(e - s)
print("kernel                                     calls   median us   mean us   p90 us   total ms")
for n, d in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    d.sort()
    print(f"{n:42s} {len(d):6d} {d[len(d)//2]/1e3:10.1f} {sum(d)/len(d)/1e3:9.1f} {d[int(len(d)*0.9)]/1e3:8.1f} {sum(d)/1e6:10.1f}")
if len(sys.argv) > 2 and sys.argv[2]:
    M = list(csv.DictReader(open(sys.argv[2])))
    if M:
        print("copy columns:", list(M[0].keys()))
        tot = collections.defaultdict(lambda: [0, 0, 0])
        for r in M:
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            if s < lo: continue
            k = r.get("Direction", "?")
            tot[k][0] += 1; tot[k][1] += e - s; tot[k][2] += int(r.get("Size", r.get("Bytes", 0)) or 0)
        for k, (n, d, b) in tot.items():
            print(f"copies {k}: {n} calls, busy {d/1e9:.3f} s ({100*d/1e9/span:.0f}%), {b/1e9:.2f} GB -> {b/max(d,1):.1f} GB/s while active, {b/1e9/span:.1f} GB/s over the window")
