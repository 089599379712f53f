#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace CSV of a bench run with batches in flight and reports, per kernel, how much of its
run time other streams' kernels were running beside it, and the distribution of the number of kernels in flight.
    python tools/overlap_trace.py <kernel_trace.csv> [lo hi]
lo, hi: the part of the trace to look at, as fractions of its kernel launches (default 0.05 0.45: bench.py's timed in-flight loop
comes first, its single-context comparison and event-timed steps after it)."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if not n.startswith("k_fast_cells_cols"):
        n = n.split("<")[0]
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, r.get("Queue_Id", r.get("Stream_Id", "0"))))
ev.sort()
lo, hi = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (0.05, 0.45)
t0, t1 = ev[int(lo * len(ev))][0], ev[int(hi * len(ev))][0]
ev = [e for e in ev if t0 <= e[0] < t1]
pts = []
for s, e, n, q in ev:
    pts.append((s, 1)); pts.append((e, -1))
pts.sort()
depth_time = collections.Counter()
cur, last = 0, pts[0][0]
for t, d in pts:
    depth_time[cur] += t - last
    cur += d; last = t
tot = sum(depth_time.values())
print("kernels in flight -> share of wall time:", {k: round(v / tot, 3) for k, v in sorted(depth_time.items())})
# per kernel: own duration and the average number of OTHER kernels running during it
import bisect
starts = sorted((s, e) for s, e, _, _ in ev)
acc = collections.defaultdict(lambda: [0, 0.0, 0.0])
for s, e, n, q in ev:
    ov = 0
    for s2, e2 in starts:
        if s2 >= e:
            break
        if e2 > s and (s2, e2) != (s, e):
            ov += min(e, e2) - max(s, s2)
    a = acc[n]; a[0] += 1; a[1] += e - s; a[2] += ov
print(f"{'kernel':28s} {'calls':>6s} {'avg us':>9s} {'others running beside it (avg count)':>40s}")
for n, (c, d, ov) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"{n:28s} {c:6d} {d / c / 1e3:9.1f} {ov / d:40.2f}")
span = ev[-1][1] - ev[0][0]
print(f"span {span / 1e6:.2f} ms, sum of kernel time {sum(e - s for s, e, _, _ in ev) / 1e6:.2f} ms -> average concurrency {sum(e - s for s, e, _, _ in ev) / span:.2f}")
