#!/usr/bin/env python3
"""Where the streams of a path wait: python tools/stream_gaps.py <kernel_trace.csv of rocprofv3 --kernel-trace> [min ms of a listed kernel]
Per hardware queue: the busy fraction over the trace's span; per stream of the long kernels: each long kernel with the idle time of its
stream before it and the long kernels of OTHER streams that were running on the same hardware queue during that idle time."""
import csv, collections, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], int(r["Queue_Id"]), int(r["Stream_Id"])))
rows.sort()
minms = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
long_ = [r for r in rows if r[1] - r[0] > minms * 1e6]
t0, t1 = long_[0][0], max(r[1] for r in long_)
print("span of the long kernels: %.1f ms" % ((t1 - t0) / 1e6))
byq = collections.defaultdict(list)
for s, e, n, q, st in rows:
    if e > t0 and s < t1:
        byq[q].append((s, e, n, st))
for q, v in sorted(byq.items()):
    busy = 0; cur = None
    for s, e, n, st in sorted(v):
        if cur is None or s > cur[1]:
            if cur: busy += cur[1] - cur[0]
            cur = [s, e]
        else:
            cur[1] = max(cur[1], e)
    busy += cur[1] - cur[0]
    print("queue %2d: busy %5.1f %% of the span, %5d dispatches, streams %s" % (q, 100.0 * busy / (t1 - t0), len(v), sorted(set(st for _, _, _, st in v))))
bys = collections.defaultdict(list)
for s, e, n, q, st in rows:
    bys[st].append((s, e, n, q))
for st, v in sorted(bys.items()):
    if not any(e - s > minms * 1e6 for s, e, n, q in v):
        continue
    print("stream %d" % st)
    prev_end = None
    for s, e, n, q in v:
        if prev_end is not None and (e - s > minms * 1e6 or s - prev_end > 1e6):
            others = [(n2, st2) for s2, e2, n2, q2, st2 in long_ if q2 == q and st2 != st and s2 < s and e2 > prev_end]
            print("   %-16s queue %d  start %8.1f  dur %7.1f ms  idle before %6.1f ms%s" % (n[:16], q, (s - t0) / 1e6, (e - s) / 1e6, (s - prev_end) / 1e6,
                  ("   same queue meanwhile: " + ", ".join("%s(stream %d)" % o for o in others)) if others and s - prev_end > 1e6 else ""))
        prev_end = e if prev_end is None else max(prev_end, e)
