"""Per-kernel totals and the start/end of the long kernels from a rocprofv3 --kernel-trace database (rocpd format):
    python tools/kernel_timeline.py gpurun_out/ffprof/ff_results.db"""
import collections, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end, stream_id, queue_id from kernels order by start"))
t0 = rows[0][1]
agg = collections.defaultdict(list)
for n, s, e, st, q in rows:
    agg[n.split('(')[0]].append((s - t0, e - t0, st, q))
for n, v in sorted(agg.items(), key=lambda kv: -sum(e - s for s, e, _, _ in kv[1]))[:16]:
    tot = sum(e - s for s, e, _, _ in v) / 1e6
    print("%-40s n=%6d total %9.1f ms  avg %8.3f ms  max %8.2f ms" % (n[:40], len(v), tot, tot / len(v), max(e - s for s, e, _, _ in v) / 1e6))
long_ = sorted(((s, e, n, st, q) for n, v in agg.items() for s, e, st, q in v if e - s > 2e6))
for s, e, n, st, q in long_[-int(sys.argv[2]) if len(sys.argv) > 2 else -14:]:
    print("%-16s start %8.1f end %8.1f  (%6.1f ms)  stream %s queue %s" % (n[:16], s / 1e6, e / 1e6, (e - s) / 1e6, st, q))
