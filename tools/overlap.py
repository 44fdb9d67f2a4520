"""Concurrency of a rocprofv3 kernel trace: wall span, summed kernel time, time with >= k kernels in flight."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
tot = 0
byq = collections.Counter()
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    ev.append((s, 1)); ev.append((e, -1)); tot += e - s
    byq[r.get("Queue_Id", "?")] += 1
ev.sort()
span = ev[-1][0] - ev[0][0]
cur = 0; last = ev[0][0]; hist = collections.Counter()
for t, d in ev:
    hist[cur] += t - last; last = t; cur += d
print("kernels %d  span %.2f ms  sum of kernel time %.2f ms  mean concurrency %.2f" % (len(rows), span / 1e6, tot / 1e6, tot / span))
print("queues used:", len(byq), dict(byq.most_common(12)))
for k in sorted(hist):
    print("  %2d in flight: %5.1f %%" % (k, 100.0 * hist[k] / span))
