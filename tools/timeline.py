"""Timeline statistics of a rocprofv3 --kernel-trace CSV: wall time covered by >= 1 kernel, average number of kernels in
flight, per-kernel stretch (duration with the sub-batch streams overlapping vs the single-stream average).
usage: python tools/timeline.py <kernel_trace.csv> [name-filter]"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r.get("Kernel_Name") or r.get("kernel_name") or r.get("Name")
        s = int(r.get("Start_Timestamp") or r.get("start_timestamp"))
        e = int(r.get("End_Timestamp") or r.get("end_timestamp"))
        rows.append((s, e, name))
rows.sort()
flt = sys.argv[2] if len(sys.argv) > 2 else "gtts::"
rows = [r for r in rows if flt in r[2]]
# restrict to the sampler region: from the first to the last gtts kernel
t0, t1 = rows[0][0], max(r[1] for r in rows)
ev = []
for s, e, _ in rows:
    ev.append((s, 1))
    ev.append((e, -1))
ev.sort()
busy = 0
area = 0
depth = 0
last = t0
hist = defaultdict(int)
for t, d in ev:
    dt = t - last
    if depth > 0:
        busy += dt
    area += depth * dt
    hist[depth] += dt
    depth += d
    last = t
wall = t1 - t0
print("kernels %d  wall %.3f ms  busy %.3f ms (%.1f %%)  mean kernels in flight while busy %.2f  sum of durations %.3f ms" %
      (len(rows), wall / 1e6, busy / 1e6, 100.0 * busy / wall, area / max(busy, 1), area / 1e6))
for k in sorted(hist):
    print("  depth %d: %.1f %%" % (k, 100.0 * hist[k] / wall))
per = defaultdict(lambda: [0, 0])
for s, e, n in rows:
    per[n][0] += 1
    per[n][1] += e - s
print("%-90s %8s %10s" % ("kernel", "launches", "avg us"))
for n, (c, tot) in sorted(per.items(), key=lambda kv: -kv[1][1])[:14]:
    print("%-90s %8d %10.1f" % (n[:90], c, tot / c / 1e3))
