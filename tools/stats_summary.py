"""Summarise a rocprofv3 kernel_stats csv: top kernels, gtts:: share.  usage: stats_summary.py file.csv [top]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
tot = sum(float(r["TotalDurationNs"]) for r in rows)
g = sum(float(r["TotalDurationNs"]) for r in rows if "gtts::" in r["Name"])
print("total GPU time %.2f ms over %d kernels names; gtts:: share %.1f %%" % (tot / 1e6, len(rows), 100 * g / tot))
for r in rows[:top]:
    print("%6.2f%% %5d calls %8.1f us avg  %s" % (100 * float(r["TotalDurationNs"]) / tot, int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"][:120]))
