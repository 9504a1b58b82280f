"""Aggregate a rocprofv3 counter_collection.csv per kernel name: sum of each counter and launches."""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
agg = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(set)
with open(path) as f:
    for row in csv.DictReader(f):
        k = row.get("Kernel_Name", "?")
        k = k.replace("void ", "").split("(")[0]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[k].add(row.get("Dispatch_Id"))
names = sorted({c for v in agg.values() for c in v})
print("%-72s %8s " % ("kernel", "launches") + " ".join("%22s" % n for n in names))
key = names[0] if names else None
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get(key, 0)):
    print("%-72s %8d " % (k, len(cnt[k])) + " ".join("%22.4g" % v.get(n, 0) for n in names))
