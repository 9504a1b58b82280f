"""Aggregate a rocprofv3 counter_collection.csv per kernel name: sum of each counter and launches."""
import csv
import sys
from collections import defaultdict

def demangle(name):
    """rocprofv3 leaves the bf16 template instances mangled (its demangler does not know DF16b): restore the printed form of
    the gtts:: kernels (int, float and __bf16 template arguments are all this library uses)."""
    import re
    m = re.match(r"_ZN4gtts(\d+)", name)
    if not m:
        return name
    n = int(m.group(1))
    base, rest = name[m.end():m.end() + n], name[m.end() + n:]
    if not rest.startswith("I"):
        return "gtts::" + base
    args, i = [], 1
    while i < len(rest) and rest[i] != "E":
        if rest.startswith("Li", i):
            j = rest.index("E", i)
            v = rest[i + 2:j]
            args.append("-" + v[1:] if v.startswith("n") else v)
            i = j + 1
        elif rest.startswith("DF16b", i):
            args.append("__bf16")
            i += 5
        elif rest[i] == "f":
            args.append("float")
            i += 1
        else:
            return name
    return "gtts::%s<%s>" % (base, ", ".join(args))


path = sys.argv[1]
agg = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(set)
with open(path) as f:
    for row in csv.DictReader(f):
        k = row.get("Kernel_Name", "?")
        k = demangle(k.replace("void ", "").split("(")[0])
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[k].add(row.get("Dispatch_Id"))
names = sorted({c for v in agg.values() for c in v})
print("%-72s %8s " % ("kernel", "launches") + " ".join("%22s" % n for n in names))
key = names[0] if names else None
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get(key, 0)):
    print("%-72s %8d " % (k, len(cnt[k])) + " ".join("%22.4g" % v.get(n, 0) for n in names))
