"""Per-kernel effective clock and MFMA-busy fraction from one rocprofv3 --kernel-trace --pmc pass (counter CSV + kernel trace CSV)."""
import csv, sys, collections
cc, kt = sys.argv[1], sys.argv[2]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r.get("Dispatch_Id") or r.get("Correlation_Id")] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
seen = set()
for r in csv.DictReader(open(cc)):
    did = r["Dispatch_Id"]
    name = r["Kernel_Name"]
    acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
    if did not in seen:
        seen.add(did)
        cnt[name] += 1
        if did in dur:
            acc[name]["_ns"] += dur[did][1]
rows = []
for name, c in acc.items():
    ns = c.get("_ns", 0.0)
    if ns <= 0:
        continue
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    rows.append((ns, name, cnt[name], gui / ns, c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(gui, 1) , c.get("SQ_INSTS_VALU", 0) / max(c.get("SQ_INSTS_MFMA", 0), 1), c.get("SQ_BUSY_CYCLES", 0) / max(gui, 1)))
rows.sort(reverse=True)
print("%-80s %6s %9s %8s %14s %10s" % ("kernel", "calls", "avg us", "GHz", "mfma_busy/gui", "valu:mfma"))
for ns, name, n, ghz, mb, vm, sb in rows[:16]:
    print("%-80s %6d %9.1f %8.3f %14.3f %10.2f" % (name[:80], n, ns / n / 1000.0, ghz, mb, vm))
