"""Per-kernel table of ONE steady-state training step from a rocprofv3 --kernel-trace csv of tools/train_prof.py.

The stats file of the whole process is dominated by MIOpen's solver search in step 0 (it times naive reference kernels);
a step here is everything between the last two gtts::score_loss_kernel dispatches: the backward pass of step k and the
forward pass of step k + 1.  usage: train_step_stats.py <kernel_trace.csv> [top]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "score_loss_kernel" in r["Kernel_Name"]]
if len(marks) < 2:
    sys.exit("need two training steps in the trace")
sel = rows[marks[-2]:marks[-1]]
agg = defaultdict(lambda: [0, 0.0])
for r in sel:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    a = agg[r["Kernel_Name"].replace("void ", "").split("(")[0]]
    a[0] += 1
    a[1] += d
tot = sum(v[1] for v in agg.values())
g = sum(v[1] for k, v in agg.items() if k.startswith("gtts::"))
span = int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])
print("one steady-state step: %d dispatches, %.2f ms of kernel time in a %.2f ms span; gtts:: kernels %.1f %% of kernel time"
      % (len(sel), tot / 1e6, span / 1e6, 100.0 * g / tot))
print("%7s %6s %9s  %s" % ("share", "calls", "avg us", "kernel"))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print("%6.2f%% %6d %9.1f  %s" % (100.0 * v[1] / tot, v[0], v[1] / v[0] / 1e3, k[:110]))
