#!/usr/bin/env python
"""Per-layer table of the HiFi-GAN V1 generator from a rocprofv3 --kernel-trace CSV of `bench.py --workload hifigan --steps 1
--warmup 1` (B = 16, T = 1024): the last 78 conv1d launches in program order (voc.hip:gtts_voc_forward) with their algorithmic
FLOPs -> TFLOP/s (x3 = executed, bf16x3).   python tools/hifigan_layers.py <kernel_trace.csv> [B] [T]"""
import csv
import sys

path = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
T = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        name = r.get("Kernel_Name") or r.get("Name")
        if "conv1d_mfma_kernel" in name:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
rows.sort()
layers = [("conv_pre 80->512 k7", 2.0 * 80 * 512 * 7 * T)]
ch, L = 512, T
for u, k in zip((8, 8, 2, 2), (16, 16, 4, 4)):
    layers.append(("ups %d->%d k%d s%d" % (ch, ch // 2, k, u), 2.0 * ch * (ch // 2) * k * L))
    ch //= 2
    L *= u
    for rk in (3, 7, 11):
        for d in (1, 3, 5):
            layers.append(("rb k%d c1 d%d  C=%d L=%d" % (rk, d, ch, L), 2.0 * ch * ch * rk * L))
            layers.append(("rb k%d c2 d1  C=%d L=%d" % (rk, ch, L), 2.0 * ch * ch * rk * L))
n = len(layers)
last = rows[-n:]
print("%d conv1d launches in the trace, the last %d are one forward" % (len(rows), n))
tot_t = tot_f = 0.0
print("%-34s %9s %9s %9s  %s" % ("layer", "us", "alg TF/s", "exec TF/s", "kernel"))
for (name, fl), (t0, t1, kn) in zip(layers, last):
    us = (t1 - t0) * 1e-3
    f = fl * B
    tot_t += us
    tot_f += f
    short = kn[kn.index("<"):kn.index(">") + 1] if "<" in kn else kn
    print("%-34s %9.1f %9.1f %9.1f  %s" % (name, us, f / us * 1e-6, 3 * f / us * 1e-6, short))
print("total %.1f us, %.2f TFLOP algorithmic -> %.1f TF/s (%.1f executed)" % (tot_t, tot_f * 1e-12, tot_f / tot_t * 1e-6, 3 * tot_f / tot_t * 1e-6))
