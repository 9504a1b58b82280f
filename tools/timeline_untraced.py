#!/usr/bin/env python
"""Un-traced timeline of the sampler's sub-batch streams (the review's item 1d): HIP events recorded inside the library around
every launch ON ITS OWN STREAM (gtts_profile_enable(plan, 2)), no rocprofv3.  Prints how many kernels are in flight together
(depth histogram over the sampled region), the busy fraction and each stream's own busy time.
    python tools/timeline_untraced.py [--streams 3] [--steps 4] [--conv-ws 0]"""
import argparse
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as BN      # noqa: E402  (fixture weights)

ap = argparse.ArgumentParser()
ap.add_argument("--streams", type=int, default=3)
ap.add_argument("--steps", type=int, default=4, help="Euler steps in the sampled call")
ap.add_argument("--conv-ws", type=int, default=0)
ap.add_argument("--batch", type=int, default=16)
args = ap.parse_args()
S = importlib.import_module("speech-backbones_amd")
dev = torch.device("cuda:0")
B, T = args.batch, 1024
plan = S.Plan(streams=args.streams, conv_ws=bool(args.conv_ws))
blob = plan.pack(BN.fixture_state(1), dev)
g = torch.Generator().manual_seed(1)
mu = torch.randn(B, 80, T, generator=g).to(dev)
z = mu + torch.randn(B, 80, T, generator=g).to(dev) / 1.5
mask = torch.ones(B, 1, T, device=dev)
plan.reverse_diffusion(blob, z, mask, mu, 2)
torch.cuda.synchronize()
# plain (no events) reference time of the same call
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
plan.reverse_diffusion(blob, z, mask, mu, args.steps)
e1.record()
torch.cuda.synchronize()
plain_ms = e0.elapsed_time(e1)
plan.profile(2)
plan.reverse_diffusion(blob, z, mask, mu, args.steps)
torch.cuda.synchronize()
recs = plan.profile_timeline()
plan.profile(False)
ops = plan.ops(B, T)
names = [o[1] for o in ops] + ["prep_input", "time_mlp", "final_euler", "mul_mask", "spk_mlp"]
ev = []
for op, st, t0, t1 in recs:
    ev.append((t0, 1, st))
    ev.append((t1, -1, st))
ev.sort()
lo, hi = min(r[2] for r in recs), max(r[3] for r in recs)
depth, last = 0, lo
hist = {}
for t, d, st in ev:
    hist[depth] = hist.get(depth, 0.0) + (t - last)
    last = t
    depth += d
span = hi - lo
print("streams %d  conv_ws %d  %d launches over %d Euler steps: span %.3f ms with events (%.3f ms per U-Net call); the same call without "
      "events %.3f ms (%.3f per call)" % (args.streams, args.conv_ws, len(recs), args.steps, span, span / args.steps, plain_ms, plain_ms / args.steps))
for d in sorted(hist):
    print("  depth %d: %5.1f %%" % (d, 100 * hist[d] / span))
per = {}
for op, st, t0, t1 in recs:
    per[st] = per.get(st, 0.0) + (t1 - t0)
print("  sum of launch durations per stream:", {k: round(v, 2) for k, v in sorted(per.items())}, " total %.2f ms = %.2f x span" %
      (sum(per.values()), sum(per.values()) / span))
kern = {}
for op, st, t0, t1 in recs:
    k = kern.setdefault(names[op] if op < len(names) else str(op), [0, 0.0])
    k[0] += 1
    k[1] += t1 - t0
print("  %-64s %8s %9s" % ("kernel (as enqueued on a sub-batch stream)", "launches", "avg us"))
for name, (c, ms) in sorted(kern.items(), key=lambda kv: -kv[1][1])[:12]:
    print("  %-64s %8d %9.1f" % (name[:64], c, ms * 1e3 / c))
