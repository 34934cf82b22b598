#!/usr/bin/env python3
"""Where a detector / locator training step goes: kernel time by kernel name (library event profiler) vs wall time.
    python tools/trainprof.py [--kind detector] [--batch 64]"""
import argparse, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from waveverify_amd import profile
from waveverify_amd.config import default_config
from waveverify_amd.init import random_state_dict, synthetic_clips
from waveverify_amd.train import EncoderNetTrainer

ap = argparse.ArgumentParser()
ap.add_argument("--kind", default="detector")
ap.add_argument("--batch", type=int, default=64)
a = ap.parse_args()
cfg = default_config(a.kind)
tr = EncoderNetTrainer(cfg, random_state_dict(cfg, 0, parametrized=True))
x_np, msg_np = synthetic_clips(a.batch, 16000, seed=1)
x, msg = torch.from_numpy(x_np).cuda(), torch.from_numpy(msg_np.astype(np.float32)).cuda()
mask = (torch.rand(a.batch, 1, 16000, device="cuda") < 0.8).float()
m = msg if a.kind == "detector" else None
for _ in range(2):
    tr.step(x, mask, m)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    tr.step(x, mask, m)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 3 * 1e3
profile.enable(True); profile.reset()
for _ in range(3):
    tr.step(x, mask, m)
rows = profile.collect(); profile.enable(False)
by = {}
for r in rows:
    k = by.setdefault(r["kernel"], [0.0, 0])
    k[0] += r["ms"] / 3; k[1] += r["launches"] // 3
tot = sum(v[0] for v in by.values())
print(f"{a.kind} B={a.batch}: wall {wall:.1f} ms/step, profiled kernels {tot:.1f} ms/step (training-only kernels are not instrumented)")
for k, v in sorted(by.items(), key=lambda kv: -kv[1][0])[:25]:
    print(f"  {k:34s} n={v[1]:4d} {v[0]:8.2f} ms")
