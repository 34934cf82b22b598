#!/usr/bin/env python3
"""Embed + detect (+ locate) at 256 clips x 1 s (BASELINE configs[1]): exact f32 path vs the f16-operand mode (csrc/wv_h16.hip), wm / bit
agreement between the two, and the per-kernel table of the f16 pass.   python tools/g16time.py [--lib tools/bin/libwv_X.so] [B]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from waveverify_amd import _lib
args = [a for a in sys.argv[1:]]
if "--lib" in args:                      # an A/B variant built by tools/variant.sh
    i = args.index("--lib"); _lib.LIB_PATH = os.path.abspath(args[i + 1]); del args[i:i + 2]
B = int(args[0]) if args else 256
from waveverify_amd.config import default_config
from waveverify_amd.init import random_state_dict, synthetic_clips
from waveverify_amd.nets import HipNet
from waveverify_amd import profile
nets = {}
for k in ("generator", "detector", "locator"):
    cfg = default_config(k); nets[k] = HipNet(cfg, random_state_dict(cfg, 0))
G, D, L = nets["generator"], nets["detector"], nets["locator"]
x_np, m_np = synthetic_clips(B, 16000)
x, msg = torch.from_numpy(x_np).cuda(), torch.from_numpy(m_np).cuda()

def step(prec, loc=False):
    wm = G.generator(x, msg, add_input=True, precision=prec)
    mp = D.detector_mean_prob(wm, precision=prec)
    lg = L.locator(wm, precision=prec) if loc else None
    return wm, mp, lg

res = {}
for prec in ("f32", "f16"):
    for loc in (False, True):
        for _ in range(3): step(prec, loc)
        torch.cuda.synchronize(); t = time.time()
        for _ in range(10): out = step(prec, loc)
        torch.cuda.synchronize(); ms = (time.time() - t) / 10 * 1e3
        print(f"{prec} embed+detect{'+locate' if loc else ''}: {ms:.2f} ms per {B} clips = {B / ms * 1e3:.0f} clips/s", flush=True)
    res[prec] = out
(w32, p32, l32), (w16, p16, l16) = res["f32"], res["f16"]
print("wm max|d| f16 vs f32:", float((w16 - w32).abs().max()), " mean-prob max|d|:", float((p16 - p32).abs().max()),
      " bits differ:", int(((p16 >= .5) != (p32 >= .5)).sum()), "of", p32.numel(), " bits vs message differ (f32 | f16):",
      int(((p32 >= .5).float() != msg).sum()), int(((p16 >= .5).float() != msg).sum()))
print("locator logits max|d|:", float((l16 - l32).abs().max()), "of |max|", float(l32.abs().max()), " decisions (>0.5) differ:",
      int(((l16 > .5) != (l32 > .5)).sum()), "of", l32.numel())
profile.enable(True); profile.reset()
for _ in range(5): step("f16", True)
tot = 0.0
rows = sorted(profile.collect(), key=lambda r: -r["ms"])
for r in rows: tot += r["ms"] / 5
print(f"f16 pass, kernel time {tot:.2f} ms per step")
for r in rows[:40]:
    print(f'{r["name"][:52]:52s} {r["launches"]//5:4d} {r["ms"]/5:8.3f} ms/step  {r["flops"]/r["ms"]/1e9 if r["ms"] else 0:8.1f} TF  {r["bytes"]/r["ms"]/1e6 if r["ms"] else 0:8.1f} GB/s')
