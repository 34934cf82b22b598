#!/usr/bin/env python3
"""Detector at 1024 clips x 1 s: exact f32 path vs the f16-operand mode (csrc/wv_h16.hip), with the per-kernel table of the f16 pass.
python tools/h16time.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from waveverify_amd import _lib
if "--lib" in sys.argv:                      # an A/B variant built by tools/variant.sh
    _lib.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
from waveverify_amd.config import default_config
from waveverify_amd.init import random_state_dict, synthetic_clips
from waveverify_amd.nets import HipNet
from waveverify_amd import profile
cfg = default_config("detector")
D = HipNet(cfg, random_state_dict(cfg, 0))
x = torch.from_numpy(synthetic_clips(1024, 16000, seed=1)[0]).cuda()
for prec in ("f32", "f16"):
    for _ in range(3): D.detector_mean_prob(x, precision=prec)
    torch.cuda.synchronize(); t = time.time()
    for _ in range(10): D.detector_mean_prob(x, precision=prec)
    torch.cuda.synchronize(); print(prec, (time.time() - t) / 10 * 1e3, "ms per 1024 clips")
a = D.detector_mean_prob(x, precision="f32"); b = D.detector_mean_prob(x, precision="f16")
print("max |dp|", float((a - b).abs().max()), "bits differ", int(((a >= .5) != (b >= .5)).sum()), "of", a.numel(), "min margin", float((a - .5).abs().min()))
profile.enable(True); profile.reset()
for _ in range(5): D.detector_mean_prob(x, precision="f16")
for r in sorted(profile.collect(), key=lambda r: -r["ms"])[:20]:
    print(f'{r["name"][:60]:60s} {r["launches"]:4d} {r["ms"]/5:8.3f} ms/step  {r["flops"]/r["ms"]/1e9 if r["ms"] else 0:8.1f} TF  {r["bytes"]/r["ms"]/1e6 if r["ms"] else 0:8.1f} GB/s')
