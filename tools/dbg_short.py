"""f16 detector on very short / odd clips against the exact mode (robustness sweep).  python tools/dbg_short.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from waveverify_amd.config import default_config
from waveverify_amd.init import random_state_dict, synthetic_clips
from waveverify_amd.nets import HipNet
cfg = default_config("detector")
D = HipNet(cfg, random_state_dict(cfg, 0))
for B, T in [(1, 1), (2, 5), (3, 40), (1, 319), (2, 320), (2, 321), (1, 639), (7, 1000), (2, 15999), (1, 32000), (1, 48001), (300, 640)]:
    x = torch.from_numpy(synthetic_clips(B, T, seed=T)[0]).cuda()
    a = D.detector_mean_prob(x); b = D.detector_mean_prob(x, precision="f16")
    lg = D.detector(x, precision="f16")
    print(B, T, "max|dp|", float((a - b).abs().max()), "finite", bool(torch.isfinite(b).all() and torch.isfinite(lg).all()), "bits equal", bool(((a >= .5) == (b >= .5)).all()))
