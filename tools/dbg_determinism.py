"""Run-to-run determinism of the forward paths (races show up as differing bits): 20 runs each of the f16 detector (mean-prob and logits
outputs), the exact detector and the generator on one batch.  python tools/dbg_determinism.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from waveverify_amd.config import default_config
from waveverify_amd.init import random_state_dict, synthetic_clips
from waveverify_amd.nets import HipNet
cfgD, cfgG = default_config("detector"), default_config("generator")
D, G = HipNet(cfgD, random_state_dict(cfgD, 0)), HipNet(cfgG, random_state_dict(cfgG, 0))
x_np, msg_np = synthetic_clips(96, 16000, seed=5)
x, msg = torch.from_numpy(x_np).cuda(), torch.from_numpy(msg_np).cuda()
ref = dict(mp16=D.detector_mean_prob(x, precision="f16"), lg16=D.detector(x[:8], precision="f16"), mp32=D.detector_mean_prob(x), wm=G.generator(x, msg, add_input=True))
bad = {k: 0 for k in ref}
for it in range(20):
    cur = dict(mp16=D.detector_mean_prob(x, precision="f16"), lg16=D.detector(x[:8], precision="f16"), mp32=D.detector_mean_prob(x), wm=G.generator(x, msg, add_input=True))
    for k in ref:
        bad[k] += int(not torch.equal(cur[k], ref[k]))
print("runs differing from the first:", bad)
