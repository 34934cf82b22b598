import sys, torch, numpy as np
sys.path.insert(0, "/root/repo")
from waveverify_amd.config import default_config
from waveverify_amd.init import random_state_dict, synthetic_clips
from waveverify_amd.nets import HipNet
nets = {k: HipNet(default_config(k), random_state_dict(default_config(k), 0)) for k in ("generator", "detector", "locator")}
for B, T in ((3, 480000), (32, 480000), (2, 479999)):
    x_np, m_np = synthetic_clips(B, T, seed=5)
    x, m = torch.from_numpy(x_np).cuda(), torch.from_numpy(m_np).cuda()
    w32 = nets["generator"].generator(x, m, add_input=True)
    w16 = nets["generator"].generator(x, m, add_input=True, precision="f16")
    p32 = nets["detector"].detector_mean_prob(w32); p16 = nets["detector"].detector_mean_prob(w16, precision="f16")
    l32 = nets["locator"].locator(w32); l16 = nets["locator"].locator(w32, precision="f16")
    print(B, T, "wm", float((w16 - w32).abs().max()), "mp", float((p16 - p32).abs().max()), "bits differ", int(((p16 >= .5) != (p32 >= .5)).sum()),
          "loc", float((l16 - l32).abs().max()), "finite", bool(torch.isfinite(w16).all() and torch.isfinite(l16).all()), flush=True)
    del w32, w16, l32, l16
    torch.cuda.empty_cache()
