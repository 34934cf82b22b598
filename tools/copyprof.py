#!/usr/bin/env python3
"""Which Python lines of a WatermarkTrainer step issue device copies / fills (torch ops that become __amd_rocclr_copyBuffer /
fillBuffer launches)?  Counts aten::copy_ / fill_ / zero_ per calling source line over one step.
    python tools/copyprof.py [--batch 8]"""
import argparse, collections, os, sys, traceback
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from waveverify_amd.config import default_config
from waveverify_amd.init import random_state_dict, synthetic_clips
from waveverify_amd.train import WatermarkTrainer
from torch.utils._python_dispatch import TorchDispatchMode

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
a = ap.parse_args()
x_np, msg_np = synthetic_clips(a.batch, 16000, seed=1)
x, msg = torch.from_numpy(x_np).cuda(), torch.from_numpy(msg_np.astype(np.float32)).cuda()
cfgs = [default_config(k) for k in ("generator", "detector", "locator")]
sds = [random_state_dict(c, 0, parametrized=True) for c in cfgs]
tr = WatermarkTrainer(cfgs[0], sds[0], cfgs[1], sds[1], cfgs[2], sds[2], device="cuda")
np.random.seed(1); torch.manual_seed(1)
for _ in range(2):
    tr.step(x, msg)
counts = collections.Counter()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Count(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__ if hasattr(func, "__name__") else str(func)
        on_gpu = any(isinstance(t, torch.Tensor) and t.is_cuda for t in list(args) + list((kwargs or {}).values()))
        out = func(*args, **(kwargs or {}))
        if isinstance(out, torch.Tensor) and out.is_cuda:
            on_gpu = True
        if on_gpu and not any(s in name for s in ("view", "reshape", "as_strided", "detach", "alias", "slice", "select", "empty", "unsqueeze", "squeeze", "expand", "t.default", "transpose", "permute", "_unsafe_view", "is_")):
            for fr in reversed(traceback.extract_stack()):
                if fr.filename.startswith(ROOT) and "copyprof" not in fr.filename:
                    counts[(name, os.path.relpath(fr.filename, ROOT), fr.lineno, fr.line.strip()[:90])] += 1
                    break
        return out


with Count():
    tr.step(x, msg)
torch.cuda.synchronize()
tot = sum(counts.values())
print("device torch ops in one step:", tot)
for (name, f, ln, src), n in counts.most_common(45):
    print(f"{n:4d}  {name:28s} {f}:{ln}  {src}")
