#!/usr/bin/env python3
"""Run ONE K1 shape a few times (for rocprofv3 --pmc runs). python tools/kone.py C T B [pre_elu] [resid]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from waveverify_amd import ops
C, T, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
pre_elu = bool(int(sys.argv[4])) if len(sys.argv) > 4 else False
resid = bool(int(sys.argv[5])) if len(sys.argv) > 5 else True
rng = np.random.default_rng(0)
X = torch.randn(B, C, T, device="cuda")
R = torch.randn(B, C, T, device="cuda") if resid else None
w_pw = rng.standard_normal((C, C, 1)).astype(np.float32) * C ** -0.5
w_dw = rng.standard_normal((C, 1, 5)).astype(np.float32)
b = rng.standard_normal(C).astype(np.float32)
for _ in range(5):
    ops.pw_dw(X, w_pw, w_dw, b, resid=R, pre_scale=0.87 if pre_elu else 1.0, pre_elu=pre_elu, out_scale=0.5, act_scale=0.9)
torch.cuda.synchronize()
