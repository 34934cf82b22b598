#!/usr/bin/env python3
"""Run ONE K1 shape a few times (for rocprofv3 --pmc runs). python tools/kone.py C T B prec"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from waveverify_amd import ops
C, T, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
prec = sys.argv[4] if len(sys.argv) > 4 else "f32"
rng = np.random.default_rng(0)
X = torch.randn(B, C, T, device="cuda")
R = torch.randn(B, C, T, device="cuda")
w_pw = rng.standard_normal((C, C, 1)).astype(np.float32) * C ** -0.5
w_dw = rng.standard_normal((C, 1, 5)).astype(np.float32)
b = rng.standard_normal(C).astype(np.float32)
ops.set_precision(prec)
for _ in range(3):
    ops.pw_dw(X, w_pw, w_dw, b, resid=R, pre_scale=0.87, pre_elu=True, out_scale=0.5)
torch.cuda.synchronize()
