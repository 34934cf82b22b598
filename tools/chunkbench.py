#!/usr/bin/env python3
"""Does running a bandwidth-bound ResnetBlock (two K1 launches) over chunks of clips keep the intermediate in the
256 MiB Infinity Cache?  Kernel time (event profiler) of the whole batch at once vs chunks.  python tools/chunkbench.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from waveverify_amd import ops, profile

B = 256
for C, T in ((64, 16000), (96, 16000), (128, 8000), (192, 8000)):
    rng = np.random.default_rng(0)
    X = torch.randn(B, C, T, device="cuda")
    w1 = rng.standard_normal((C, C, 1)).astype(np.float32) * C ** -0.5; w2 = w1[::-1].copy()
    d1 = rng.standard_normal((C, 1, 5)).astype(np.float32) * 0.4; d2 = d1[::-1].copy()
    b1 = rng.standard_normal(C).astype(np.float32) * 0.1; b2 = b1[::-1].copy()
    Y = torch.empty_like(X)

    def block(x):
        _, u = ops.pw_dw(x, w1, d1, b1, pre_scale=0.87, pre_elu=True, act_scale=1.0)      # raw in -> activated u
        return ops.pw_dw(u, w2, d2, b2, resid=x, pre_elu=False, out_scale=0.5)             # u, raw residual -> raw out

    fl = 2 * 2.0 * B * C * (C * T + 5 * T)
    for ch in (256, 32, 16, 8, 4):
        def run():
            for i in range(0, B, ch):
                block(X[i:i + ch])
        run(); profile.reset(); profile.enable(True)
        for _ in range(3): run()
        profile.enable(False)
        es = profile.collect()
        us = sum(e["ms"] for e in es) / 3 * 1e3
        print(f"C={C:4d} T={T:6d} chunk={ch:4d} {us:9.1f} us  {fl / us / 1e6:6.1f} TF/s  {'+'.join(sorted(set(e['kernel'] for e in es)))}", flush=True)
