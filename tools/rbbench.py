#!/usr/bin/env python3
"""One-launch ResnetBlock kernel (wv_rb.hip, raw in / raw out) vs the two-launch form on the narrow layer shapes.
python tools/rbbench.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from waveverify_amd import _lib
if "--lib" in sys.argv:                      # an A/B variant built by tools/variant.sh
    _lib.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
from waveverify_amd import ops, profile

def t_of(f, reps=5):
    f(); profile.reset(); profile.enable(True)
    for _ in range(reps): f()
    profile.enable(False)
    es = profile.collect()
    return sum(e["ms"] for e in es) / reps * 1e3, "+".join(sorted(set(e["kernel"] for e in es)))

B = 256
for C, T in ((64, 16000), (96, 16000), (128, 8000), (192, 8000)):
    rng = np.random.default_rng(0)
    X = torch.randn(B, C, T, device="cuda")
    w1 = rng.standard_normal((C, C, 1)).astype(np.float32) * C ** -0.5; w2 = w1[::-1].copy()
    d1 = rng.standard_normal((C, 1, 5)).astype(np.float32) * 0.4; d2 = d1[::-1].copy()
    b1 = rng.standard_normal(C).astype(np.float32) * 0.1; b2 = b1[::-1].copy()
    def fused(): ops.resblock(X, w1, d1, b1, w2, d2, b2, pre_scale=0.87, out_scale=0.5)
    def two():
        _, u = ops.pw_dw(X, w1, d1, b1, pre_scale=0.87, pre_elu=True, act_scale=1.0)
        ops.pw_dw(u, w2, d2, b2, resid=X, pre_elu=False, out_scale=0.5)
    if "--stamps" in sys.argv:                   # a -DRB_STAMP variant: per-phase shader-clock totals of every wave at the head of Y
        Y = ops.resblock(X, w1, d1, b1, w2, d2, b2, pre_scale=0.87, out_scale=0.5)
        torch.cuda.synchronize()
        nw = {64: 512 * 4, 96: 256 * 12, 128: 256 * 8, 192: 256 * 12}[C]
        ph = Y.flatten()[:nw * 9].reshape(nw, 9).double().cpu().numpy()
        names = ["act", "B1", "gemm1", "B2", "epi1", "B3", "gemm2", "B4", "epi2"]
        tot = ph.sum(1).mean()
        print(f"C={C}: cycles per wave {tot:.0f}: " + "  ".join(f"{n} {100 * ph[:, i].mean() / tot:.1f}%" for i, n in enumerate(names)), flush=True)
        continue
    fl = 2 * 2.0 * B * C * (C * T + 5 * T)
    for nm, f in ((("fused", fused),) if "--fused-only" in sys.argv else (("fused", fused), ("two launches", two))):
        us, k = t_of(f)
        print(f"C={C:4d} T={T:6d} {nm:14s} {us:9.1f} us  {fl / us / 1e6:6.1f} TF/s   {k}", flush=True)
