#!/usr/bin/env python3
"""Check the persistent pipelined K1 kernel against the classic one (same inputs), then time both."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from waveverify_amd import ops, profile, _lib
lib = _lib.load()
S = 255 << 8

def one(C, T, B, flags, resid=True, reps=1, film=False):
    rng = np.random.default_rng(0)
    g = torch.Generator(device="cuda").manual_seed(1)
    X = torch.randn(B, C, T, device="cuda", generator=g)
    R = torch.randn(B, C, T, device="cuda", generator=g) if resid else None
    w_pw = rng.standard_normal((C, C, 1)).astype(np.float32) * C ** -0.5
    w_dw = rng.standard_normal((C, 1, 5)).astype(np.float32)
    b = rng.standard_normal(C).astype(np.float32)
    lib.wv_debug_flags(flags)
    profile.reset(); profile.enable(True)
    for _ in range(reps):
        Y = ops.pw_dw(X, w_pw, w_dw, b, resid=R, pre_scale=0.87, pre_elu=True, out_scale=0.5)
    profile.enable(False)
    e = profile.collect()[0]
    lib.wv_debug_flags(0)
    return Y, e["ms"] / e["launches"] * 1e3, e["kernel"], e["flops"] / e["ms"] / 1e9

if __name__ == "__main__":
    # debug flag 32 = classic kernel everywhere; default = pipelined kernel where the launcher picks it
    shapes = [(128, 500, 3), (64, 2000, 5), (64, 124, 9), (128, 8000, 2), (64, 16000, 1), (128, 60, 700), (64, 4, 3)]
    for C, T, B in shapes:
        for resid in (True, False):
            Y0, _, k0, _ = one(C, T, B, 32, resid)
            Y1, _, k1, _ = one(C, T, B, 0, resid)
            d = (Y0 - Y1).abs().max().item()
            print(f"C={C} T={T} B={B} resid={resid} {k0} vs {k1}: max|d|={d:.3e}", flush=True)
            assert d == 0.0, "MISMATCH"
            assert "pipe" in k1, "pipelined kernel not selected"
    if len(sys.argv) > 1:
        for C, T in ((128, 8000), (64, 16000)):
            one(C, T, 256, S)
            for fl, nm in ((32, "classic"), (0, "default")):
                _, us, k, tf = one(C, T, 256, S | fl, reps=4)
                print(f"C={C:4d} T={T:6d} {nm:15s} {k:28s} {us:9.1f} us {tf:6.1f} TF/s", flush=True)
