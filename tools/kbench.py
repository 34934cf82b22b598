#!/usr/bin/env python3
"""Kernel micro-benchmark for K1 (pw_dw): one layer shape at a time, event-timed through the library's
profiler.  The ablation switches of round 1 are gone from the product kernels; what a structure costs is
measured in tools/mfma_peak.hip instead.

    python tools/kbench.py [C T B]          one ResnetBlock-half shape
    python tools/kbench.py down             the four Downsample+FiLM shapes"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from waveverify_amd import _lib
if "--lib" in sys.argv:                      # an A/B variant built by tools/variant.sh
    i = sys.argv.index("--lib"); _lib.LIB_PATH = os.path.abspath(sys.argv[i + 1]); del sys.argv[i:i + 2]
from waveverify_amd import ops, profile


def run(C, T, B, M=None, ks=5, stride=1, resid=True, reps=5, film=False, pre_elu=True, yact=False):
    M = M or C
    rng = np.random.default_rng(0)
    X = torch.randn(B, C, T, device="cuda")
    w_pw = rng.standard_normal((M, C, 1)).astype(np.float32) * C ** -0.5
    w_dw = rng.standard_normal((M, 1, ks)).astype(np.float32)
    b = rng.standard_normal(M).astype(np.float32)
    Tout = -(-T // stride)
    R = torch.randn(B, M, Tout, device="cuda") if resid else None
    F = torch.randn(B, 4, 2, device="cuda") if film else None
    kw = dict(resid=R, film=F, bands=4 if film else 1, stride=stride, pre_scale=0.87 if pre_elu else 1.0,
              pre_elu=pre_elu, out_scale=0.5, act_scale=0.9 if yact else None)
    ops.pw_dw(X, w_pw, w_dw, b, **kw)                    # warm-up
    profile.reset(); profile.enable(True)
    for _ in range(reps):
        ops.pw_dw(X, w_pw, w_dw, b, **kw)
    profile.enable(False)
    e = profile.collect()[0]
    us = e["ms"] / e["launches"] * 1e3
    tf = e["flops"] / e["launches"] / (us * 1e-6) / 1e12
    gb = e["bytes"] / e["launches"] / (us * 1e-6) / 1e9
    return us, tf, gb, e["kernel"]


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "down":
        for C, T, r in ((64, 16000, 2), (128, 8000, 4), (256, 2000, 5), (512, 400, 8)):
            for pe in (True, False):
                us, tf, gb, k = run(C, T, 256, M=2 * C, ks=2 * r, stride=r, resid=False, film=True, pre_elu=pe, yact=True)
                print(f"down C={C:4d}->{2*C:4d} T={T:6d} r={r} {k:24s} pre_elu={int(pe)} {us:9.1f} us  {tf:6.1f} TF/s {gb:7.1f} GB/s", flush=True)
        sys.exit(0)
    shapes = [(128, 8000, 256), (256, 2000, 256), (384, 2000, 256), (512, 400, 256), (768, 400, 256), (1024, 50, 256),
              (64, 16000, 256), (96, 16000, 256), (192, 8000, 256)]
    if len(sys.argv) == 4:
        shapes = [tuple(int(v) for v in sys.argv[1:4])]
    for C, T, B in shapes:
        for resid in (True, False):
            for pe in (True, False):
                us, tf, gb, k = run(C, T, B, resid=resid, pre_elu=pe, yact=not resid)
                print(f"C={C:4d} T={T:6d} {k:26s} resid={int(resid)} pre_elu={int(pe)} {us:9.1f} us  {tf:6.1f} TF/s {gb:7.1f} GB/s", flush=True)
