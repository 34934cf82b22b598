#!/usr/bin/env python3
"""Kernel micro-benchmark / ablation for K1 (pw_dw): one layer shape, event-timed through the
library's profiler.  python tools/kbench.py [C T B]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from waveverify_amd import ops, profile, _lib

def run(C, T, B, M=None, ks=5, stride=1, resid=True, reps=5, flags=0, film=False, prec="f32"):
    M = M or C
    rng = np.random.default_rng(0)
    X = torch.randn(B, C, T, device="cuda")
    w_pw = rng.standard_normal((M, C, 1)).astype(np.float32) * C ** -0.5
    w_dw = rng.standard_normal((M, 1, ks)).astype(np.float32)
    b = rng.standard_normal(M).astype(np.float32)
    Tout = -(-T // stride)
    R = torch.randn(B, M, Tout, device="cuda") if resid else None
    _lib.load().wv_debug_flags(flags)
    ops.set_precision(prec)
    profile.reset(); profile.enable(True)
    for _ in range(reps):
        ops.pw_dw(X, w_pw, w_dw, b, resid=R, stride=stride, pre_scale=0.87, pre_elu=True, out_scale=0.5)
    profile.enable(False)
    e = profile.collect()[0]
    _lib.load().wv_debug_flags(0)
    ops.set_precision("f32")
    us = e["ms"] / e["launches"] * 1e3
    tf = e["flops"] / e["launches"] / (us * 1e-6) / 1e12
    gb = e["bytes"] / e["launches"] / (us * 1e-6) / 1e9
    return us, tf, gb, e["kernel"]

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "down":
        for C, T, r in ((64, 16000, 2), (128, 8000, 4), (256, 2000, 5), (512, 400, 8)):
            for fl, nm in (((255 << 8) | 64, "warmup"), (255 << 8, "full"), ((255 << 8) | 1, "no-epilogue")):
                us, tf, gb, k = run(C, T, 256, M=2 * C, ks=2 * r, stride=r, resid=False, flags=fl)
                if nm != "warmup":
                    print(f"down C={C:4d}->{2*C:4d} T={T:6d} r={r} {k:22s} {nm:14s} {us:9.1f} us  {tf:6.1f} TF/s {gb:7.1f} GB/s", flush=True)
        sys.exit(0)
    shapes = [(128, 8000, 256), (64, 16000, 256), (512, 400, 256), (768, 400, 256), (96, 16000, 256)]
    if len(sys.argv) == 4:
        shapes = [tuple(int(v) for v in sys.argv[1:4])]
    names = {(255 << 8) | 64: "warmup", 255 << 8: "f32 full", (255 << 8) | 17: "no-epi, no global loads (LDS+MFMA)",
             (255 << 8) | 25: "no-epi, no loads, no LDS reads (MFMA only)"}
    for C, T, B in shapes:
        for fl, nm in list(names.items()):
            us, tf, gb, k = run(C, T, B, flags=fl & ~128, prec="f16x3" if fl & 128 else "f32")
            print(f"C={C:4d} T={T:6d} {k:24s} {nm:42s} {us:9.1f} us  {tf:6.1f} TF/s {gb:7.1f} GB/s", flush=True)
