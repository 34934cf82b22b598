#!/usr/bin/env python3
"""The f16 mode's one-launch ResnetBlock kernel (wv_h16.hip rh_kernel) alone, on the layer shapes of the three nets at 256 clips:
time per launch (the library's own event pair around the kernel), GB/s of algorithmic bytes (read x + write y, f16), TFLOP/s.
python tools/rhbench.py [--lib tools/bin/libwv_X.so] [--only C[,C..]]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from waveverify_amd import _lib
if "--lib" in sys.argv:                      # an A/B variant built by tools/variant.sh
    _lib.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
from waveverify_amd import ops, profile
only = [int(v) for v in sys.argv[sys.argv.index("--only") + 1].split(",")] if "--only" in sys.argv else None
B = 256
for C, T in ((32, 16000), (64, 16000), (128, 8000), (256, 2000), (512, 400), (768, 400), (384, 2000), (192, 8000), (96, 16000)):
    if only and C not in only:
        continue
    rng = np.random.default_rng(0)
    X16 = ops.h16_from_f32(torch.randn(B, C, T, device="cuda"))
    w1 = rng.standard_normal((C, C, 1)).astype(np.float32) * C ** -0.5; w2 = w1[::-1].copy()
    d1 = rng.standard_normal((C, 1, 5)).astype(np.float32) * 0.4; d2 = d1[::-1].copy()
    b1 = rng.standard_normal(C).astype(np.float32) * 0.1; b2 = b1[::-1].copy()
    def f(): ops.h16_resblock(X16, w1, d1, b1, w2, d2, b2, pre_scale=0.87, out_scale=0.5)
    f(); f(); profile.reset(); profile.enable(True)
    for _ in range(5): f()
    profile.enable(False)
    es = [e for e in profile.collect() if e["kernel"].startswith("resblock16")]
    us = sum(e["ms"] for e in es) / 5 * 1e3
    by, fl = 2 * 2.0 * B * C * T, 2 * 2.0 * B * C * (C * T + 5 * T)
    print(f"C={C:4d} T={T:6d} {us:9.1f} us  {by / us / 1e3:7.1f} GB/s  {fl / us / 1e6:7.1f} TF/s   {es[0]['kernel']}", flush=True)
