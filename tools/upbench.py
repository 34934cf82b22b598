import os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from waveverify_amd import _lib
if "--lib" in sys.argv: _lib.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
from waveverify_amd import ops, profile
B = 256
for K, M, Tin, r in ((1536, 768, 50, 8), (768, 384, 400, 5), (384, 192, 2000, 4), (192, 96, 8000, 2)):
    rng = np.random.default_rng(0)
    X16 = ops.h16_from_f32(torch.randn(B, K, Tin, device="cuda"))
    w_ct = rng.standard_normal((K, 1, 2 * r)).astype(np.float32) * 0.5
    w_pw = rng.standard_normal((M, K, 1)).astype(np.float32) * K ** -0.5
    b = rng.standard_normal(M).astype(np.float32) * 0.1
    def f(): ops.h16_upsample(X16, w_ct, w_pw, b, r)
    f(); f(); profile.reset(); profile.enable(True)
    for _ in range(5): f()
    profile.enable(False)
    es = [e for e in profile.collect() if e["kernel"].startswith("conv16")]
    us = sum(e["ms"] for e in es) / 5 * 1e3
    fl = 2.0 * B * M * r * 2 * K * Tin
    print(f"K={K} M={M} Tin={Tin} r={r}: {us:8.1f} us {fl / us / 1e6:7.1f} TF  {es[0]['kernel']}", flush=True)
