#!/usr/bin/env python3
"""Per-role time table of one embed+detect pass (library event profiler): where the step goes.

    python tools/roles.py [--batch 256] [--seconds 1]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from waveverify_amd import _lib                         # noqa: E402
if "--lib" in sys.argv:                                 # an A/B variant built by tools/variant.sh
    i = sys.argv.index("--lib")
    _lib.LIB_PATH = os.path.abspath(sys.argv[i + 1])
    del sys.argv[i:i + 2]
from waveverify_amd import profile                      # noqa: E402
from waveverify_amd.core import WaveVerify              # noqa: E402
from waveverify_amd.init import synthetic_clips         # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--seconds", type=float, default=1.0)
    ap.add_argument("--steps", type=int, default=3)
    a = ap.parse_args()
    wv = WaveVerify.random_init(seed=0, device="cuda:0")
    gen, det = wv.model.generator, wv.model.detector
    T = int(16000 * a.seconds)
    x_np, msg_np = synthetic_clips(a.batch, T, seed=1)
    x, msg = torch.from_numpy(x_np).cuda(), torch.from_numpy(msg_np).cuda()
    for it in range(a.steps + 1):
        if it == 1:
            profile.enable(True)
            profile.reset()
        wm = gen.generator(x, msg, add_input=True)
        det.detector_mean_prob(wm)
    rows = profile.collect()
    profile.enable(False)
    tot = sum(r["ms"] for r in rows)
    print(f"total {tot / a.steps:.2f} ms/step")
    for r in sorted(rows, key=lambda r: -r["ms"]):
        ms = r["ms"] / r["launches"]
        print(f"{r['kernel']:28s} {r['role']:18s} n={r['launches'] // a.steps:3d} {1e3 * ms:8.1f} us "
              f"{r['ms'] / a.steps:7.2f} ms/step {r['flops'] / r['ms'] / 1e9:6.1f} TF/s "
              f"{r['bytes'] / r['ms'] / 1e6:7.0f} GB/s")


if __name__ == "__main__":
    main()
