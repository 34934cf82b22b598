#!/usr/bin/env python3
"""Time of the sinc-filter / resample kernels (csrc/wv_fx.hip) at the training batch (64 clips x 1 s).  python tools/fxbench.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from waveverify_amd import effects as E

x = torch.randn(64, 1, 16000, device="cuda") * 0.1
def t_of(f, reps=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for name, f, macs in (("lowpass 3000 Hz (21 taps)", lambda: E.lowpass(x, 0.375), 21), ("highpass 500 Hz (129 taps)", lambda: E.highpass(x, 0.0625), 129),
                      ("highpass 100 Hz (641 taps)", lambda: E.highpass(x, 0.0125), 641), ("bandpass 300-3000 Hz (2 x 427 taps)", lambda: E.bandpass(x, 0.0375, 0.375), 854),
                      ("resample 16k -> 8k -> 16k", lambda: E.AudioEffects.resample(x, 8000), 28 + 28)):
    us = t_of(f)
    print(f"{name:40s} {us:9.1f} us   {2.0 * 64 * 16000 * macs / us / 1e6:7.2f} TFLOP/s (incl. the host-side tap build and upload)")
