#!/usr/bin/env python3
"""Generate tests/golden/metrics.npz by running the REFERENCE's own BER / MIOU classes
(/root/reference/scripts/evaluate.py:419-516, 575-665; build container only).

`scripts/evaluate.py` imports pystoi, pesq and audiotools at module level (none installed here, none
used by BER / MIOU), so the file is loaded by path with empty test-side stand-ins for those three
modules registered in sys.modules -- the same technique as make_golden.py's audiotools stub.  Only
data is written: inputs, the reference's outputs, and a flag where the reference raises.

Usage (from repo root, in the build container):  python tests/golden/make_golden_metrics.py
"""
from __future__ import annotations

import importlib.util
import logging
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def _import_evaluate():
    sys.dont_write_bytecode = True
    logging.disable(logging.CRITICAL)
    os.environ.setdefault("MPLBACKEND", "Agg")
    for name in ("pystoi", "pesq", "audiotools"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["pesq"].NoUtterancesError = type("NoUtterancesError", (Exception,), {})
    sys.modules["pesq"].pesq = lambda *a, **k: None
    sys.modules["audiotools"].AudioSignal = type("AudioSignal", (), {})
    spec = importlib.util.spec_from_file_location("ref_evaluate", f"{REF}/scripts/evaluate.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def main():
    import torch
    ev = _import_evaluate()
    rng = np.random.default_rng(20260)
    out = {}

    # ---------------------------------------------------------------- BER
    def ber_case(i, logits, bits, mask=None, thr=0.5):
        out[f"ber{i}_logits"] = logits.astype(np.float32)
        out[f"ber{i}_bits"] = bits
        out[f"ber{i}_thr"] = np.float32(thr)
        if mask is not None:
            out[f"ber{i}_mask"] = mask.astype(np.float32)
        try:
            v = ev.BER(threshold=thr)(torch.from_numpy(logits.astype(np.float32)), torch.from_numpy(bits),
                                      None if mask is None else torch.from_numpy(mask.astype(np.float32)))
            out[f"ber{i}_out"] = np.float64(float(v))
        except RuntimeError:
            out[f"ber{i}_out"] = np.float64("nan")          # the reference raises RuntimeError

    n = 0
    for (B, W, T) in ((1, 16, 50), (3, 16, 400), (2, 8, 33), (4, 16, 1)):
        logits = rng.standard_normal((B, W, T)) * 0.3      # narrow margins: many near-threshold means
        bits = rng.integers(0, 2, (B, W)).astype(np.int64)
        ber_case(n, logits, bits); n += 1
        mask = (rng.random((B, 1, T)) < 0.5).astype(np.float32)
        ber_case(n, logits, bits, mask); n += 1
        ber_case(n, logits, bits.astype(np.float32), mask, thr=0.45); n += 1
    B, W, T = 3, 16, 40
    logits = rng.standard_normal((B, W, T))
    bits = rng.integers(0, 2, (B, W)).astype(np.int64)
    mask = np.ones((B, 1, T), np.float32); mask[1] = 0.0                      # one clip with no valid step
    ber_case(n, logits, bits, mask); n += 1
    ber_case(n, logits, bits, np.zeros((B, 1, T), np.float32)); n += 1          # no valid bits at all -> 0
    ber_case(n, np.zeros((B, W, T)), bits); n += 1                             # p = 0.5 exactly: >= decodes 1
    ber_case(n, np.zeros((B, W, T)), bits, mask); n += 1                       # masked mean = 0.5*n/(n+eps) < 0.5
    ber_case(n, logits, bits[:, :8]); n += 1                                   # shape mismatch -> RuntimeError
    ber_case(n, logits, bits, np.ones((B, 1, T + 1), np.float32)); n += 1      # mask mismatch -> RuntimeError
    part = np.zeros((B, 1, T), np.float32); part[:, :, 7:19] = 1.0
    ber_case(n, logits * 4.0, bits, part); n += 1
    out["n_ber"] = np.int64(n)

    # ---------------------------------------------------------------- MIOU
    def miou_case(i, p, g, as_torch=False):
        out[f"miou{i}_p"], out[f"miou{i}_g"] = p, g
        out[f"miou{i}_torch"] = np.int64(as_torch)
        try:
            a, b = (torch.from_numpy(p), torch.from_numpy(g)) if as_torch else (p, g)
            out[f"miou{i}_out"] = np.float64(float(ev.MIOU()(a, b)))
        except RuntimeError:
            out[f"miou{i}_out"] = np.float64("nan")

    n = 0
    for shape in ((1, 1, 64), (4, 1, 1000), (2, 1, 7), (16000,)):
        p = (rng.random(shape) < 0.4).astype(np.float32)
        g = (rng.random(shape) < 0.6).astype(np.float32)
        miou_case(n, p, g); n += 1
        miou_case(n, p.astype(np.int64), g.astype(np.int64), as_torch=True); n += 1
        miou_case(n, p > 0.5, g > 0.5); n += 1                                  # boolean masks
    z, o = np.zeros((2, 1, 30), np.float32), np.ones((2, 1, 30), np.float32)
    for p, g in ((z, z), (o, o), (z, o), (o, z)):                              # empty foreground / background
        miou_case(n, p, g); n += 1
    half = z.copy(); half[:, :, :15] = 1.0
    miou_case(n, half, z); n += 1
    miou_case(n, half, o); n += 1
    miou_case(n, half * 0.5, half); n += 1                                      # non-binary -> RuntimeError
    miou_case(n, half, half[:, :, :20]); n += 1                                 # shape mismatch -> RuntimeError
    out["n_miou"] = np.int64(n)

    path = os.path.join(HERE, "metrics.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {int(out['n_ber'])} BER cases, {int(out['n_miou'])} MIOU cases, {os.path.getsize(path)} bytes")


if __name__ == "__main__":
    main()
