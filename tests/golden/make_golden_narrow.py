#!/usr/bin/env python3
"""Generate tests/golden/narrow_margin_T16000.npz by running the REFERENCE Detector (build container only).

The other fixtures use a wide last-layer bias so that every time-averaged bit probability sits >= 0.17 from
the 0.5 threshold ("0 BER" is then an arithmetic statement).  This one keeps the reference's own scale for
that bias (N(0, 0.05)): probabilities crowd around the threshold, and the fixture records every bit's margin
|p - 0.5| so that a test can compare decisions exactly where the margin exceeds the measured |dp|.
Same import technique as make_golden.py (audiotools stub, model files loaded by path).

Usage (from repo root):  python tests/golden/make_golden_narrow.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import make_golden as MG  # noqa: E402


def main():
    from waveverify_amd.config import default_config
    from waveverify_amd.init import random_state_dict, synthetic_clips
    torch, AudioSignal, RG, RD, RL = MG._import_reference()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    cg, cd = default_config("generator", zero_init=True), default_config("detector", zero_init=True)
    G = MG.build_ref(torch, RG, cg, 0)
    sd = random_state_dict(cd, 0, parametrized=True)
    bias = np.random.default_rng(77).normal(0.0, 0.05, size=sd["last_layer.bias"].shape).astype(np.float32)
    sd["last_layer.bias"] = bias
    D = RD(**MG._ref_kwargs(cd)).eval()
    missing, unexpected = D.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert not [m for m in missing if not m.endswith("spec.weight")] and not unexpected
    x, msg = synthetic_clips(6, 16000, seed=4242)
    with torch.no_grad():
        xt = torch.from_numpy(x)
        wm = G(AudioSignal(xt.clone()), torch.from_numpy(msg)).audio_data + xt
        mp = torch.sigmoid(D(AudioSignal(wm.clone()))).mean(dim=2)
    mp = mp.numpy()
    path = os.path.join(HERE, "narrow_margin_T16000.npz")
    np.savez_compressed(path, wm=wm.numpy(), last_layer_bias=bias, det_mean_prob=mp,
                        det_bits=(mp >= 0.5).astype(np.int32), margin=np.abs(mp - 0.5), seed=np.int64(0))
    m = np.abs(mp - 0.5)
    print(f"wrote {path}: {mp.size} bits, margins min {m.min():.2e} median {np.median(m):.2e} max {m.max():.2e}; "
          f"{(m < 1e-3).sum()} bits within 1e-3 of the threshold")


if __name__ == "__main__":
    main()
