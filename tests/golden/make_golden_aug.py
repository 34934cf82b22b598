#!/usr/bin/env python3
"""Generate tests/golden/augment.npz and tests/golden/effect_scheduler.json by running the REFERENCE's own
LocalizationAugmentation / SequenceAugmentation (/root/reference/utils/localization_augmentation.py,
seq_augmentation.py) and EffectScheduler (utils/effect_scheduler.py) under fixed seeds (build container only).

The two augmentation files import torchaudio and audiotools at module level (absent here, unused by the code that
is run), so they are loaded by path with test-side stand-ins registered in sys.modules, like make_golden.py does.
Only data is written: seeds, inputs, and the reference's outputs.

Usage (from repo root, in the build container):  python tests/golden/make_golden_aug.py"""
from __future__ import annotations

import importlib.util
import json
import logging
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"

# (seed, B, C, T, sample_rate, window_duration): seeds picked so that every branch of both modules is hit
CASES = [(0, 4, 2, 1030, 1000, 0.05), (1, 4, 2, 1030, 1000, 0.05), (2, 4, 1, 1030, 1000, 0.05), (3, 3, 1, 2000, 1000, 0.05),
         (4, 1, 1, 1000, 1000, 0.05), (5, 5, 1, 900, 1000, 0.05), (6, 2, 1, 16000, 16000, 0.1), (7, 3, 1, 16000, 16000, 0.1),
         (8, 4, 2, 1030, 1000, 0.05), (11, 4, 1, 1030, 1000, 0.05), (16, 5, 1, 900, 1000, 0.05)]

GRID = {
    "identity": {},
    "highpass_filter": {"cutoff_freq": {"choices": [100, 200, 300]}},
    "median_filter": {"kernel_size": {"choices": [3, 5, 7]}},
    "bandpass_filter": {"cutoff_freq_low": {"choices": [100, 200, 4000]}, "cutoff_freq_high": {"choices": [3000, 4000, 5000]}},
    "speed": {"speed": {"choices": [0.9, 1.1]}, "mode": "fixed"},
}


def _load(name):
    spec = importlib.util.spec_from_file_location("ref_" + name, f"{REF}/utils/{name}.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _stubs():
    sys.dont_write_bytecode = True
    logging.disable(logging.CRITICAL)
    os.environ.setdefault("MPLBACKEND", "Agg")
    for name in ("torchaudio", "audiotools"):
        sys.modules[name] = types.ModuleType(name)

    class AudioSignal:                       # the two attributes the augmenters' callers read
        def __init__(self, audio_data, sample_rate):
            self.audio_data, self.sample_rate = audio_data, sample_rate
    sys.modules["audiotools"].AudioSignal = AudioSignal


def scheduler_trace(es_mod):
    """A seeded scenario: selections, metric updates, adaptation -- everything the scheduler returns, as JSON."""
    np.random.seed(123)
    s = es_mod.EffectScheduler(GRID, beta=0.9, ber_threshold=0.2, miou_threshold=0.6)
    trace = []
    for it in range(12):
        sel = s.select_effects(4 if it % 3 else 9)
        step = {"selected": [[str(n), {k: (v if isinstance(v, str) else float(v)) for k, v in p.items()}] for n, p in sel], "metrics": []}
        for n, p in sel:
            ber, miou = float(np.random.uniform(0.0, 0.5)), float(np.random.uniform(0.4, 1.0))
            s.update_effect_metrics(n, p, ber, miou)
            step["metrics"].append([ber, miou])
        if it % 2:
            s.adapt_effect_probabilities()
        step["probabilities"] = {k: float(v) for k, v in s.get_effect_probabilities().items()}
        trace.append(step)
    allsel = s.select_all_effects()
    stats = s.get_effect_statistics()
    return {"grid": GRID, "seed": 123, "beta": 0.9, "ber_threshold": 0.2, "miou_threshold": 0.6, "trace": trace,
            "select_all": [[str(n), {k: (v if isinstance(v, str) else float(v)) for k, v in p.items()}] for n, p in allsel],
            "statistics": {n: {k: (None if v is None else float(v)) for k, v in st.items()} for n, st in stats.items()},
            "usage": {k: int(v) for k, v in s.effect_usage_stats.items()}, "total_effects": int(s.total_effects)}


def main():
    import torch
    _stubs()
    la, sa, es = _load("localization_augmentation"), _load("seq_augmentation"), _load("effect_scheduler")
    out = {"cases": np.array(CASES, dtype=np.float64)}
    methods, branches = [], set()
    for i, (seed, B, C, T, sr, win) in enumerate(CASES):
        # inputs that identify their own position (exact in float32): every sample of the outputs names the tensor,
        # clip, channel and time it was copied from -- the augmentations are pure copies -- and the file stays small
        row = np.arange(B * C, dtype=np.float32).reshape(B, C, 1)
        orig = (row + np.arange(T, dtype=np.float32) / 65536.0).astype(np.float32)
        wm = (-orig - 0.5).astype(np.float32)
        np.random.seed(seed)
        torch.manual_seed(seed)
        loc, seq = la.LocalizationAugmentation(sr, win), sa.SequenceAugmentation(sr)
        sig, gt, upd, st_loc = loc(torch.from_numpy(orig), torch.from_numpy(wm))
        st_loc = dict(st_loc)
        sig2, upd2, gt2, st_seq, method = seq(upd, sig.audio_data, gt)
        out[f"c{i}_orig"], out[f"c{i}_wm"] = orig, wm
        out[f"c{i}_loc_wm"], out[f"c{i}_loc_gt"], out[f"c{i}_loc_upd"] = sig.audio_data.numpy(), gt.numpy().astype(np.uint8), upd.numpy()
        out[f"c{i}_seq_wm"], out[f"c{i}_seq_gt"], out[f"c{i}_seq_upd"] = sig2.audio_data.numpy(), gt2.numpy().astype(np.uint8), upd2.numpy()
        out[f"c{i}_stats_loc"] = np.array([st_loc[k] for k in ("original_revert", "zero_replace", "cross_substitute", "unchanged")])
        out[f"c{i}_stats_seq"] = np.array([st_seq[k] for k in ("reverse", "circular_shift", "shuffle", "chunk_shuffle", "unchanged")])
        methods.append(method)
        branches.add(method)
        print(i, (seed, B, C, T), method, {k: round(v, 2) for k, v in st_loc.items()}, sig2.audio_data.shape)
    out["methods"] = np.array(methods)
    assert {"reverse", "circular_shift", "shuffle", "unchanged"} <= branches, branches
    np.savez_compressed(os.path.join(HERE, "augment.npz"), **out)
    with open(os.path.join(HERE, "effect_scheduler.json"), "w") as f:
        json.dump(scheduler_trace(es), f, indent=1)
    print("wrote augment.npz, effect_scheduler.json")


if __name__ == "__main__":
    main()
