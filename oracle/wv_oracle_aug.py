"""CPU restatement (numpy) of the reference's temporal augmentations -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file; the product
(waveverify_amd/augment.py -> csrc/wv_aug.hip) never does.  Pinned to the reference itself by
tests/golden/augment.npz (tests/golden/make_golden_aug.py runs the reference's classes under fixed seeds).

    localization_forward   /root/reference/utils/localization_augmentation.py:212-321
    sequence_forward       /root/reference/utils/seq_augmentation.py:100-273
Both walk the data the way the reference does (clip by clip, slice assignments), drawing from numpy's global
generator (and torch's, for the shuffle permutation) in the reference's order; `apply_plan` / `apply_seqmap` are
the draw-free halves used to check the HIP kernel on arbitrary plans."""
from __future__ import annotations

import numpy as np


def localization_forward(original, watermarked, segment_length):
    """-> (watermarked', presence mask, updated original, stats in % of B*T).  Draws: per clip one
    choice(starts, n, replace=False); per chosen segment rand(); for a cross-substitution choice(others)."""
    original = original.astype(np.float32).copy()
    upd = original.copy()
    wm = watermarked.astype(np.float32).copy()
    B, _, T = wm.shape
    gt = np.ones_like(wm)
    stats = dict(original_revert=0, zero_replace=0, cross_substitute=0, unchanged=0)
    nmod = int(int(np.ceil(T / segment_length)) * 0.20)
    for b in range(B):
        for start in np.random.choice(np.arange(0, T, segment_length), nmod, replace=False):
            end = min(start + segment_length, T)
            u = np.random.rand()
            if u < 0.33:                                             # :128-133
                wm[b, :, start:end] = original[b, :, start:end]
                stats["original_revert"] += end - start
            elif u < 0.66:                                           # :151-155
                wm[b, :, start:end] = 0
                upd[b, :, start:end] = 0
                stats["zero_replace"] += end - start
            elif B >= 2:                                             # :157-193
                j = np.random.choice([k for k in range(B) if k != b])
                wm[b, :, start:end] = original[j, :, start:end]
                upd[b, :, start:end] = original[j, :, start:end]
                stats["cross_substitute"] += end - start
            else:
                continue
            gt[b, :, start:end] = 0
    stats["unchanged"] = B * T - stats["original_revert"] - stats["zero_replace"] - stats["cross_substitute"]
    return wm, gt, upd, {k: float(v / (B * T) * 100) for k, v in stats.items()}


def sequence_forward(upd, wm, gt, sample_rate):
    """-> (watermarked', updated original', mask', stats in % of B, method).  Draws: rand(); for the circular
    shift randint(1, T); for the shuffle torch.randperm(n_segments)."""
    import torch
    B, _, T = wm.shape
    stats = dict(reverse=0, circular_shift=0, shuffle=0, chunk_shuffle=0, unchanged=0)
    u = np.random.rand()
    arrs = [wm, upd, gt]
    if u < 0.3:
        method = "reverse"
        arrs = [a[:, :, ::-1].copy() for a in arrs]
        stats["reverse"] += B
    elif u < 0.7:
        method = "circular_shift"
        s = np.random.randint(1, T)
        arrs = [np.roll(a, s, axis=2) for a in arrs]
        stats["circular_shift"] += B
    elif u < 1.0:
        method = "shuffle"
        seg = int(0.5 * sample_rate)
        if T >= 2 * seg:
            n = T // seg
            perm = torch.randperm(n).numpy()
            arrs = [a[:, :, : n * seg].reshape(a.shape[0], a.shape[1], n, seg)[:, :, perm].reshape(a.shape[0], a.shape[1], -1)
                    for a in arrs]
        else:
            method = "unchanged"
        stats["shuffle"] += B
    else:
        method = "unchanged"
        stats["unchanged"] += B
    return arrs[0], arrs[1], arrs[2], {k: float(v / B * 100) for k, v in stats.items()}, method


def apply_plan(original, watermarked, plan, seg_len):
    """plan[B][nseg]: 0 keep, 1 revert, 2 zero, 3 + j clip j's original -> (watermarked', updated original, mask)."""
    wm, upd, gt = watermarked.copy(), original.copy(), np.ones_like(watermarked)
    B, _, T = wm.shape
    for b in range(B):
        for s, code in enumerate(plan[b]):
            lo, hi = s * seg_len, min((s + 1) * seg_len, T)
            if code == 1:
                wm[b, :, lo:hi] = original[b, :, lo:hi]
            elif code == 2:
                wm[b, :, lo:hi] = 0
                upd[b, :, lo:hi] = 0
            elif code >= 3:
                wm[b, :, lo:hi] = original[code - 3, :, lo:hi]
                upd[b, :, lo:hi] = original[code - 3, :, lo:hi]
            if code:
                gt[b, :, lo:hi] = 0
    return wm, upd, gt


def apply_seqmap(x, mode, a=0, b=0, c=0, perm=None):
    """mode 0 identity, 1 flip, 2 roll by a, 3 permutation of a-sample segments, 4 swap of chunks [a,a+c) / [b,b+c)."""
    if mode == 0:
        return x.copy()
    if mode == 1:
        return x[..., ::-1].copy()
    if mode == 2:
        return np.roll(x, a, axis=-1)
    if mode == 3:
        n = len(perm)
        return x[..., : x.shape[-1] // a * a].reshape(*x.shape[:-1], -1, a)[..., perm, :].reshape(*x.shape[:-1], n * a)
    y = x.copy()
    y[..., a:a + c] = x[..., b:b + c]
    y[..., b:b + c] = x[..., a:a + c]
    return y
