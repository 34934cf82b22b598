"""GPU parity of the augmentation kernels (csrc/wv_aug.hip, through the C ABI): bit-exact against the outputs of the
reference's own classes under the same seeds (tests/golden/augment.npz) and against the oracle on arbitrary plans at
the training batch of BASELINE configs[2] (64 clips x 1 s)."""
import os

import numpy as np
import pytest
import torch

from oracle import wv_oracle_aug as OA
from waveverify_amd import augment as A

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def eq(t, ref):
    got = t.cpu().numpy()
    return got.shape == ref.shape and np.array_equal(got, ref.astype(np.float32))


def test_reference_seeds_bit_exact():
    g = np.load(os.path.join(GOLD, "augment.npz"))
    for i, row in enumerate(g["cases"]):
        seed, B, C, T, sr = (int(v) for v in row[:5])
        win = float(row[5])
        orig, wm = cu(g[f"c{i}_orig"]), cu(g[f"c{i}_wm"])
        # the two modules one after the other, as the reference calls them
        np.random.seed(seed); torch.manual_seed(seed)
        sig, gt, upd, st = A.LocalizationAugmentation(sr, win)(orig, wm)
        assert eq(sig.audio_data, g[f"c{i}_loc_wm"]) and eq(upd, g[f"c{i}_loc_upd"]) and eq(gt, g[f"c{i}_loc_gt"])
        assert sig.sample_rate == sr
        assert np.array_equal([st[k] for k in ("original_revert", "zero_replace", "cross_substitute", "unchanged")], g[f"c{i}_stats_loc"])
        sig2, upd2, gt2, st2, method = A.SequenceAugmentation(sr)(upd, sig.audio_data, gt)
        assert method == str(g["methods"][i])
        assert eq(sig2.audio_data, g[f"c{i}_seq_wm"]) and eq(upd2, g[f"c{i}_seq_upd"]) and eq(gt2, g[f"c{i}_seq_gt"])
        assert np.array_equal([st2[k] for k in ("reverse", "circular_shift", "shuffle", "chunk_shuffle", "unchanged")], g[f"c{i}_stats_seq"])
        # the fused single launch (watermarking.py:487-519)
        np.random.seed(seed); torch.manual_seed(seed)
        sig3, mask3, upd3, st3 = A.TemporalAugmenter(sr, win)(orig, wm)
        assert eq(sig3.audio_data, g[f"c{i}_seq_wm"]) and eq(upd3, g[f"c{i}_seq_upd"]) and eq(mask3, g[f"c{i}_seq_gt"])
        assert st3 == {**st, **st2}


@pytest.mark.parametrize("B,C,T,seg", [(64, 1, 16000, 1600), (7, 2, 16001, 1600), (3, 1, 999, 50), (2, 3, 5, 2), (1, 1, 1, 1600)])
def test_random_plans_and_every_map_vs_oracle(B, C, T, seg):
    rng = np.random.default_rng(B * 1000 + T)
    orig = rng.standard_normal((B, C, T)).astype(np.float32)
    wm = rng.standard_normal((B, C, T)).astype(np.float32)
    nseg = -(-T // seg)
    plan = rng.integers(0, 3 + B, (B, nseg)).astype(np.int32)
    ref = OA.apply_plan(orig, wm, plan, seg)                  # (wm, upd, gt)
    maps = [A.SeqMap(t_out=T), A.SeqMap(A.SEQ_REVERSE, t_out=T)]
    if T > 1:
        maps += [A.SeqMap(A.SEQ_ROLL, a=1, t_out=T), A.SeqMap(A.SEQ_ROLL, a=T - 1, t_out=T), A.SeqMap(A.SEQ_ROLL, a=max(1, T // 3), t_out=T)]
    for sz in (max(1, T // 7), 2):
        if T >= 2 * sz:
            n = T // sz
            maps.append(A.SeqMap(A.SEQ_PERMUTE, a=sz, perm=rng.permutation(n).astype(np.int32), t_out=n * sz))
    if T >= 9:
        c = T // 4
        maps += [A.SeqMap(A.SEQ_CHUNK_SWAP, a=0, b=T - c, c=c, t_out=T), A.SeqMap(A.SEQ_CHUNK_SWAP, a=2 * c, b=c - 1, c=c, t_out=T)]
    for sm in maps:
        outs = A._launch(cu(orig), cu(wm), plan, seg, sm)
        for got, r in zip(outs, ref):
            assert eq(got, OA.apply_seqmap(r, sm.mode, sm.a, sm.b, sm.c, sm.perm)), (sm.mode, sm.a, sm.b, sm.c)
        seq_only = A.apply_sequence_map([cu(wm), None, cu(orig)], sm)
        assert seq_only[1] is None
        assert eq(seq_only[0], OA.apply_seqmap(wm, sm.mode, sm.a, sm.b, sm.c, sm.perm))
        assert eq(seq_only[2], OA.apply_seqmap(orig, sm.mode, sm.a, sm.b, sm.c, sm.perm))
    # no plan at all = sequence map of the untouched inputs, mask all ones
    w, u, m = A._launch(cu(orig), cu(wm), None, seg, maps[1])
    assert eq(w, wm[..., ::-1]) and eq(u, orig[..., ::-1]) and bool((m == 1).all())


def test_properties_at_training_batch():
    """64 x 1 s (BASELINE configs[2] per-GPU batch): mask is 0 exactly on the modified segments, about 20 % of them;
    untouched samples are the watermarked input, modified ones never are; inputs are not written."""
    B, T = 64, 16000
    rng = np.random.default_rng(9)
    orig = rng.standard_normal((B, 1, T)).astype(np.float32)
    wm = (orig + 3.0 + np.abs(rng.standard_normal((B, 1, T)))).astype(np.float32)     # never equal to any original sample
    o_d, w_d = cu(orig), cu(wm)
    np.random.seed(3)
    loc = A.LocalizationAugmentation(16000, 0.1)
    sig, mask, upd, st = loc(o_d, w_d)
    assert torch.equal(o_d, cu(orig)) and torch.equal(w_d, cu(wm))
    m = mask.cpu().numpy()
    assert set(np.unique(m)) == {0.0, 1.0} and abs((1 - m.mean(dtype=np.float64)) - 0.2) < 1e-9
    segs = m.reshape(B, 10, 1600)
    assert ((segs.min(-1) == segs.max(-1))).all()                       # whole segments
    out = sig.audio_data.cpu().numpy()
    assert np.array_equal(out[m == 1], wm[m == 1]) and not np.any(out[m == 0] == wm[m == 0])
    assert abs(sum(st.values()) - 100.0) < 1e-9 and abs(st["unchanged"] - 80.0) < 1e-9
    u = upd.cpu().numpy()
    assert np.array_equal(u[m == 1], orig[m == 1])


def test_bad_arguments_are_rejected():
    x = torch.zeros(2, 1, 100).cuda()
    with pytest.raises(ValueError, match="plan out of range"):
        A._launch(x, x, np.full((2, 1), 3 + 2, np.int32), 100, A.SeqMap(t_out=100))
    with pytest.raises(ValueError, match="permutation out of range"):
        A._launch(x, x, None, 100, A.SeqMap(A.SEQ_PERMUTE, a=50, perm=np.array([0, 2], np.int32), t_out=100))
    with pytest.raises(RuntimeError, match="failed"):
        A._launch(x, x, np.zeros((2, 3), np.int32), 100, A.SeqMap(t_out=100))      # nseg != ceil(T / seg_len)
    with pytest.raises(RuntimeError, match="failed"):
        A._launch(x, x, None, 100, A.SeqMap(A.SEQ_ROLL, a=100, t_out=100))
    with pytest.raises(RuntimeError, match="failed"):
        A._launch(x, x, None, 100, A.SeqMap(A.SEQ_CHUNK_SWAP, a=0, b=10, c=25, t_out=100))   # overlapping chunks
