"""CPU tests of the training-step augmentation row: the oracle against the fixtures the
reference's own classes produced (tests/golden/make_golden_aug.py), the product's host-side planning against the
same fixtures (plans applied by the oracle -- no GPU here), and the EffectScheduler mirror against the reference's
seeded trace."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import wv_oracle_aug as OA
from waveverify_amd import augment as A
from waveverify_amd import effect_scheduler as ES

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "augment.npz"))


def cases(g):
    for i, row in enumerate(g["cases"]):
        seed, B, C, T, sr = (int(v) for v in row[:5])
        yield i, seed, B, C, T, sr, float(row[5])


def test_oracle_reproduces_reference(gold):
    for i, seed, B, C, T, sr, win in cases(gold):
        np.random.seed(seed)
        torch.manual_seed(seed)
        wm, gt, upd, st = OA.localization_forward(gold[f"c{i}_orig"], gold[f"c{i}_wm"], int(sr * win))
        assert np.array_equal(wm, gold[f"c{i}_loc_wm"]) and np.array_equal(upd, gold[f"c{i}_loc_upd"])
        assert np.array_equal(gt, gold[f"c{i}_loc_gt"].astype(np.float32))
        assert np.array_equal([st[k] for k in ("original_revert", "zero_replace", "cross_substitute", "unchanged")],
                              gold[f"c{i}_stats_loc"])
        wm2, upd2, gt2, st2, method = OA.sequence_forward(upd, wm, gt, sr)
        assert method == str(gold["methods"][i])
        assert np.array_equal(wm2, gold[f"c{i}_seq_wm"]) and np.array_equal(upd2, gold[f"c{i}_seq_upd"])
        assert np.array_equal(gt2, gold[f"c{i}_seq_gt"].astype(np.float32))
        assert np.array_equal([st2[k] for k in ("reverse", "circular_shift", "shuffle", "chunk_shuffle", "unchanged")],
                              gold[f"c{i}_stats_seq"])


def test_host_planning_draws_like_the_reference(gold):
    """Same seeds -> the product's plan and sequence map, applied by the oracle's draw-free halves, give the
    reference's tensors and statistics: the RNG call order is the reference's."""
    for i, seed, B, C, T, sr, win in cases(gold):
        np.random.seed(seed)
        torch.manual_seed(seed)
        loc, seq = A.LocalizationAugmentation(sr, win), A.SequenceAugmentation(sr)
        plan = loc.draw_plan(B, T)
        st = dict(loc._finish_stats(B, T))
        wm, upd, gt = OA.apply_plan(gold[f"c{i}_orig"], gold[f"c{i}_wm"], plan, loc.segment_length)
        assert np.array_equal(wm, gold[f"c{i}_loc_wm"]) and np.array_equal(upd, gold[f"c{i}_loc_upd"])
        assert np.array_equal(gt, gold[f"c{i}_loc_gt"].astype(np.float32))
        assert np.array_equal([st[k] for k in ("original_revert", "zero_replace", "cross_substitute", "unchanged")],
                              gold[f"c{i}_stats_loc"])
        method, sm = seq.draw(B, T)
        st2 = dict(seq._finish_stats(B))
        assert method == str(gold["methods"][i])
        out = [OA.apply_seqmap(x, sm.mode, sm.a, sm.b, sm.c, sm.perm) for x in (wm, upd, gt)]
        assert out[0].shape[-1] == sm.t_out
        assert np.array_equal(out[0], gold[f"c{i}_seq_wm"]) and np.array_equal(out[1], gold[f"c{i}_seq_upd"])
        assert np.array_equal(out[2], gold[f"c{i}_seq_gt"].astype(np.float32))
        assert np.array_equal([st2[k] for k in ("reverse", "circular_shift", "shuffle", "chunk_shuffle", "unchanged")],
                              gold[f"c{i}_stats_seq"])


def test_augmenter_argument_errors():
    with pytest.raises(ValueError, match="Sample rate must be positive"):
        A.LocalizationAugmentation(0, 0.1)
    with pytest.raises(ValueError, match="Window duration must be positive"):
        A.LocalizationAugmentation(16000, 0.0)
    with pytest.raises(ValueError, match="Invalid augmentation methods"):
        A.SequenceAugmentation(16000, ["reverse", "stretch"])
    assert A.LocalizationAugmentation(16000, 0.1).segment_length == 1600
    # the reference's selection ignores `methods`; a drawn method that is not in the statistics dict surfaces as
    # the wrapped KeyError (seq_augmentation.py:170, 267-269)
    s = A.SequenceAugmentation(16000, ["shuffle"])
    np.random.seed(1)                                   # first uniform of seed 1 is 0.417 -> circular_shift
    with pytest.raises(RuntimeError, match="Failed to apply augmentation"):
        s.forward(torch.zeros(1, 1, 8), torch.zeros(1, 1, 8), torch.zeros(1, 1, 8))
    with pytest.raises(RuntimeError, match="same shape"):
        A.SequenceAugmentation(16000).forward(torch.zeros(1, 1, 8), torch.zeros(1, 1, 9), torch.zeros(1, 1, 8))
    with pytest.raises(ValueError, match="Shape mismatch"):
        A.LocalizationAugmentation().forward(torch.zeros(1, 1, 8), torch.zeros(1, 1, 9))
    with pytest.raises(RuntimeError, match="GPU"):
        A.LocalizationAugmentation().forward(torch.zeros(2, 1, 16000), torch.zeros(2, 1, 16000))


def test_chunk_swap_map_is_a_valid_swap():
    np.random.seed(5)
    s = A.SequenceAugmentation(16000)
    for T in (16000, 1001, 9):
        sm = s.chunk_swap_map(T)
        assert sm is not None and sm.c == T // 4 and abs(sm.a - sm.b) >= sm.c and max(sm.a, sm.b) + sm.c <= T
        x = np.arange(T, dtype=np.float32)[None, None]
        y = OA.apply_seqmap(x, sm.mode, sm.a, sm.b, sm.c)
        assert np.array_equal(np.sort(y.ravel()), x.ravel())
    assert s.chunk_swap_map(3) is None


# ---- EffectScheduler ---------------------------------------------------------------------------------------------
def _plain(p):
    return {k: (v if isinstance(v, str) else float(v)) for k, v in p.items()}


def test_effect_scheduler_trace_vs_reference():
    ref = json.load(open(os.path.join(GOLD, "effect_scheduler.json")))
    np.random.seed(ref["seed"])
    s = ES.EffectScheduler(ref["grid"], beta=ref["beta"], ber_threshold=ref["ber_threshold"], miou_threshold=ref["miou_threshold"])
    for it, step in enumerate(ref["trace"]):
        sel = s.select_effects(4 if it % 3 else 9)
        assert [[str(n), _plain(p)] for n, p in sel] == step["selected"]
        for (n, p), (ber, miou) in zip(sel, step["metrics"]):
            b2, m2 = float(np.random.uniform(0.0, 0.5)), float(np.random.uniform(0.4, 1.0))
            assert (b2, m2) == (ber, miou)                 # the generator is where the reference's was
            s.update_effect_metrics(n, p, ber, miou)
        if it % 2:
            s.adapt_effect_probabilities()
        assert s.get_effect_probabilities() == step["probabilities"]
    assert [[str(n), _plain(p)] for n, p in s.select_all_effects()] == ref["select_all"]
    st = s.get_effect_statistics()
    assert {n: {k: (None if v is None else float(v)) for k, v in d.items()} for n, d in st.items()} == ref["statistics"]
    assert s.effect_usage_stats == ref["usage"] and s.total_effects == ref["total_effects"]


def test_effect_scheduler_errors_and_edges():
    with pytest.raises(ValueError, match="Beta"):
        ES.EffectScheduler({"identity": {}}, beta=1.0)
    with pytest.raises(ValueError, match="BER threshold"):
        ES.EffectScheduler({"identity": {}}, ber_threshold=2)
    with pytest.raises(ValueError, match="mIoU threshold"):
        ES.EffectScheduler({"identity": {}}, miou_threshold=-1)
    bad = {"bandpass_filter": {"cutoff_freq_low": {"choices": [5000]}, "cutoff_freq_high": {"choices": [100, 200]}}}
    with pytest.raises(ES.ParameterValidationError, match="no valid frequency combinations"):
        ES.EffectScheduler(bad)
    s = ES.EffectScheduler({"identity": {}, "bandpass_filter": {"cutoff_freq_low": {"choices": [100, 4500]},
                                                                 "cutoff_freq_high": {"choices": [3000, 5000]}}})
    with pytest.raises(ValueError, match="must be positive"):
        s.select_effects(0)
    with pytest.raises(ES.InvalidEffectError):
        s.update_effect_metrics("nope", {}, 0.1, 0.9)
    with pytest.raises(ES.InvalidMetricError, match="BER"):
        s.update_effect_metrics("identity", {}, 1.5, 0.9)
    with pytest.raises(ES.InvalidMetricError, match="mIoU"):
        s.update_effect_metrics("identity", {}, 0.5, -0.1)
    np.random.seed(0)
    for _ in range(50):                                    # low < high always holds after the repair step
        for n, p in s.select_effects(2):
            if n == "bandpass_filter":
                assert p["cutoff_freq_low"] < p["cutoff_freq_high"]
    assert len(s.select_effects(99)) == 2                  # capped at the number of effects (watermarking.py:537 quirk)
    s.adapt_effect_probabilities()                          # no metrics yet: stays uniform
    assert s.get_effect_probabilities() == {"identity": 0.5, "bandpass_filter": 0.5}
    assert s.make_hashable({"b": [1, {"c": np.array([1, 2])}], "a": (3,)}) == (("a", (3,)), ("b", (1, (("c", (1, 2)),))))
    lines = []
    s.log_adaptive_behavior(lines.append)
    assert any("EFFECT SCHEDULER ADAPTIVE BEHAVIOR" in ln for ln in lines)


# ---- sinc-filter / resample taps (host side; parity with julius / torchaudio themselves is UNPINNED: see waveverify_amd/effects.py) ----
def test_filter_taps_and_resample_kernels_vs_second_restatement():
    import math
    from oracle import wv_oracle_fx as OF
    from waveverify_amd import effects as E
    for cutoff in (0.375, 0.0625, 0.0125):
        taps, half = E.lowpass_taps([cutoff])
        assert half == int(8 / cutoff / 2) and taps.shape == (1, 2 * half + 1)
        assert np.abs(taps[0] - OF.lowpass_filter_taps(cutoff, half)).max() <= 2e-7
        assert abs(float(taps[0].astype(np.float64).sum()) - 1.0) <= 1e-5
    taps, half = E.lowpass_taps([0.05, 0.4])                        # band-pass bank: the lower cutoff sets the width
    assert half == 80 and taps.shape == (2, 161)
    with pytest.raises(ValueError, match="above 0.5"):
        E.lowpass_taps([0.6])
    for orig, new in ((16000, 8000), (8000, 16000), (44100, 16000), (16000, 12000)):
        k, width, o, n = E.resample_kernels(orig, new)
        g = math.gcd(orig, new)
        assert (o, n) == (orig // g, new // g) and k.shape == (n, 2 * width + o)
        assert width == math.ceil(6 * o / (min(o, n) * 0.99))
        # an impulse through the oracle's per-output formula reads the same taps back
        T = 4 * o
        x = np.zeros((1, T)); x[0, 2 * o] = 1.0
        y = OF.resample(x, orig, new)[0]
        m = 2 * n + (n // 2)                                         # output whose window covers the impulse
        nn, f = divmod(m, n)
        j = 2 * o - (nn * o - width)
        assert 0 <= j < k.shape[1] and abs(y[m] - k[f, j]) <= 2e-7
    assert E.AudioEffects._cutoff(3000, 16000) == 0.375 and E.AudioEffects._cutoff(9000, 16000) == (8000 - 1e-5) / 8000
