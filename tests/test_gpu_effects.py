"""Sinc-filter / resample effects on the GPU against the independent float64 restatement.  PARITY UNPINNED
with respect to julius / torchaudio themselves (absent from the image; see waveverify_amd/effects.py)."""
import numpy as np
import pytest
import torch

from oracle import wv_oracle_fx as OF
from waveverify_amd import effects as E

pytestmark = pytest.mark.gpu


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def err(got, ref):
    g = got.cpu().numpy().astype(np.float64)
    assert g.shape == ref.shape, (g.shape, ref.shape)
    return float(np.abs(g - ref).max() / max(1e-12, np.abs(ref).max()))


@pytest.mark.parametrize("T", [16000, 1001, 37])
def test_filters_vs_float64_restatement(T):
    rng = np.random.default_rng(T)
    x = (0.1 * rng.standard_normal((3, 1, T))).astype(np.float32)
    for cutoff in (0.375, 0.0625, 0.0125, 0.5):                       # 3000 / 500 / 100 Hz as the reference normalises them, and the limit
        assert err(E.lowpass(cu(x), cutoff), OF.lowpass(x, cutoff)) <= 2e-5
        assert err(E.highpass(cu(x), cutoff), OF.highpass(x, cutoff)) <= 2e-5 * max(1.0, np.abs(x).max() / np.abs(OF.highpass(x, cutoff)).max())
    assert err(E.bandpass(cu(x), 0.0375, 0.4), OF.bandpass(x, 0.0375, 0.4)) <= 5e-5
    # a constant passes a lowpass unchanged (taps sum to 1, replicate padding) and is removed by the highpass
    c = torch.full((2, 1, T), 0.25).cuda()
    assert float((E.lowpass(c, 0.05) - 0.25).abs().max()) <= 1e-6 and float(E.highpass(c, 0.05).abs().max()) <= 1e-6


@pytest.mark.parametrize("orig,new,T", [(16000, 8000, 16000), (8000, 16000, 8000), (16000, 12000, 1001), (44100, 16000, 4410), (16000, 22050, 640)])
def test_resample_vs_float64_restatement(orig, new, T):
    rng = np.random.default_rng(orig + new)
    x = (0.1 * rng.standard_normal((2, 1, T))).astype(np.float32)
    got = E.resample_waveform(cu(x), orig, new)
    ref = OF.resample(x, orig, new)
    assert err(got, ref) <= 2e-5
    # a tone well inside both bands survives the round trip
    t = np.arange(T) / orig
    tone = (0.5 * np.sin(2 * np.pi * 440.0 * t)).astype(np.float32)[None, None]
    back = E.resample_waveform(E.resample_waveform(cu(tone), orig, new), new, orig).cpu().numpy()[..., :T]
    mid = slice(T // 4, 3 * T // 4)
    assert np.abs(back[..., mid] - tone[..., : back.shape[-1]][..., mid]).max() <= 2e-2


def test_effect_wrappers_keep_the_reference_conventions():
    x = torch.randn(2, 1, 4000, device="cuda") * 0.1
    mask = torch.ones_like(x)
    y, m = E.AudioEffects.lowpass_filter(x, cutoff_freq=3000, sample_rate=16000, mask=mask)
    assert y.shape == x.shape and m is mask
    assert torch.equal(y, E.lowpass(x, 3000 / 8000))                   # cutoff / NYQUIST, as the reference passes it to julius
    y, _ = E.AudioEffects.lowpass_filter(x, cutoff_freq=5000)           # 0.625 cycles per sample: the library raises, the wrapper passes x through
    assert y is x
    y, _ = E.AudioEffects.highpass_filter(x, cutoff_freq=500)
    assert torch.equal(y, E.highpass(x, 500 / 8000))
    with pytest.raises(ValueError, match="must be less than"):
        E.AudioEffects.bandpass_filter(x, 4000, 300)
    with pytest.raises(ValueError, match="above 0.5"):
        E.AudioEffects.bandpass_filter(x, 300, 5000)                     # the reference re-raises ValueError for band-pass
    with pytest.raises(ValueError, match="positive int"):
        E.AudioEffects.resample(x, 0)
    y, _ = E.AudioEffects.resample(x, 8000, sample_rate=16000)
    assert y.shape == x.shape
    y, m2 = E.apply_effect("resample", {"new_sample_rate": 12000}, x, mask)
    assert y.shape == x.shape and m2 is mask
    assert E.apply_effect("identity", {}, x, mask)[0] is x
    with pytest.raises(NotImplementedError):
        E.apply_effect("mp3_lossy_compression", {}, x, mask)
    with pytest.raises(RuntimeError, match="GPU"):
        E.lowpass(torch.zeros(1, 1, 100), 0.1)
