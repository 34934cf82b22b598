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


def _dot(a, b):
    return float((a.double() * b.double()).sum())


@pytest.mark.parametrize("name,params", [("lowpass_filter", {"cutoff_freq": 3000}), ("lowpass_filter", {"cutoff_freq": 100}),
                                         ("highpass_filter", {"cutoff_freq": 500}), ("bandpass_filter", {"cutoff_freq_low": 300, "cutoff_freq_high": 3500}),
                                         ("resample", {"new_sample_rate": 8000}), ("resample", {"new_sample_rate": 12000}),
                                         ("resample", {"new_sample_rate": 22050}), ("identity", {}),
                                         ("lowpass_filter", {"cutoff_freq": 5000})])
@pytest.mark.parametrize("T", [16000, 1001])
def test_effect_adjoints_are_the_transposed_operators(name, params, T):
    """The reference differentiates through its julius filters and torchaudio resampler (plain torch ops), so the generator's gradient is
    A^T d for the linear effect A.  <A x, d> == <x, A^T d> for random x, d pins apply_effect_backward to apply_effect on the device (the
    replicate padding's transpose, the crop / zero pad of the resample round trip and the pass-through cases included), and the
    float64 restatement's autograd gives the same gradient."""
    g = torch.Generator(device="cuda").manual_seed(T + len(name))
    x = torch.randn(2, 1, T, device="cuda", generator=g) * 0.1
    d = torch.randn(2, 1, T, device="cuda", generator=g)
    y, _ = E.apply_effect(name, params, x, None)
    dx = E.apply_effect_backward(name, params, d)
    assert dx.shape == x.shape
    lhs, rhs = _dot(y, d), _dot(x, dx)
    assert abs(lhs - rhs) <= 2e-5 * max(abs(lhs), float(y.norm() * d.norm()) * 1e-2, 1e-6), (lhs, rhs)
    if name in ("lowpass_filter", "highpass_filter") and params["cutoff_freq"] <= 4000:
        ref = OF.filter_gradient(name, x.cpu().numpy(), d.cpu().numpy(), params["cutoff_freq"] / 8000.0)
        assert err(dx, ref) <= 2e-5


def test_straight_through_effects_keep_the_identity_gradient():
    d = torch.randn(1, 1, 400, device="cuda")
    assert E.apply_effect_backward("mp3_lossy_compression", {}, d) is d


def test_trainer_routes_the_gradient_through_the_effect_adjoint():
    """WatermarkTrainer with effects.apply_effect / apply_effect_backward attached: the gradient that reaches the generator for a clip
    with a scheduled low-pass is the transposed filter of what the detector and locator returned (not the identity)."""
    from waveverify_amd.config import default_config
    from waveverify_amd.init import random_state_dict
    from waveverify_amd.train import WatermarkTrainer

    class OneEffect:                                                   # a scheduler that always picks one low-pass for clip 0
        def select_effects(self, batch_size):
            return [("lowpass_filter", {"cutoff_freq": 2000})]

        def update_effect_metrics(self, *a, **k):
            pass
    small = dict(channels_enc=8, dimension=16, strides=[2, 2], n_fft_base=16)
    cg = default_config("generator", channels_dec=8, n_residual_dec=1, **small)
    cd, cl = default_config("detector", output_dim=8, **small), default_config("locator", output_dim=8, **small)
    seen = {}

    def back(name, params, d_out):
        seen["in"] = d_out.clone()
        seen["out"] = E.apply_effect_backward(name, params, d_out)
        return seen["out"]
    tr = WatermarkTrainer(cg, random_state_dict(cg, 1, parametrized=True), cd, random_state_dict(cd, 1, parametrized=True), cl,
                          random_state_dict(cl, 1, parametrized=True), effect_scheduler=OneEffect(), apply_effect=E.apply_effect, effect_backward=back)
    got = {}
    g_back = tr.G.backward
    tr.G.backward = lambda d_wm, need_dx=False: (got.__setitem__("d_wm", d_wm.clone()), g_back(d_wm, need_dx))[1]
    x = torch.randn(2, 1, 1600, device="cuda") * 0.1
    msg = torch.randint(0, 2, (2, 16), device="cuda").float()
    out = tr.step(x, msg, augment=False)
    assert np.isfinite(float(out["loss"].item()))
    wav_grad = got["d_wm"] - 0                                          # d_wm = effect^T (dD + dL) + d(waveform loss)
    _, d_wav = __import__("waveverify_amd.train", fromlist=["l1_loss"]).l1_loss(tr.G.forward(x, msg) * 0 + x, x, grad_scale=1.0)
    assert not torch.equal(seen["in"], seen["out"])                     # the adjoint is not the identity
    assert torch.equal(seen["out"], E.lowpass_adjoint(seen["in"], 2000 / 8000))
    # clip 0 went through the adjoint, clip 1 (no effect) did not: removing the effect's share leaves the same waveform-loss gradient form
    assert float((wav_grad[0] - seen["out"][0]).abs().max()) <= float(tr.lambdas["waveform/loss"]) / x[0].numel() + 1e-6
