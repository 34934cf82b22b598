"""The detector / locator training step on the GPU (waveverify_amd.train.EncoderNetTrainer): loss and EVERY
parameter gradient of the whole net against (a) the reference's own autograd through whole (shrunk) Locator / Detector modules
(tests/golden/netgrads_*.npz) and (b) the differentiable float64 oracle (oracle/wv_oracle_train_torch.py, pinned to (a)) on the
full-size nets at 16000-sample clips; then optimizer steps against torch's AdamW driven by the oracle's gradients."""
import ast
import os

import numpy as np
import pytest
import torch

from oracle import wv_oracle_train_torch as OTT
from waveverify_amd.config import default_config
from waveverify_amd.init import random_state_dict

pytestmark = pytest.mark.gpu


def _cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def check_grads(tr, ref_grads, tol=2e-4, loose=None):
    """Every gradient tensor within tol of its own largest magnitude.  Scalars (FiLM gamma / beta biases, scale parameters) are sums
    over a whole activation tensor whose terms cancel: they are held to the largest scalar gradient of the net instead of to themselves."""
    assert sorted(tr.gviews) == sorted(ref_grads)
    scalar_scale = max([float(np.abs(r).max()) for r in ref_grads.values() if r.size <= 4] + [1e-30])
    worst = ("", 0.0)
    for k, r in ref_grads.items():
        got = tr.gviews[k].detach().cpu().numpy().astype(np.float64)
        assert np.isfinite(got).all(), k
        scale = max(float(np.abs(r).max()), scalar_scale if r.size <= 4 else 0.0, 1e-30)
        e = float(np.abs(got - r.reshape(got.shape)).max() / scale)
        if loose and any(t in k for t in loose[0]):
            assert e <= loose[1], (k, e)
            continue
        if e > worst[1]:
            worst = (k, e)
    assert worst[1] <= tol, worst
    return worst


@pytest.mark.parametrize("name", ["locator", "detector"])
def test_whole_net_gradients_vs_reference_autograd(golden_dir, name):
    from waveverify_amd.train import EncoderNetTrainer, bce_logits
    g = np.load(os.path.join(golden_dir, f"netgrads_{name}.npz"))
    c = ast.literal_eval(str(g["cfg"][0]))
    seed, kind = c.pop("seed"), c.pop("kind")
    cfg = default_config(kind, **c)
    tr = EncoderNetTrainer(cfg, random_state_dict(cfg, seed, parametrized=True))
    logits = tr.forward(_cu(g["x"]))
    assert float(np.abs(logits.cpu().numpy() - g["logits"]).max()) <= 5e-5 * max(1.0, float(np.abs(g["logits"]).max()))
    loss, dz = bce_logits(logits, _cu(g["mask"]), _cu(g["msg"]) if "msg" in g else None)
    assert abs(float(loss.item()) - float(g["loss"])) <= 1e-5 * float(g["loss"])
    tr.backward(dz)
    check_grads(tr, {k[2:]: g[k] for k in g.files if k.startswith("g:")})


@pytest.mark.parametrize("kind,B", [("locator", 3), ("detector", 2)])
def test_full_size_net_gradients_vs_oracle_and_training_steps(kind, B):
    """Default (full-size) nets, 1 s clips: every gradient vs the float64 oracle; then two optimizer steps tracked against
    torch.optim.AdamW fed with the oracle's gradients of the evolving parameters (clip_grad_norm_ included)."""
    from waveverify_amd.train import EncoderNetTrainer
    cfg = default_config(kind)
    sd = random_state_dict(cfg, 0, parametrized=True)
    rng = np.random.default_rng(11)
    T = 16000
    x = (0.1 * rng.standard_normal((B, 1, T))).astype(np.float32)
    mask = (rng.random((B, 1, T)) < 0.7).astype(np.float32)
    msg = rng.integers(0, 2, (B, cfg.nbits)).astype(np.float32) if kind == "detector" else None
    tr = EncoderNetTrainer(cfg, sd, lr=1e-3, max_norm=1.0)
    keys = list(tr.params)
    ref_p = {k: torch.nn.Parameter(torch.from_numpy(np.asarray(sd[k], dtype=np.float32).copy())) for k in keys}
    opt = torch.optim.AdamW(list(ref_p.values()), lr=1e-3, betas=(0.8, 0.99))
    sched = torch.optim.lr_scheduler.ExponentialLR(opt, 0.999996)
    losses = []
    for it in range(2):
        cur = dict(sd)
        cur.update({k: p.detach().numpy() for k, p in ref_p.items()})
        ref_loss, _, ref_grads, _ = OTT.loss_and_grads(cfg, cur, x, mask, msg)
        # gradients of the CURRENT parameters, before the step
        logits = tr.forward(_cu(x))
        from waveverify_amd.train import bce_logits
        loss, dz = bce_logits(logits, _cu(mask), None if msg is None else _cu(msg))
        tr.backward(dz)
        assert abs(float(loss.item()) - ref_loss) <= 2e-5 * ref_loss, (it, float(loss.item()), ref_loss)
        worst = check_grads(tr, ref_grads, tol=5e-4)
        # the step itself (forward + backward again inside; deterministic kernels -> the same gradients)
        loss2, norm = tr.step(_cu(x), _cu(mask), None if msg is None else _cu(msg))
        assert float(loss2.item()) == float(loss.item())
        for k, p in ref_p.items():
            p.grad = torch.from_numpy(ref_grads[k].astype(np.float32)).view_as(p)
        ref_norm = torch.nn.utils.clip_grad_norm_(list(ref_p.values()), 1.0)
        opt.step(); sched.step()
        assert abs(float(norm.item()) - float(ref_norm)) <= 5e-4 * float(ref_norm), (it, worst)
        for k, p in ref_p.items():
            d = float((tr.params[k].detach().cpu() - p.detach()).abs().max())
            assert d <= 2e-5 + 5e-3 * 1e-3, (it, k, d)                       # AdamW normalises: a step is <= lr per element
        losses.append(float(loss.item()))
    assert all(np.isfinite(losses))


# ---- generator -----------------------------------------------------------------------------------------------------------------
def test_film_mlp_vs_torch_modules():
    """Message MLP + FiLM heads + the modulation itself against the torch modules the reference builds (nn.Linear chains,
    seanet.py:518-550,831-846), float64 autograd."""
    from waveverify_amd.train import FilmMlp
    cfg = default_config("generator")
    sd = random_state_dict(cfg, 3, parametrized=True)
    fm = FilmMlp(cfg)
    params = {k: _cu(np.asarray(sd[k], np.float32)) for k in fm.keys}
    gviews = {k: torch.zeros_like(v) for k, v in params.items()}
    rng = np.random.default_rng(0)
    B, C, T, s = 5, 128, 64, 1
    msg = rng.integers(0, 2, (B, cfg.msg_dimension)).astype(np.float32)
    x = rng.standard_normal((B, C, T)).astype(np.float32)
    dy = rng.standard_normal((B, C, T)).astype(np.float32)
    leaf = {k: torch.tensor(np.asarray(sd[k]), dtype=torch.float64, requires_grad=True) for k in fm.keys}
    e = torch.nn.functional.linear(torch.from_numpy(msg).double(), leaf["encoder.msg_embedding.0.weight"], leaf["encoder.msg_embedding.0.bias"])
    for i in range(cfg.embedding_layers):
        e = torch.relu(torch.nn.functional.linear(e, leaf[f"encoder.msg_embedding.{1 + 2 * i}.weight"], leaf[f"encoder.msg_embedding.{1 + 2 * i}.bias"]))
    xt = torch.from_numpy(x).double().requires_grad_(True)
    bw, bands = C // cfg.freq_bands, []
    for b in range(cfg.freq_bands):
        ga = torch.nn.functional.linear(e, leaf[f"encoder.film_layers.{s}.{b}.gamma_layer.weight"], leaf[f"encoder.film_layers.{s}.{b}.gamma_layer.bias"]).unsqueeze(-1)
        be = torch.nn.functional.linear(e, leaf[f"encoder.film_layers.{s}.{b}.beta_layer.weight"], leaf[f"encoder.film_layers.{s}.{b}.beta_layer.bias"]).unsqueeze(-1)
        bands.append(xt[:, b * bw:(b + 1) * bw] * ga + be)
    y = torch.cat(bands, 1)
    y.backward(torch.from_numpy(dy).double())
    film = fm.forward(_cu(msg), params)
    got = fm.apply(_cu(x), film, s)
    assert float((got.cpu().double() - y.detach()).abs().max()) <= 2e-5 * float(y.detach().abs().max())
    dfilm = torch.zeros_like(film)
    dx = fm.apply_backward(_cu(x), film, _cu(dy), dfilm, s)
    assert float((dx.cpu().double() - xt.grad).abs().max()) <= 2e-5 * float(xt.grad.abs().max())
    fm.backward(dfilm, gviews)
    for k in fm.keys:
        r = leaf[k].grad
        got_k = gviews[k].cpu().double()
        if r is None:                                      # the other scales' heads: untouched by this loss
            assert float(got_k.abs().max()) == 0.0, k
        else:
            assert float((got_k - r).abs().max()) <= 1e-4 * max(float(r.abs().max()), 1e-12), k


def test_generator_gradients_vs_reference_autograd(golden_dir):
    from waveverify_amd.train import GeneratorTrainer
    g = np.load(os.path.join(golden_dir, "netgrads_generator.npz"))
    c = ast.literal_eval(str(g["cfg"][0]))
    seed, kind = c.pop("seed"), c.pop("kind")
    cfg = default_config(kind, **c)
    tr = GeneratorTrainer(cfg, random_state_dict(cfg, seed, parametrized=True))
    wm = tr.forward(_cu(g["x"]), _cu(g["msg"]))
    assert float(np.abs(wm.cpu().numpy() - g["wm"]).max()) <= 2e-5
    d_wm = 2.0 * (wm - _cu(g["target"])) / wm.numel()
    tr.backward(d_wm)
    check_grads(tr, {k[2:]: g[k] for k in g.files if k.startswith("g:")}, tol=2e-3)      # the fixture itself is float32 autograd


def test_full_size_generator_gradients_vs_oracle_and_a_step():
    """The default generator at 2 x 0.5 s (the float64 CPU oracle sets the test's duration): watermarked audio and every gradient vs the float64 oracle, one
    optimizer step vs torch's AdamW on the oracle's gradients."""
    from waveverify_amd.train import GeneratorTrainer
    cfg = default_config("generator")
    sd = random_state_dict(cfg, 0, parametrized=True)
    rng = np.random.default_rng(21)
    B, T = 2, 8000
    x = (0.1 * rng.standard_normal((B, 1, T))).astype(np.float32)
    msg = rng.integers(0, 2, (B, cfg.nbits)).astype(np.float32)
    target = (x + 0.01 * rng.standard_normal((B, 1, T))).astype(np.float32)
    ref_loss, ref_wm, ref_grads, ref_dx = OTT.generator_loss_and_grads(cfg, sd, x, msg, target)
    tr = GeneratorTrainer(cfg, sd, lr=1e-3, max_norm=1.0)
    wm = tr.forward(_cu(x), _cu(msg))
    assert float(np.abs(wm.cpu().numpy() - ref_wm).max()) <= 2e-5
    tr.backward(2.0 * (wm - _cu(target)) / wm.numel())
    check_grads(tr, ref_grads, tol=5e-4)
    keys = list(tr.params)
    ref_p = {k: torch.nn.Parameter(torch.from_numpy(np.asarray(sd[k], dtype=np.float32).copy())) for k in keys}
    opt = torch.optim.AdamW(list(ref_p.values()), lr=1e-3, betas=(0.8, 0.99))
    for k, p in ref_p.items():
        p.grad = torch.from_numpy(ref_grads[k].astype(np.float32)).view_as(p)
    ref_norm = torch.nn.utils.clip_grad_norm_(list(ref_p.values()), 1.0)
    opt.step()
    norm = tr.apply_gradients()
    assert abs(float(norm.item()) - float(ref_norm)) <= 5e-4 * float(ref_norm)
    for k, p in ref_p.items():
        assert float((tr.params[k].detach().cpu() - p.detach()).abs().max()) <= 2e-5 + 5e-6, k


# ---- gradients towards the audio, and the generator trained THROUGH the detector ----------------------------------------------------
@pytest.mark.parametrize("n_fft,hop,T", [(64, 1, 1000), (128, 2, 1000), (256, 8, 4000), (1024, 320, 16000), (64, 4, 37)])
def test_stft_feature_backward_vs_torch_autograd(n_fft, hop, T):
    """d<dP, P(wav)>/dwav against torch autograd of the reference's formula (conv1d with the windowed DFT basis, sqrt(clamp), log(clamp),
    normalise; conv.py:1055-1078, seanet.py:484-494) in float64, with the pinned basis."""
    from oracle import wv_oracle as O
    from waveverify_amd.train import StftFeatures
    import torch.nn.functional as F
    rng = np.random.default_rng(n_fft + hop)
    B = 3
    wav = (0.1 * rng.standard_normal((B, 1, T))).astype(np.float32)
    wav[1, 0, : T // 3] = 0.0                                       # silence: the clamped region passes no gradient
    st = StftFeatures(n_fft, hop, -4.3, 2.8)
    P = st(_cu(wav))
    dP = rng.standard_normal(tuple(P.shape)).astype(np.float32)
    w = torch.from_numpy(wav).double().requires_grad_(True)
    basis = torch.from_numpy(O.dft_basis(n_fft)).double()[:, None, :]
    c = F.conv1d(F.pad(w, (n_fft - 1, 0)), basis, None, stride=hop)
    Fq = n_fft // 2 + 1
    y = (((c[:, :Fq] ** 2 + c[:, Fq:] ** 2).clamp_min(1e-12).sqrt().clamp_min(1e-5).log()) - (-4.3)) / 2.8
    y.backward(torch.from_numpy(dP).double())
    dw = torch.zeros(B, 1, T, device="cuda")
    st.backward(_cu(wav), _cu(dP), dw, accumulate=False)
    ref = w.grad.numpy()
    assert float(np.abs(dw.cpu().numpy() - ref).max()) <= 2e-4 * float(np.abs(ref).max())
    st.backward(_cu(wav), _cu(dP), dw, accumulate=True)             # accumulates
    assert float(np.abs(dw.cpu().numpy() - 2 * ref).max()) <= 4e-4 * float(np.abs(ref).max())


def test_generator_trained_through_the_detector_vs_oracle():
    """The watermarking objective in miniature: wm = G(x, msg) + x, DecodingLoss(D(wm), mask, msg).  The detector's backward hands
    dL/d(wm) -- through conv_pre AND all five spectrogram branches -- to the generator's backward; every gradient of BOTH nets against
    the float64 oracle over the two nets."""
    from waveverify_amd.train import EncoderNetTrainer, GeneratorTrainer, bce_logits
    # half-width nets (same depth, strides, scales): the two-net float64 CPU oracle sets the duration; full-size nets are covered one by one above
    cfgG, cfgD = default_config("generator", channels_enc=32, channels_dec=48), default_config("detector", channels_enc=32)
    sdG, sdD = random_state_dict(cfgG, 0, parametrized=True), random_state_dict(cfgD, 0, parametrized=True)
    rng = np.random.default_rng(5)
    B, T = 2, 8000
    x = (0.1 * rng.standard_normal((B, 1, T))).astype(np.float32)
    msg = rng.integers(0, 2, (B, cfgG.nbits)).astype(np.float32)
    mask = (rng.random((B, 1, T)) < 0.8).astype(np.float32)
    ref_loss, gG, gD, ref_dwm = OTT.joint_decoding_loss_and_grads(cfgG, sdG, cfgD, sdD, x, msg, mask)
    G, D = GeneratorTrainer(cfgG, sdG), EncoderNetTrainer(cfgD, sdD)
    wm = G.forward(_cu(x), _cu(msg))
    logits = D.forward(wm)
    loss, dz = bce_logits(logits, _cu(mask), _cu(msg))
    assert abs(float(loss.item()) - ref_loss) <= 2e-5 * ref_loss
    d_wm = D.backward(dz, need_dx=True)
    # conv_pre + five spectrogram branches; d log|STFT| = re / |STFT|^2 amplifies the f32 rounding of quiet bins: 5e-3 of the peak
    assert float(np.abs(d_wm.cpu().numpy() - ref_dwm).max()) <= 5e-3 * float(np.abs(ref_dwm).max())
    G.backward(d_wm)
    check_grads(D, gD, tol=5e-4)
    # FiLM / message-MLP gradients are sums of dL/dy over whole activation tensors whose terms cancel: the error of dL/d(wm) shows there
    check_grads(G, gG, tol=1e-2, loose=(("film_layers", "msg_embedding"), 1e-1))
    # the same generator backward fed with the oracle's exact dL/d(wm): the strict bar again
    G.forward(_cu(x), _cu(msg))
    G.backward(_cu(ref_dwm.astype(np.float32)))
    check_grads(G, gG, tol=1e-3)


def test_watermark_step_vs_oracle():
    """One whole generator-update step of the reference's loop for the losses on this path: G -> augmentation -> D, L -> weighted BCE +
    waveform losses -> backward through D, L, the augmentation's select and G.  Losses and the gradients of all three nets against the
    float64 oracle composed of the same pieces (the augmentation as a differentiable torch select with the SAME plan)."""
    from waveverify_amd.train import WatermarkTrainer
    # half-width generator / detector (see above), the default locator
    cfgs = [default_config("generator", channels_enc=32, channels_dec=48), default_config("detector", channels_enc=32), default_config("locator")]
    sds = [random_state_dict(c, 0, parametrized=True) for c in cfgs]
    rng = np.random.default_rng(8)
    B, T = 2, 8000
    x = (0.1 * rng.standard_normal((B, 1, T))).astype(np.float32)
    msg = rng.integers(0, 2, (B, 16)).astype(np.float32)
    tr = WatermarkTrainer(cfgs[0], sds[0], cfgs[1], sds[1], cfgs[2], sds[2], lr=1e-4)
    # gradients BEFORE the optimizer moves the parameters: take them from the arenas right after step() (AdamW does not touch .grads)
    np.random.seed(4); torch.manual_seed(4)
    out = tr.step(_cu(x), _cu(msg))
    plan, seg_len, sm, _ = tr.aug.last
    ref, gG, gD, gL = OTT.watermark_step_loss_and_grads(cfgs[0], sds[0], cfgs[1], sds[1], cfgs[2], sds[2], x, msg, plan, seg_len,
                                                         (sm.mode, sm.a, sm.b, sm.c, sm.perm, sm.t_out), tr.lambdas)
    for k in ("dec/loss", "loc/loss", "waveform/loss", "loss"):
        assert abs(float(out[k].item()) - ref[k]) <= 5e-5 * abs(ref[k]), (k, float(out[k].item()), ref[k])
    st = out["stats"]          # merged as the reference merges them: the sequence stats' 'unchanged' overrides the localisation one
    assert abs(st["original_revert"] + st["zero_replace"] + st["cross_substitute"] - 20.0) < 1e-9      # 1 of the 5 segments of every clip
    check_grads(tr.D, gD, tol=1e-3)
    check_grads(tr.L, gL, tol=1e-3)
    check_grads(tr.G, gG, tol=2e-2, loose=(("film_layers", "msg_embedding"), 2e-1))
    # the parameters moved, and by no more than one AdamW step (lr per element, + decay)
    for net, sd in zip((tr.G, tr.D, tr.L), sds):
        for k in list(net.params)[:20]:
            d = float((net.params[k].cpu() - torch.from_numpy(np.asarray(sd[k], np.float32))).abs().max())
            assert 0.0 < d <= 1.2e-4 + 1e-6 * float(np.abs(sd[k]).max()), (k, d)


def test_watermark_training_learns():
    """Twelve generator-update steps on a fixed batch: the decoding and localisation losses fall, every arena stays finite, the run is
    bitwise repeatable under the same seeds (deterministic kernels, reference RNG order for the augmentation plan)."""
    from waveverify_amd.train import WatermarkTrainer
    cfgs = [default_config(k) for k in ("generator", "detector", "locator")]
    sds = [random_state_dict(c, 0, parametrized=True) for c in cfgs]
    rng = np.random.default_rng(2)
    x = (0.1 * rng.standard_normal((8, 1, 16000))).astype(np.float32)
    msg = rng.integers(0, 2, (8, 16)).astype(np.float32)

    def run():
        tr = WatermarkTrainer(cfgs[0], sds[0], cfgs[1], sds[1], cfgs[2], sds[2], lr=5e-4)
        np.random.seed(0); torch.manual_seed(0)
        hist = [tr.step(_cu(x), _cu(msg)) for _ in range(12)]
        return tr, [(float(h["dec/loss"].item()), float(h["loc/loss"].item()), float(h["waveform/loss"].item())) for h in hist]
    a, la = run()
    b, lb = run()
    assert la == lb and torch.equal(a.G.arena, b.G.arena) and torch.equal(a.D.arena, b.D.arena)
    assert all(np.isfinite(v) for row in la for v in row)
    assert bool(torch.isfinite(a.G.arena).all() and torch.isfinite(a.D.arena).all() and torch.isfinite(a.L.arena).all())
    first, last = np.mean(la[:3], axis=0), np.mean(la[-3:], axis=0)
    assert last[0] < first[0] and last[1] < first[1], (first, last)


def test_watermark_step_with_effect_scheduler():
    """The adaptive effect scheduler inside the step (watermarking.py:521-612,697-752): effects are selected for the first clips with
    the reference's RNG call, a caller-supplied straight-through effect runs, per-clip BER / mIoU are fed back; an unsupported effect
    without a callable is refused."""
    from waveverify_amd.effect_scheduler import EffectScheduler
    from waveverify_amd.metrics import BER, MIOU
    from waveverify_amd.train import WatermarkTrainer
    cfgs = [default_config(k) for k in ("generator", "detector", "locator")]
    sds = [random_state_dict(c, 0, parametrized=True) for c in cfgs]
    rng = np.random.default_rng(3)
    x = (0.1 * rng.standard_normal((4, 1, 16000))).astype(np.float32)
    msg = rng.integers(0, 2, (4, 16)).astype(np.float32)
    calls = []

    def halve(name, params, audio, mask):
        calls.append((name, dict(params)))
        return audio * params["factor"], mask

    sched = EffectScheduler({"identity": {}, "amplitude_scaling": {"factor": {"choices": [0.5, 0.8]}}})
    tr = WatermarkTrainer(cfgs[0], sds[0], cfgs[1], sds[1], cfgs[2], sds[2], effect_scheduler=sched, apply_effect=halve)
    np.random.seed(1); torch.manual_seed(1)
    out = tr.step(_cu(x), _cu(msg))
    applied = out["stats"]["selected_effects"]
    assert len(applied) == 2 and tr.effect_update_count == 2             # capped at the number of known effects (watermarking.py:537 quirk)
    assert [c[0] for c in calls] == [str(n) for n, _ in applied if str(n) != "identity"]
    st = sched.get_effect_statistics()
    assert sum(v["selection_count"] for v in st.values()) == 2
    for name, _ in applied:
        assert st[str(name)]["ema_ber"] is not None and 0.0 <= st[str(name)]["ema_ber"] <= 1.0 and 0.0 <= st[str(name)]["ema_miou"] <= 1.0
    assert np.isfinite(float(out["loss"].item()))
    bad = WatermarkTrainer(cfgs[0], sds[0], cfgs[1], sds[1], cfgs[2], sds[2], effect_scheduler=EffectScheduler({"lowpass_filter": {"cutoff_freq": 3000}}))
    with pytest.raises(NotImplementedError, match="lowpass_filter"):
        bad.step(_cu(x), _cu(msg))


def test_watermark_step_with_gpu_effects():
    """The scheduler's sinc-filter / resample effects run on the GPU inside the step (waveverify_amd.effects.apply_effect as the hook)."""
    from waveverify_amd.effect_scheduler import EffectScheduler
    from waveverify_amd.effects import apply_effect
    from waveverify_amd.train import WatermarkTrainer
    cfgs = [default_config(k) for k in ("generator", "detector", "locator")]
    sds = [random_state_dict(c, 0, parametrized=True) for c in cfgs]
    rng = np.random.default_rng(6)
    x = (0.1 * rng.standard_normal((4, 1, 16000))).astype(np.float32)
    msg = rng.integers(0, 2, (4, 16)).astype(np.float32)
    grid = {"identity": {}, "lowpass_filter": {"cutoff_freq": {"choices": [3000, 2000]}}, "highpass_filter": {"cutoff_freq": {"choices": [100, 500]}},
            "resample": {"new_sample_rate": {"choices": [8000, 12000]}}}
    sched = EffectScheduler(grid)
    tr = WatermarkTrainer(cfgs[0], sds[0], cfgs[1], sds[1], cfgs[2], sds[2], effect_scheduler=sched, apply_effect=apply_effect)
    np.random.seed(2); torch.manual_seed(2)
    outs = [tr.step(_cu(x), _cu(msg)) for _ in range(2)]
    assert all(np.isfinite(float(o["loss"].item())) for o in outs)
    assert tr.effect_update_count == 8 and sum(v["selection_count"] for v in sched.get_effect_statistics().values()) == 8


def test_trained_nets_round_trip_through_the_reference_checkpoint_format(tmp_path):
    """Train three steps -> save_checkpoint (atomic format, weight norm stripped: scripts/train.py:1589-1676) -> WaveVerify(path)
    (waveverify/core.py:324-426) -> embed / detect / locate agree with the trainers' own forward passes.  The state dicts carry every
    key of the reference's modules: the DFT buffers and the detector's / locator's unused message MLP + FiLM tensors included.
    (Not bit for bit: the SpecBlock's scalar rides in the GEMM operand of the training unit and in the epilogue of the inference
    unit -- same function, one rounding apart.)"""
    from waveverify_amd import WaveVerify
    from waveverify_amd.checkpoint import load_checkpoint
    from waveverify_amd.params import param_specs
    from waveverify_amd.train import WatermarkTrainer
    cfgs = [default_config(k) for k in ("generator", "detector", "locator")]
    sds = [random_state_dict(c, 0, parametrized=True) for c in cfgs]
    tr = WatermarkTrainer(cfgs[0], sds[0], cfgs[1], sds[1], cfgs[2], sds[2], lr=5e-4)
    rng = np.random.default_rng(5)
    x = _cu((0.1 * rng.standard_normal((4, 1, 16000))).astype(np.float32))
    msg = _cu(rng.integers(0, 2, (4, 16)).astype(np.float32))
    np.random.seed(0); torch.manual_seed(0)
    for _ in range(3):
        tr.step(x, msg)
    path = tr.save_checkpoint(tmp_path, "best")
    assert path.name == "best.pth" and not (tmp_path / "best.tmp").exists()
    ck = torch.load(str(path), map_location="cpu", weights_only=True)           # tensors and plain containers only
    assert ck["step"] == 3 and set(ck["models"]) == {"generator", "detector", "locator"}
    for kind, cfg in zip(("generator", "detector", "locator"), cfgs):
        keys = {k for k, _, _ in param_specs(cfg)} | {f"encoder.spec_blocks.{s}.spec.weight" for s in range(len(cfg.strides))} | {"encoder.spec_post.spec.weight"}
        assert set(ck["models"][kind]) == keys, kind
        assert not any("parametrizations" in k for k in ck["models"][kind])
    # the trained weights differ from the initial ones, the untouched FiLM tensors of the detector do not
    w0 = random_state_dict(cfgs[1], 0)
    assert not np.array_equal(ck["models"]["detector"]["encoder.conv_post.2.conv.conv.weight"].numpy(), w0["encoder.conv_post.2.conv.conv.weight"])
    assert np.array_equal(ck["models"]["detector"]["encoder.film_layers.0.0.gamma_layer.weight"].numpy(), w0["encoder.film_layers.0.0.gamma_layer.weight"])
    sds2, cfgs2 = load_checkpoint(tmp_path)
    assert cfgs2["generator"].to_dict() == cfgs[0].to_dict()
    wv = WaveVerify(str(tmp_path))
    wm_t = tr.G.forward(x, msg)
    wm_i = wv.embed_batch(x, msg)
    assert float((wm_t - wm_i).abs().max()) <= 2e-6
    bits, mp = wv.detect_batch(wm_t)
    mp_t = torch.sigmoid(tr.D.forward(wm_t)).mean(-1)
    assert float((mp - mp_t).abs().max()) <= 2e-6
    loc_t = torch.sigmoid(tr.L.forward(wm_t))[:, 0]
    assert float((wv.locate_batch(wm_t) - loc_t).abs().max()) <= 2e-5
    # the live weight-norm layout round-trips too (host fold at load)
    p2 = tr.save_checkpoint(tmp_path / "live", "latest", parametrized=True)
    ck2 = torch.load(str(p2), map_location="cpu", weights_only=True)
    assert any(k.endswith("parametrizations.weight.original0") for k in ck2["models"]["generator"])
    wv2 = WaveVerify(str(tmp_path / "live"))
    assert float((wv2.embed_batch(x, msg) - wm_t).abs().max()) <= 2e-6


def test_one_message_for_the_whole_batch():
    """watermarking.py:320-329: a [1, nbits] (or shorter) message is repeated over the batch -- for G, the decoding loss and the metrics."""
    from waveverify_amd.train import WatermarkTrainer
    cfgs = [default_config(k) for k in ("generator", "detector", "locator")]
    sds = [random_state_dict(c, 0, parametrized=True) for c in cfgs]
    rng = np.random.default_rng(6)
    x = _cu((0.1 * rng.standard_normal((3, 1, 16000))).astype(np.float32))
    one = _cu(rng.integers(0, 2, (1, 16)).astype(np.float32))
    outs = []
    for m in (one, one.repeat(3, 1)):
        tr = WatermarkTrainer(cfgs[0], sds[0], cfgs[1], sds[1], cfgs[2], sds[2])
        np.random.seed(1); torch.manual_seed(1)
        o = tr.step(x, m)
        outs.append((float(o["dec/loss"].item()), float(o["loss"].item()), tr.G.arena.clone()))
    assert outs[0][0] == outs[1][0] and outs[0][1] == outs[1][1] and torch.equal(outs[0][2], outs[1][2])


def test_resume_from_a_checkpoint_continues_the_same_run(tmp_path):
    """save_checkpoint(parametrized=True) -> WatermarkTrainer.from_checkpoint: weights, AdamW moments and step count come back, and
    the next step equals the uninterrupted run's bit for bit; from the stripped layout (the reference's own files) the weights are
    put back under weight norm (original0 = ||w||, original1 = w) -- the same function, a fresh optimizer at the saved step."""
    from waveverify_amd.train import WatermarkTrainer
    small = dict(channels_enc=16, dimension=32)
    cfgs = [default_config("generator", channels_dec=16, n_residual_dec=1, **small), default_config("detector", **small),
            default_config("locator")]
    sds = [random_state_dict(c, 3, parametrized=True) for c in cfgs]
    rng = np.random.default_rng(9)
    x = _cu((0.1 * rng.standard_normal((2, 1, 16000))).astype(np.float32))
    msg = _cu(rng.integers(0, 2, (2, 16)).astype(np.float32))
    a = WatermarkTrainer(cfgs[0], sds[0], cfgs[1], sds[1], cfgs[2], sds[2], lr=5e-4)
    np.random.seed(0); torch.manual_seed(0)
    for _ in range(2):
        a.step(x, msg)
    a.save_checkpoint(tmp_path / "live", "latest", parametrized=True)
    a.save_checkpoint(tmp_path / "stripped", "latest")
    b = WatermarkTrainer.from_checkpoint(tmp_path / "live", lr=5e-4)
    assert b.G.opt.t == 2 and torch.equal(b.G.arena.cpu(), a.G.arena.cpu()) and torch.equal(b.D.opt.m.cpu(), a.D.opt.m.cpu())
    for tr in (a, b):
        np.random.seed(5); torch.manual_seed(5)
        tr.step(x, msg)
    assert torch.equal(a.G.arena, b.G.arena) and torch.equal(a.D.arena, b.D.arena) and torch.equal(a.L.arena, b.L.arena)
    c = WatermarkTrainer.from_checkpoint(tmp_path / "stripped", lr=5e-4)
    assert c.G.opt.t == 2 and float(c.G.opt.m.abs().max()) == 0.0
    wm_a = a.__class__.from_checkpoint(tmp_path / "live").G.forward(x, msg)        # the weights of step 2, live layout
    wm_c = c.G.forward(x, msg)
    assert float((wm_a - wm_c).abs().max()) <= 2e-6


def test_a_checkpoints_dft_bases_are_used_and_kept(tmp_path):
    """a checkpoint's `...spec.weight` tensors (learned when the reference trains with spec_learnable: true, conf/base.yml;
    modules/conv.py:1023) must reach the TRAINING forward the way they reach the inference nets, and survive save -> load unchanged.
    A perturbed basis through from_checkpoint: the trainers' forward equals WaveVerify(<same file>) (and differs from the analytic-basis
    result), and the saved file holds the perturbed tensors."""
    from waveverify_amd import WaveVerify
    from waveverify_amd.checkpoint import save_atomic_checkpoint, stft_basis, argbind_config
    from waveverify_amd.train import WatermarkTrainer
    kinds = ("generator", "detector", "locator")
    cfgs = {k: default_config(k) for k in kinds}
    sds = {k: {n: torch.from_numpy(v) for n, v in random_state_dict(cfgs[k], 0).items()} for k in kinds}
    g = torch.Generator().manual_seed(3)
    for k in kinds:
        S = len(cfgs[k].strides)
        for s in range(S + 1):
            key = ("encoder.spec_post" if s == S else f"encoder.spec_blocks.{s}") + ".spec.weight"
            b = stft_basis((2 ** s) * cfgs[k].n_fft_base)
            sds[k][key] = b + 0.02 * float(b.abs().max()) * torch.randn(b.shape, generator=g)
    save_atomic_checkpoint(tmp_path / "a", "best", sds, 0, argbind_config(cfgs))
    tr = WatermarkTrainer.from_checkpoint(tmp_path / "a")
    wv = WaveVerify(str(tmp_path / "a"))
    rng = np.random.default_rng(11)
    x = _cu((0.1 * rng.standard_normal((2, 1, 8000))).astype(np.float32))
    msg = _cu(rng.integers(0, 2, (2, 16)).astype(np.float32))
    wm_t, wm_i = tr.G.forward(x, msg), wv.embed_batch(x, msg)
    assert float((wm_t - wm_i).abs().max()) <= 2e-6
    mp_t = torch.sigmoid(tr.D.forward(wm_t)).mean(-1)
    assert float((wv.detect_batch(wm_t)[1] - mp_t).abs().max()) <= 2e-6
    # ... and the perturbation matters: the analytic-basis nets give something else
    plain = WatermarkTrainer(cfgs["generator"], random_state_dict(cfgs["generator"], 0, parametrized=True), cfgs["detector"],
                             random_state_dict(cfgs["detector"], 0, parametrized=True), cfgs["locator"], random_state_dict(cfgs["locator"], 0, parametrized=True))
    assert float((plain.G.forward(x, msg) - wm_t).abs().max()) > 1e-4
    tr.save_checkpoint(tmp_path / "b", "best")
    ck = torch.load(str(tmp_path / "b" / "best.pth"), map_location="cpu", weights_only=True)
    for k in kinds:
        for key, v in sds[k].items():
            if key.endswith("spec.weight"):
                assert torch.equal(ck["models"][k][key], v), (k, key)
    assert "optimizers" not in ck and "wv_amd_optimizers" in ck
