"""The numpy oracle (oracle/wv_oracle.py) against outputs of the reference itself
(tests/golden/*.npz, produced by tests/golden/make_golden.py in the build container).
CPU only.  Tolerances: float32 nets, different summation order -> 2e-5 abs on O(1) tensors."""
import os

import numpy as np
import pytest

from oracle import wv_oracle as O
from waveverify_amd.config import default_config
from waveverify_amd.init import random_state_dict

TOL = 2e-5


def _close(a, b, tol=TOL, what=""):
    a = np.asarray(a); b = np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = float(np.abs(a - b).max()) if a.size else 0.0
    tol = tol * max(1.0, float(np.abs(b).max()) if b.size else 1.0)   # scale-aware abs tolerance
    assert err <= tol, f"{what}: max|d|={err:.3e} > {tol:.3e}"


@pytest.fixture(scope="module")
def full_nets():
    cfgs = {k: default_config(k) for k in ("generator", "detector", "locator")}
    return {k: O._Net(c, random_state_dict(c, 0, parametrized=True)) for k, c in cfgs.items()}


@pytest.mark.parametrize("T", [16000, 16001, 4800])
def test_full_size_against_reference(golden_dir, full_nets, T):
    g = np.load(os.path.join(golden_dir, f"full_T{T}.npz"))
    x, msg = g["x"], g["msg"]
    taps = {}
    G = full_nets["generator"]
    delta = O.generator_forward(G.cfg, G, x, msg, taps)
    _close(taps["conv_pre"][..., ::97], g["conv_pre_sub"], what="conv_pre")
    _close(taps["enc_scale0_down"][..., ::53], g["enc0_sub"], 5e-5, "enc0")
    _close(taps["enc_scale3_down"][:, ::8], g["enc3"], 5e-5, "enc3")
    _close(taps["latent"], g["latent"], 5e-5, "latent")
    _close(taps["dec_scale0_out"][:, ::8], g["dec_up0_sub"], 5e-5, "dec_up0")
    _close(delta, g["delta"], 1e-5, "delta")
    wm = (delta + x).astype(np.float32)
    _close(wm, g["wm"], 1e-5, "wm")
    D = full_nets["detector"]
    logits = O.detector_forward(D.cfg, D, g["wm"])
    _close(logits[..., ::37], g["det_logits_sub"], 1e-4, "det logits")
    mp = O.mean_probabilities(logits)
    _close(mp, g["det_mean_prob"], 1e-5, "mean prob")
    assert (O.decide_bits(mp) == g["det_bits"]).all()
    assert float(g["det_margin"]) > 0.1          # decisions are far from the 0.5 threshold
    L = full_nets["locator"]
    ll = O.locator_forward(L.cfg, L, g["wm"])
    _close(ll[..., ::7], g["loc_logits_sub"], 1e-4, "loc logits")


def test_speech_clip(golden_dir, full_nets):
    """BASELINE.json configs[0]: real 1 s / 16 kHz speech slices, 16-bit IDs 42, 0xBEEF, 0x1234."""
    g = np.load(os.path.join(golden_dir, "speech_T16000.npz"))
    G, D, L = (full_nets[k] for k in ("generator", "detector", "locator"))
    wm = O.embed(G.cfg, G, g["x"], g["msg"])
    _close(wm, g["wm"], 1e-5, "wm")
    mp = O.mean_probabilities(O.detector_forward(D.cfg, D, g["wm"]))
    _close(mp, g["det_mean_prob"], 1e-5)
    assert (O.decide_bits(mp) == g["det_bits"]).all()
    _close(O.locator_forward(L.cfg, L, g["wm"])[..., ::7], g["loc_logits_sub"], 1e-4)


SMALL = dict(channels_enc=8, dimension=16, strides=[2, 2], n_fft_base=16)


def small_cfgs():
    sg = default_config("generator", channels_dec=8, n_residual_dec=2, **SMALL)
    sd = default_config("detector", output_dim=8, nbits=16, **SMALL)
    sl = default_config("locator", **{**SMALL, "channels_enc": 4, "dimension": 8,
                                      "n_residual_enc": 1, "output_dim": 8})
    return sg, sd, sl


@pytest.mark.parametrize("T", [64, 67, 1])
def test_small_nets_every_tensor(golden_dir, T):
    g = np.load(os.path.join(golden_dir, f"small_T{T}.npz"))
    sg, sd, sl = small_cfgs()
    taps = {}
    delta = O.generator_forward(sg, random_state_dict(sg, 7, parametrized=True), g["x"], g["msg"], taps)
    _close(taps["conv_pre"], g["tap_conv_pre"], what="conv_pre")
    _close(taps["enc_scale0_down"], g["tap_down0"], what="down0")
    _close(taps["enc_scale1_down"], g["tap_down1"], what="down1")
    _close(taps["latent"], g["tap_latent"], what="latent")
    _close(taps["dec_scale0_out"], g["tap_dec_up0"], what="dec_up0")
    _close(delta, g["delta"], what="delta")
    assert delta.shape[-1] == T                       # generator.py:410 trims to the input length
    _close(O.detector_forward(sd, random_state_dict(sd, 7, parametrized=True), g["x"]),
           g["det_logits"], what="det")
    _close(O.locator_forward(sl, random_state_dict(sl, 7, parametrized=True), g["x"]),
           g["loc_logits"], what="loc")


def test_dilated_resblock(golden_dir):
    """dilation_base=2 exercises the d>1 padding math (conv.py:732, seanet.py:691)."""
    g = np.load(os.path.join(golden_dir, "small_dilated_T50.npz"))
    cfg = default_config("detector", output_dim=8, dilation_base=2, **{**SMALL, "n_residual_enc": 2})
    _close(O.detector_forward(cfg, random_state_dict(cfg, 11, parametrized=True), g["x"]),
           g["det_logits"], what="dilated det")


def test_dft_basis_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "dft_basis.npz"))
    _close(O.dft_basis(64), g["n64"], 2e-6, "basis 64")
    _close(O.dft_basis(1024)[::19], g["n1024_rows_every19"], 3e-4, "basis 1024")


def test_fold_matches_plain_state_dict():
    cfg = default_config("locator")
    a = O.fold_state_dict(random_state_dict(cfg, 3, parametrized=True))
    b = random_state_dict(cfg, 3, parametrized=False)
    assert set(a) == set(b)
    for k in a:
        _close(a[k], b[k], 1e-6, k)


def test_ber_and_miou_edge_cases():
    """scripts/evaluate.py:504-510 (no valid bits -> 0) and :640-653 (empty fg/bg -> IoU 1)."""
    rng = np.random.default_rng(0)
    bits = rng.integers(0, 2, (4, 16))
    logits = np.repeat(((bits * 2 - 1) * 2.0)[:, :, None], 100, axis=2).astype(np.float32)
    logits += 0.5 * rng.standard_normal(logits.shape).astype(np.float32)
    assert O.ber(logits, bits) == 0.0
    assert O.ber(-logits, bits) == 1.0
    half = np.zeros((4, 1, 100), np.float32); half[:, :, :50] = 1
    assert O.ber(logits, bits, half) == 0.0
    assert O.ber(logits, bits, np.zeros((4, 1, 100), np.float32)) == 0.0
    z = np.zeros(10, int); o = np.ones(10, int)
    assert O.miou(z, z) == 1.0 and O.miou(o, o) == 1.0 and O.miou(z, o) == 0.0
    m = np.array([1, 1, 0, 0]); n = np.array([1, 0, 0, 0])
    assert abs(O.miou(m, n) - (0.5 + 2 / 3) / 2) < 1e-12


def test_metrics_pinned_to_reference_classes(golden_dir):
    """BER / MIOU against outputs of the reference's own classes (scripts/evaluate.py:419-516, 575-665),
    generated by tests/golden/make_golden_metrics.py: the numpy oracle AND the shipped
    waveverify_amd.metrics restatement, incl. the edge cases at :504-510 and :640-653 and the inputs
    on which the reference raises RuntimeError (stored as NaN)."""
    import pytest
    import torch
    from waveverify_amd import metrics
    g = np.load(os.path.join(golden_dir, "metrics.npz"))
    n_raise = 0
    for i in range(int(g["n_ber"])):
        logits, bits, thr, want = g[f"ber{i}_logits"], g[f"ber{i}_bits"], float(g[f"ber{i}_thr"]), float(g[f"ber{i}_out"])
        mask = g[f"ber{i}_mask"] if f"ber{i}_mask" in g.files else None
        call = lambda: metrics.BER(threshold=thr)(torch.from_numpy(logits), torch.from_numpy(bits),
                                                  None if mask is None else torch.from_numpy(mask))
        if np.isnan(want):
            n_raise += 1
            with pytest.raises(RuntimeError):
                call()
            continue
        assert abs(float(call()) - want) < 1e-7, f"metrics.BER case {i}"
        assert abs(O.ber(logits, bits, mask, threshold=thr) - want) < 1e-7, f"oracle.ber case {i}"
    for i in range(int(g["n_miou"])):
        p, q, want = g[f"miou{i}_p"], g[f"miou{i}_g"], float(g[f"miou{i}_out"])
        args = (torch.from_numpy(p), torch.from_numpy(q)) if int(g[f"miou{i}_torch"]) else (p, q)
        if np.isnan(want):
            n_raise += 1
            with pytest.raises(RuntimeError):
                metrics.MIOU()(*args)
            continue
        assert abs(metrics.MIOU()(*args) - want) < 1e-12, f"metrics.MIOU case {i}"
        assert abs(O.miou(p, q) - want) < 1e-12, f"oracle.miou case {i}"
    assert n_raise == 4


def test_narrow_margin_detector_fixture(golden_dir):
    """The oracle on the narrow-margin detector fixture (tests/golden/make_golden_narrow.py): same bar as the
    wide-margin ones, and every bit is decidable (margin > 4 x error)."""
    g = np.load(os.path.join(golden_dir, "narrow_margin_T16000.npz"))
    cfg = default_config("detector")
    sd = random_state_dict(cfg, 0)
    sd["last_layer.bias"] = g["last_layer_bias"]
    mp = O.mean_probabilities(O.detector_forward(cfg, sd, g["wm"][:2]))
    err = float(np.abs(mp - g["det_mean_prob"][:2]).max())
    assert err <= 1e-5
    assert (g["margin"][:2] > 4 * max(err, 1e-7)).all()
    assert ((mp >= 0.5).astype(np.int32) == g["det_bits"][:2]).all()


def test_torch_flavoured_oracle_matches_reference(golden_dir):
    """oracle/wv_oracle_torch.py (what bench.py times as the CPU baseline) is pinned like the numpy one."""
    import torch
    from oracle import wv_oracle_torch as OT
    g = np.load(os.path.join(golden_dir, "full_T16001.npz"))
    cg, cd, cl = (default_config(k) for k in ("generator", "detector", "locator"))
    G = OT.Net(cg, random_state_dict(cg, 0, parametrized=True))
    D = OT.Net(cd, random_state_dict(cd, 0))
    L = OT.Net(cl, random_state_dict(cl, 0))
    wm = OT.embed(G, g["x"], g["msg"]).numpy()
    _close(wm, g["wm"], 1e-5, "wm (torch oracle)")
    lg = OT.detector_logits(D, g["wm"])
    _close(lg.numpy()[..., ::37], g["det_logits_sub"], 1e-4, "det logits (torch oracle)")
    mp = OT.mean_probabilities(lg).numpy()
    _close(mp, g["det_mean_prob"], 1e-5)
    assert ((mp >= 0.5).astype(int) == g["det_bits"]).all()
    _close(OT.detector_logits(L, g["wm"]).numpy()[..., ::7], g["loc_logits_sub"], 1e-4, "loc (torch oracle)")


@pytest.mark.parametrize("tag", ["small", "c64", "c96", "c160"])
def test_training_half_oracle_vs_reference_autograd(golden_dir, tag):
    """First training-step slice: forward + backward of a ResnetBlock half with live weight norm.
    oracle/wv_oracle_train.py against the reference modules' autograd (tests/golden/make_golden_grads.py)."""
    from oracle import wv_oracle_train as OT
    g = np.load(os.path.join(golden_dir, f"grads_half_{tag}.npz"))
    out = OT.half_backward(g["x"], float(g["pre_scale"]), g["g_pw"], g["v_pw"], g["g_dw"], g["v_dw"], g["b_dw"], g["dy"])
    for k in ("y", "dx", "dg_pw", "dv_pw", "dg_dw", "dv_dw", "db_dw"):
        ref = g[k]
        err = np.abs(out[k].reshape(ref.shape) - ref).max() / max(1.0, np.abs(ref).max())
        assert err <= 2e-6, (k, err)


# ---- training slice: whole block and the BCE losses vs the reference's autograd ---------------------------------
def _block_params(g):
    return [{k: g[f"h{i}_{k}"] for k in ("g_pw", "v_pw", "g_dw", "v_dw", "b_dw")} for i in (1, 2)]


@pytest.mark.parametrize("tag", ["small", "c64", "c96", "c160"])
def test_train_oracle_block_vs_reference_autograd(golden_dir, tag):
    from oracle import wv_oracle_train as OT
    g = np.load(os.path.join(golden_dir, f"grads_block_{tag}.npz"))
    rsp = g["res_scale_param"] if "res_scale_param" in g else None
    r = OT.block_backward(g["x"], _block_params(g), rsp, float(g["pre_scale"]), float(g["res_scale"]), g["dy"])

    def rel(a, b):
        return float(np.abs(a.reshape(b.shape) - b).max() / max(np.abs(b).max(), 1e-30))
    assert rel(r["y"], g["y"]) <= 2e-6 and rel(r["dx"], g["dx"]) <= 2e-6
    for i in (1, 2):
        for k in ("g_pw", "v_pw", "g_dw", "v_dw", "b_dw"):
            assert rel(r["halves"][i - 1]["d" + k], g[f"h{i}_d{k}"]) <= 2e-6, (i, k)
    if rsp is not None:
        assert abs(r["d_res_scale_param"] - float(g["d_res_scale_param"][0])) <= 2e-6 * abs(float(g["d_res_scale_param"][0]))


def test_train_oracle_bce_vs_reference_losses(golden_dir):
    from oracle import wv_oracle_train as OT
    g = np.load(os.path.join(golden_dir, "bce_losses.npz"))
    for i in range(4):
        ld, dz = OT.bce_logits(g[f"c{i}_z"], g[f"c{i}_mask"], g[f"c{i}_msg"])
        ll, dzl = OT.bce_logits(g[f"c{i}_zl"], g[f"c{i}_mask"], None)
        assert abs(ld - float(g[f"c{i}_dec"])) <= 2e-6 * abs(ld) and abs(ll - float(g[f"c{i}_loc"])) <= 2e-6 * abs(ll)
        assert np.abs(dz - g[f"c{i}_dz"]).max() <= 1e-6 / dz.size ** 0.5 and np.abs(dzl - g[f"c{i}_dzl"]).max() <= 1e-6 / dzl.size ** 0.5


@pytest.mark.parametrize("tag", ["r2", "r4", "r5", "r8"])
def test_train_oracle_downsample_unit_vs_reference_autograd(golden_dir, tag):
    from oracle import wv_oracle_train as OT
    g = np.load(os.path.join(golden_dir, f"grads_down_{tag}.npz"))
    r = OT.unit_backward(g["x"], float(g["pre_scale"]), g["g_pw"], g["v_pw"], g["g_dw"], g["v_dw"], g["b_dw"], g["dy"], stride=int(g["ratio"]))
    for k, ref in (("y", "y"), ("dx", "dx"), ("dg_pw", "dg_pw"), ("dv_pw", "dv_pw"), ("dg_dw", "dg_dw"), ("dv_dw", "dv_dw"), ("db_dw", "db_dw")):
        b = g[ref]
        assert float(np.abs(r[k].reshape(b.shape) - b).max() / max(np.abs(b).max(), 1e-30)) <= 2e-6, k
    # stride 1 / k5 is the half
    h = np.load(os.path.join(golden_dir, "grads_half_c64.npz"))
    r = OT.unit_backward(h["x"], float(h["pre_scale"]), h["g_pw"], h["v_pw"], h["g_dw"], h["v_dw"], h["b_dw"], h["dy"])
    assert float(np.abs(r["dx"] - h["dx"]).max() / np.abs(h["dx"]).max()) <= 2e-6


def test_train_oracle_convpre_and_spec_add_vs_reference_autograd(golden_dir):
    from oracle import wv_oracle_train as OT
    g = np.load(os.path.join(golden_dir, "grads_pre_spec.npz"))

    def rel(a, b):
        return float(np.abs(np.asarray(a).reshape(b.shape) - b).max() / max(np.abs(b).max(), 1e-30))
    for i in range(2):
        r = OT.convpre_backward(g[f"pre{i}_x"], float(g[f"pre{i}_in_scale"]), g[f"pre{i}_g"], g[f"pre{i}_v"], g[f"pre{i}_b"], g[f"pre{i}_dy"])
        for k in ("y", "dx", "dg", "dv", "db"):
            assert rel(r[k], g[f"pre{i}_{k}"]) <= 2e-6, (i, k)
    for i in range(3):
        n_fft, hop, rs, mean, std = g[f"spec{i}_meta"]
        sp = g[f"spec{i}_scale_param"] if f"spec{i}_scale_param" in g else None
        # the features the unit consumes are the reference's own (CausalSTFT -> log -> normalise); the oracle's STFT matches them
        mag = O.causal_stft_mag(g[f"spec{i}_wav"], int(n_fft), int(hop))
        P = ((np.log(np.maximum(mag, np.float32(1e-5))) - np.float32(mean)) / np.float32(std)).astype(np.float32)
        assert np.abs(P - g[f"spec{i}_P"]).max() <= 5e-3 and np.abs(P - g[f"spec{i}_P"])[mag > 1e-3].max() <= 2e-5
        r = OT.spec_add_backward(g[f"spec{i}_x"], g[f"spec{i}_P"], g[f"spec{i}_g"], g[f"spec{i}_v"], sp, float(rs), g[f"spec{i}_dy"])
        for k in ("y", "dg", "dv"):
            assert rel(r[k], g[f"spec{i}_{k}"]) <= 2e-6, (i, k)
        assert np.array_equal(g[f"spec{i}_dx"], g[f"spec{i}_dy"])                    # identity path
        if sp is not None:
            assert rel(r["d_scale_param"], g[f"spec{i}_d_scale_param"]) <= 2e-6


def test_train_oracle_convpost_vs_reference_autograd(golden_dir):
    from oracle import wv_oracle_train as OT
    g = np.load(os.path.join(golden_dir, "grads_pre_spec.npz"))
    for i in range(2):
        r = OT.convpost_backward(*(g[f"post{i}_{k}"] for k in ("x", "g_dw", "v_dw", "g_pw", "v_pw", "b", "dy")))
        for k, ref in (("y", "y"), ("dx", "dx"), ("dg_dw", "dg_dw"), ("dv_dw", "dv_dw"), ("dg_pw", "dg_pw"), ("dv_pw", "dv_pw"), ("db", "db")):
            b = g[f"post{i}_{ref}"]
            assert float(np.abs(r[k].reshape(b.shape) - b).max() / max(np.abs(b).max(), 1e-30)) <= 2e-6, (i, k)


@pytest.mark.parametrize("name", ["locator", "detector"])
def test_training_gradient_oracle_vs_whole_reference_nets(golden_dir, name):
    """oracle/wv_oracle_train_torch.py (differentiable restatement, live weight norm) against the reference's own autograd
    through whole Locator / Detector modules and its loss classes: loss, logits, every parameter gradient, dL/dx."""
    import ast
    from oracle import wv_oracle_train_torch as OTT
    from waveverify_amd.config import default_config
    from waveverify_amd.init import random_state_dict
    g = np.load(os.path.join(golden_dir, f"netgrads_{name}.npz"))
    c = ast.literal_eval(str(g["cfg"][0]))
    seed, kind = c.pop("seed"), c.pop("kind")
    cfg = default_config(kind, **c)
    sd = random_state_dict(cfg, seed, parametrized=True)
    loss, logits, grads, dx = OTT.loss_and_grads(cfg, sd, g["x"], g["mask"], g["msg"] if "msg" in g else None, need_dx=True)
    assert abs(loss - float(g["loss"])) <= 1e-8 * abs(loss)
    assert np.abs(logits - g["logits"]).max() <= 2e-6 * max(1.0, np.abs(g["logits"]).max())
    assert np.abs(dx - g["dx"]).max() <= 1e-4 * np.abs(g["dx"]).max()    # through log|STFT| near its clamps: amplified rounding of the float32 fixture
    ref_keys = [k[2:] for k in g.files if k.startswith("g:")]
    assert sorted(ref_keys) == sorted(grads)                       # the same parameters receive gradients (msg MLP / FiLM do not)
    for k in ref_keys:
        r = g["g:" + k]
        assert np.abs(grads[k] - r).max() <= 5e-6 * max(np.abs(r).max(), 1e-12) + 1e-14, k


def test_train_oracle_upsample_unit_vs_reference_autograd(golden_dir):
    from oracle import wv_oracle_train as OT
    g = np.load(os.path.join(golden_dir, "grads_pre_spec.npz"))
    for i in range(4):
        r = OT.up_backward(g[f"up{i}_x"], float(g[f"up{i}_meta"][1]), *(g[f"up{i}_{k}"] for k in ("g_ct", "v_ct", "g_pw", "v_pw", "b", "dy")))
        for k in ("y", "dx", "dg_ct", "dv_ct", "dg_pw", "dv_pw", "db"):
            b = g[f"up{i}_{k}"]
            assert float(np.abs(r[k].reshape(b.shape) - b).max() / max(np.abs(b).max(), 1e-30)) <= 2e-6, (i, k)


def test_training_gradient_oracle_vs_whole_reference_generator(golden_dir):
    """The generator's gradient oracle against the reference Generator's own autograd (float32: its encoder casts the message to
    float32, so the module cannot run in float64) -- all 190-odd tensors incl. message MLP, FiLM heads, decoder."""
    import ast
    from oracle import wv_oracle_train_torch as OTT
    from waveverify_amd.config import default_config
    from waveverify_amd.init import random_state_dict
    g = np.load(os.path.join(golden_dir, "netgrads_generator.npz"))
    c = ast.literal_eval(str(g["cfg"][0]))
    seed, kind = c.pop("seed"), c.pop("kind")
    cfg = default_config(kind, **c)
    sd = random_state_dict(cfg, seed, parametrized=True)
    loss, wm, grads, dx = OTT.generator_loss_and_grads(cfg, sd, g["x"], g["msg"], g["target"], need_dx=True)
    assert abs(loss - float(g["loss"])) <= 1e-4 * abs(loss)
    assert np.abs(wm - g["wm"]).max() <= 2e-6
    ref_keys = [k[2:] for k in g.files if k.startswith("g:")]
    assert sorted(ref_keys) == sorted(grads)
    for k in ref_keys:
        r = g["g:" + k]
        assert np.abs(grads[k] - r).max() <= 2e-3 * max(np.abs(r).max(), 1e-12) + 1e-9, (k, np.abs(grads[k] - r).max(), np.abs(r).max())
