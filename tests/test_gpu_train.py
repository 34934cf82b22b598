"""First training-step slice: forward + backward of a ResnetBlock half with live weight
norm on the GPU, against (a) the REFERENCE modules' own autograd (tests/golden/grads_half_*.npz) and (b) the numpy
oracle on larger seeded shapes.  Bar: 1e-4 relative to the tensor's largest magnitude (f32 MFMA chains, two-stage
deterministic reductions)."""
import os

import numpy as np
import pytest
import torch

from oracle import wv_oracle_train as OT

pytestmark = pytest.mark.gpu
KEYS = ("dx", "dg_pw", "dv_pw", "dg_dw", "dv_dw", "db_dw")


def rel(got, ref):
    got = got.detach().cpu().numpy().reshape(ref.shape)
    assert np.isfinite(got).all()
    return float(np.abs(got - ref).max() / max(1.0, np.abs(ref).max()))


def run(x, s, p, dy):
    from waveverify_amd.train import TrainHalf
    half = TrainHalf(x.shape[1])
    pt = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in p.items()}
    xt, dyt = torch.from_numpy(x).cuda(), torch.from_numpy(dy).cuda()
    y = half.forward(xt, pt, s)
    g = half.backward(xt, pt, s, dyt)
    g2 = half.backward(xt, pt, s, dyt)
    for k in KEYS:
        assert torch.equal(g[k], g2[k]), f"{k}: the backward pass must be deterministic"
    return y, g


@pytest.mark.parametrize("tag", ["c64", "c96", "c160"])
def test_half_block_gradients_vs_reference_autograd(golden_dir, tag):
    f = np.load(os.path.join(golden_dir, f"grads_half_{tag}.npz"))
    p = {k: f[k] for k in ("g_pw", "v_pw", "g_dw", "v_dw", "b_dw")}
    y, g = run(f["x"], float(f["pre_scale"]), p, f["dy"])
    assert rel(y, f["y"]) <= 2e-5
    for k in KEYS:
        assert rel(g[k], f[k]) <= 1e-4, (k, rel(g[k], f[k]))


@pytest.mark.parametrize("B,C,T", [(4, 128, 1000), (3, 256, 400), (2, 64, 16000), (5, 96, 36)])
def test_half_block_gradients_vs_oracle(B, C, T):
    rng = np.random.default_rng(B * 1000 + C + T)
    x = rng.standard_normal((B, C, T)).astype(np.float32)
    dy = rng.standard_normal((B, C, T)).astype(np.float32)
    p = dict(g_pw=(0.5 + np.abs(rng.standard_normal((C, 1, 1)))).astype(np.float32),
             v_pw=(rng.standard_normal((C, C, 1)) * C ** -0.5).astype(np.float32),
             g_dw=(0.5 + np.abs(rng.standard_normal((C, 1, 1)))).astype(np.float32),
             v_dw=(rng.standard_normal((C, 1, 5)) * 0.45).astype(np.float32),
             b_dw=(rng.standard_normal(C) * 0.1).astype(np.float32))
    s = 0.7071068
    ref = OT.half_backward(x, s, p["g_pw"], p["v_pw"], p["g_dw"], p["v_dw"], p["b_dw"], dy)
    y, g = run(x, s, p, dy)
    assert rel(y, ref["y"]) <= 2e-5
    for k in KEYS:
        assert rel(g[k], ref[k]) <= 1e-4, (k, rel(g[k], ref[k]))


def test_training_slice_errors_are_loud():
    """Ragged lengths are served by the round-1 core (covered by test_unit_gradients_vs_oracle); what is rejected:
    CPU tensors, bad geometry, and whole blocks at T % 4 != 0 (their glue kernels are 16-byte vectorised)."""
    from waveverify_amd.train import TrainBlock, TrainHalf, TrainUnit
    half = TrainHalf(64)
    p = dict(g_pw=torch.ones(64, device="cuda"), v_pw=torch.randn(64, 64, device="cuda"), g_dw=torch.ones(64, device="cuda"),
             v_dw=torch.randn(64, 5, device="cuda"), b_dw=torch.zeros(64, device="cuda"))
    assert half.forward(torch.randn(1, 64, 10, device="cuda"), p, 1.0).shape == (1, 64, 10)
    with pytest.raises(RuntimeError, match="GPU"):
        half.forward(torch.randn(1, 64, 16), p, 1.0)
    with pytest.raises(RuntimeError, match="kernel size"):
        TrainUnit(64, 64, 17, 1)
    with pytest.raises(RuntimeError, match="kernel size"):
        TrainUnit(64, 64, 2, 4)
    with pytest.raises(RuntimeError, match="T % 4"):
        TrainBlock(64).forward(torch.randn(1, 64, 10, device="cuda"), [p, p], None, 1.0, 0.5)


# ---- whole SEANetResnetBlock ----------------------------------------------------------------------------------------
def _cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def run_block(x, ps, rsp, pre, rs, dy):
    from waveverify_amd.train import TrainBlock
    blk = TrainBlock(x.shape[1])
    pt = [{k: _cu(v) for k, v in p.items()} for p in ps]
    rt = None if rsp is None else _cu(rsp)
    y, saved = blk.forward(_cu(x), pt, rt, pre, rs)
    g = blk.backward(_cu(x), pt, rt, pre, rs, _cu(dy), saved)
    g2 = blk.backward(_cu(x), pt, rt, pre, rs, _cu(dy), saved)
    assert torch.equal(g["dx"], g2["dx"]) and all(torch.equal(g["halves"][i][k], g2["halves"][i][k]) for i in (0, 1) for k in KEYS[1:])
    if rsp is not None:
        assert torch.equal(g["d_res_scale_param"], g2["d_res_scale_param"])
    return y, g


@pytest.mark.parametrize("tag", ["small", "c64", "c96", "c160"])
def test_block_gradients_vs_reference_autograd(golden_dir, tag):
    f = np.load(os.path.join(golden_dir, f"grads_block_{tag}.npz"))
    ps = [{k: f[f"h{i}_{k}"] for k in ("g_pw", "v_pw", "g_dw", "v_dw", "b_dw")} for i in (1, 2)]
    rsp = f["res_scale_param"] if "res_scale_param" in f else None
    y, g = run_block(f["x"], ps, rsp, float(f["pre_scale"]), float(f["res_scale"]), f["dy"])
    assert rel(y, f["y"]) <= 2e-5 and rel(g["dx"], f["dx"]) <= 1e-4
    for i in (1, 2):
        for k in KEYS[1:]:
            assert rel(g["halves"][i - 1][k], f[f"h{i}_{k}"]) <= 1e-4, (i, k)
    if rsp is not None:
        ref = float(f["d_res_scale_param"][0])
        assert abs(float(g["d_res_scale_param"].item()) - ref) <= 1e-4 * max(1.0, abs(ref))


@pytest.mark.parametrize("B,C,T,with_param", [(4, 128, 1000, True), (2, 64, 16000, False), (3, 192, 400, True),
                                               # the fused epilogues (saved 1x1 output, residual sum, ELU' + shortcut in the dx GEMM) on every tiling the
                                               # launcher picks: one clip (per-clip tiles), K >= 256 (three DMA stages), 96-row tiles, a padded
                                               # 160-channel layer, flat tiles at T = 400 / 2000, a narrow ragged layer on the round-1 core
                                               (1, 128, 2000, True), (2, 256, 400, True), (5, 96, 2000, False), (2, 160, 1000, True),
                                               (3, 512, 52, True), (2, 40, 36, False)])
def test_block_gradients_vs_oracle(B, C, T, with_param):
    rng = np.random.default_rng(B * 77 + C + T)
    x = rng.standard_normal((B, C, T)).astype(np.float32)
    dy = rng.standard_normal((B, C, T)).astype(np.float32)
    ps = [dict(g_pw=(0.5 + np.abs(rng.standard_normal((C, 1, 1)))).astype(np.float32),
               v_pw=(rng.standard_normal((C, C, 1)) * C ** -0.5).astype(np.float32),
               g_dw=(0.5 + np.abs(rng.standard_normal((C, 1, 1)))).astype(np.float32),
               v_dw=(rng.standard_normal((C, 1, 5)) * 0.45).astype(np.float32),
               b_dw=(rng.standard_normal(C) * 0.1).astype(np.float32)) for _ in range(2)]
    rsp = np.array([0.8], np.float32) if with_param else None
    pre, rs = 0.8164966, 0.5773503
    ref = OT.block_backward(x, ps, rsp, pre, rs, dy)
    y, g = run_block(x, ps, rsp, pre, rs, dy)
    assert rel(y, ref["y"]) <= 2e-5 and rel(g["dx"], ref["dx"]) <= 1e-4
    for i in (0, 1):
        for k in KEYS[1:]:
            assert rel(g["halves"][i][k], ref["halves"][i][k]) <= 1e-4, (i, k)
    if with_param:
        assert abs(float(g["d_res_scale_param"].item()) - ref["d_res_scale_param"]) <= 1e-4 * max(1.0, abs(ref["d_res_scale_param"]))


@pytest.mark.parametrize("B,C,T", [(3, 128, 2000), (2, 40, 52)])
def test_block_backward_paths_agree(B, C, T):
    """The block keeps its 1x1 outputs from the forward kernel, takes ELU' and the identity shortcut in the dx GEMM's epilogue and writes
    parameter gradients into caller-provided arena views; the standalone half recomputes the 1x1 output and runs those steps as
    separate kernels.  Both routes must give the same numbers (wide layer: the LDS-DMA core; narrow ragged one: the round-1 core,
    where the block falls back to the unfused steps)."""
    from waveverify_amd.train import TrainBlock, TrainHalf
    rng = np.random.default_rng(C + T)
    x, dy = (_cu(rng.standard_normal((B, C, T)).astype(np.float32)) for _ in range(2))
    ps = [dict(g_pw=_cu((0.5 + np.abs(rng.standard_normal((C, 1, 1)))).astype(np.float32)),
               v_pw=_cu((rng.standard_normal((C, C, 1)) * C ** -0.5).astype(np.float32)),
               g_dw=_cu((0.5 + np.abs(rng.standard_normal((C, 1, 1)))).astype(np.float32)),
               v_dw=_cu((rng.standard_normal((C, 1, 5)) * 0.45).astype(np.float32)),
               b_dw=_cu((rng.standard_normal(C) * 0.1).astype(np.float32))) for _ in range(2)]
    pre, rs = 0.8164966, 0.5773503
    blk = TrainBlock(C)
    y, saved = blk.forward(x, ps, None, pre, rs)
    g = blk.backward(x, ps, None, pre, rs, dy, saved)
    # the same block with destinations inside one flat arena, at offsets that are not multiples of 4 floats
    sizes = dict(dg_pw=C, dv_pw=C * C, dg_dw=C, dv_dw=C * 5, db_dw=C)
    arena = torch.full((2 * sum(sizes.values()) + 16,), float("nan"), device="cuda")
    off, into = 3, dict(halves=[{}, {}])
    for i in (0, 1):
        for k, n in sizes.items():
            into["halves"][i][k] = arena[off:off + n]
            off += n + 1
    g_in = blk.backward(x, ps, None, pre, rs, dy, saved, into)
    assert torch.equal(g_in["dx"], g["dx"])
    for i in (0, 1):
        for k in sizes:
            assert g_in["halves"][i][k].data_ptr() == into["halves"][i][k].data_ptr()
            assert torch.equal(into["halves"][i][k].reshape(-1), g["halves"][i][k].reshape(-1)), (i, k)
    # the two halves on their own (recompute route) + the residual arithmetic in torch
    h1, h2 = TrainHalf(C), TrainHalf(C)
    u = h1.forward(x, ps[0], pre)
    v = h2.forward(u, ps[1], 1.0)
    assert torch.equal(y, x + rs * v) or float((y - (x + rs * v)).abs().max()) <= 1e-6 * float(y.abs().max())
    g2 = h2.backward(u, ps[1], 1.0, rs * dy)
    g1 = h1.backward(x, ps[0], pre, g2["dx"])
    scale = float(g["dx"].abs().max())
    assert float((g["dx"] - (g1["dx"] + dy)).abs().max()) <= 2e-6 * scale
    for i, gh in enumerate((g1, g2)):
        for k in sizes:
            ref = gh[k].reshape(-1)
            assert float((g["halves"][i][k].reshape(-1) - ref).abs().max()) <= 2e-6 * max(1.0, float(ref.abs().max())), (i, k)


# ---- BCE losses ---------------------------------------------------------------------------------------------------------
def test_bce_losses_vs_reference_classes(golden_dir):
    from waveverify_amd.train import bce_logits
    f = np.load(os.path.join(golden_dir, "bce_losses.npz"))
    for i in range(4):
        z, zl, mask, msg = (_cu(f[f"c{i}_{k}"]) for k in ("z", "zl", "mask", "msg"))
        ld, dz = bce_logits(z, mask, msg)
        ll, dzl = bce_logits(zl, mask, None)
        assert abs(float(ld.item()) - float(f[f"c{i}_dec"])) <= 2e-6 * abs(float(f[f"c{i}_dec"]))
        assert abs(float(ll.item()) - float(f[f"c{i}_loc"])) <= 2e-6 * abs(float(f[f"c{i}_loc"]))
        n, nl = z.numel(), zl.numel()
        assert float((dz.cpu() - torch.from_numpy(f[f"c{i}_dz"])).abs().max()) <= 2e-7 / n * max(1.0, n ** 0.5) + 1e-7 / n
        assert float((dzl.cpu() - torch.from_numpy(f[f"c{i}_dzl"])).abs().max()) <= 2e-7 / nl * max(1.0, nl ** 0.5) + 1e-7 / nl
        l2, dz2 = bce_logits(z, mask, msg)
        assert torch.equal(ld, l2) and torch.equal(dz, dz2)                    # fixed-order reduction


def test_bce_at_training_size_vs_oracle_and_errors():
    """64 clips x 16 bits x 16000 samples (BASELINE configs[2] per-GPU batch): loss and gradient against the float64
    oracle; grad_scale scales the gradient only; shape errors are ValueError as in the reference."""
    from waveverify_amd.train import bce_logits
    rng = np.random.default_rng(5)
    B, nb, T = 64, 16, 16000
    z = (2.0 * rng.standard_normal((B, nb, T))).astype(np.float32)
    mask = (rng.random((B, 1, T)) < 0.8).astype(np.float32)
    msg = rng.integers(0, 2, (B, nb)).astype(np.float32)
    ref_l, ref_g = OT.bce_logits(z, mask, msg)
    l, dz = bce_logits(_cu(z), _cu(mask), _cu(msg), grad_scale=3.0)
    assert abs(float(l.item()) - ref_l) <= 2e-6 * ref_l
    assert float(np.abs(dz.cpu().numpy() / 3.0 - ref_g).max()) <= 2e-7 / z.size
    l0, none = bce_logits(_cu(z), None, _cu(msg), want_grad=False)
    assert none is None and abs(float(l0.item()) - OT.bce_logits(z, None, msg)[0]) <= 2e-6 * ref_l
    with pytest.raises(ValueError, match="3D"):
        bce_logits(_cu(z[0]), None, None)
    with pytest.raises(ValueError, match="ground_truth_message"):
        bce_logits(_cu(z), _cu(mask), _cu(msg[:, :8]))
    with pytest.raises(ValueError, match="ground_truth_presence"):
        bce_logits(_cu(z), _cu(mask[:, :, :100]), _cu(msg))
    with pytest.raises(ValueError, match="same shape"):
        bce_logits(_cu(z), _cu(mask), None)


# ---- the strided unit (encoder Downsample) ---------------------------------------------------------------------------
def run_unit(x, s, p, dy, ks, stride, elu=True, need_dx=True):
    from waveverify_amd.train import TrainUnit
    u = TrainUnit(x.shape[1], p["v_pw"].shape[0], ks, stride)
    pt = {k: _cu(v) for k, v in p.items()}
    y = u.forward(_cu(x), pt, s, elu)
    g = u.backward(_cu(x), pt, s, _cu(dy), elu, need_dx)
    g2 = u.backward(_cu(x), pt, s, _cu(dy), elu, need_dx)
    for k in KEYS:
        assert (g[k] is None and g2[k] is None) or torch.equal(g[k], g2[k]), k
    return y, g


@pytest.mark.parametrize("tag", ["r2", "r4", "r5", "r8"])
def test_downsample_unit_gradients_vs_reference_autograd(golden_dir, tag):
    f = np.load(os.path.join(golden_dir, f"grads_down_{tag}.npz"))
    p = {k: f[k] for k in ("g_pw", "v_pw", "g_dw", "v_dw", "b_dw")}
    r = int(f["ratio"])
    y, g = run_unit(f["x"], float(f["pre_scale"]), p, f["dy"], 2 * r, r)
    assert rel(y, f["y"]) <= 2e-5
    for k in KEYS:
        assert rel(g[k], f[k]) <= 1e-4, (k, rel(g[k], f[k]))


@pytest.mark.parametrize("B,K,M,T,ks,stride,elu", [(4, 64, 128, 16000, 4, 2, True), (3, 128, 256, 8000, 8, 4, True), (2, 256, 512, 2000, 10, 5, True),
                                                   (2, 512, 1024, 400, 16, 8, True), (3, 96, 64, 404, 5, 1, False), (2, 64, 128, 36, 7, 3, True),
                                                   # ragged lengths and narrow layers leave the LDS-DMA core: the locator's first stage (C = 32), T = 50
                                                   (3, 32, 32, 50, 5, 1, True), (2, 32, 64, 1001, 16, 8, True), (2, 512, 1024, 50, 4, 2, True), (2, 24, 40, 7, 5, 1, False)])
def test_unit_gradients_vs_oracle(B, K, M, T, ks, stride, elu):
    rng = np.random.default_rng(K + M + T)
    x = rng.standard_normal((B, K, T)).astype(np.float32)
    p = dict(g_pw=(0.5 + np.abs(rng.standard_normal((M, 1, 1)))).astype(np.float32),
             v_pw=(rng.standard_normal((M, K, 1)) * K ** -0.5).astype(np.float32),
             g_dw=(0.5 + np.abs(rng.standard_normal((M, 1, 1)))).astype(np.float32),
             v_dw=(rng.standard_normal((M, 1, ks)) * ks ** -0.5).astype(np.float32),
             b_dw=(rng.standard_normal(M) * 0.1).astype(np.float32))
    dy = rng.standard_normal((B, M, -(-T // stride))).astype(np.float32)
    s = 0.7071068
    ref = OT.unit_backward(x, s, p["g_pw"], p["v_pw"], p["g_dw"], p["v_dw"], p["b_dw"], dy, stride=stride, elu=elu)
    y, g = run_unit(x, s, p, dy, ks, stride, elu)
    assert rel(y, ref["y"]) <= 2e-5
    for k in KEYS:
        assert rel(g[k], ref[k]) <= 1e-4, (k, rel(g[k], ref[k]))
    _, g0 = run_unit(x, s, p, dy, ks, stride, elu, need_dx=False)                  # first layer: no input gradient
    assert g0["dx"] is None and torch.equal(g0["dv_pw"], g["dv_pw"])


# ---- optimizer ---------------------------------------------------------------------------------------------------------
def test_flat_adamw_with_clipping_vs_torch():
    """clip_grad_norm_ -> AdamW -> ExponentialLR exactly as scripts/train.py:1346-1358 chains them, against torch's own
    CPU implementations, over several steps (one of them clipped, one not)."""
    from waveverify_amd.train import FlatAdamW
    rng = np.random.default_rng(3)
    n = 100_003
    p0 = rng.standard_normal(n).astype(np.float32)
    ref_p = torch.nn.Parameter(torch.from_numpy(p0.copy()))
    opt = torch.optim.AdamW([ref_p], lr=1e-2, betas=(0.8, 0.99))
    sched = torch.optim.lr_scheduler.ExponentialLR(opt, 0.99)
    mine = FlatAdamW(n, lr=1e-2, betas=(0.8, 0.99), gamma=0.99)
    p = torch.from_numpy(p0.copy()).cuda()
    for it, gs in enumerate((1.0, 1e-3, 0.05, 2.0)):
        g = (gs * rng.standard_normal(n)).astype(np.float32)
        ref_p.grad = torch.from_numpy(g.copy())
        ref_norm = torch.nn.utils.clip_grad_norm_([ref_p], 10.0)
        opt.step(); sched.step()
        norm = mine.step(p, torch.from_numpy(g).cuda(), max_norm=10.0)
        assert abs(float(norm.item()) - float(ref_norm)) <= 2e-6 * float(ref_norm)
        err = float((p.cpu() - ref_p.detach()).abs().max())
        assert err <= 2e-6, (it, err)
    # without clipping
    g = rng.standard_normal(n).astype(np.float32)
    ref_p.grad = torch.from_numpy(g.copy()); opt.step(); sched.step()
    assert mine.step(p, torch.from_numpy(g).cuda()) is None
    assert float((p.cpu() - ref_p.detach()).abs().max()) <= 2e-6
    with pytest.raises(ValueError):
        mine.step(p[:10].contiguous(), p[:10].contiguous())


def test_block_trainer_closed_loop():
    """fold -> forward -> DecodingLoss -> backward -> (all-reduce) -> clip + AdamW: the loss of a single block fitted to a
    fixed message falls monotonically-ish, parameters stay finite, and two trainers with the same seed agree bitwise."""
    from waveverify_amd.train import BlockTrainer
    rng = np.random.default_rng(0)
    B, C, T = 8, 64, 2000
    x = _cu(rng.standard_normal((B, C, T)).astype(np.float32))
    mask = _cu((rng.random((B, 1, T)) < 0.8).astype(np.float32))
    msg = _cu(rng.integers(0, 2, (B, C)).astype(np.float32))
    a, b = BlockTrainer(C, pre_scale=0.8, seed=1, lr=5e-3), BlockTrainer(C, pre_scale=0.8, seed=1, lr=5e-3)
    la = [float(a.step(x, mask, msg)[0].item()) for _ in range(25)]
    lb = [float(b.step(x, mask, msg)[0].item()) for _ in range(25)]
    assert la == lb and torch.equal(a.arena, b.arena)
    assert la[-1] < 0.8 * la[0] and all(np.isfinite(la)) and bool(torch.isfinite(a.arena).all())
    assert a.opt.t == 25 and abs(a.opt.lr - 5e-3 * 0.999996 ** 25) < 1e-12


# ---- conv_pre and the SpecBlock add -----------------------------------------------------------------------------------
def test_convpre_gradients_vs_reference_autograd_and_oracle(golden_dir):
    from waveverify_amd.train import TrainConvPre
    f = np.load(os.path.join(golden_dir, "grads_pre_spec.npz"))
    for i in range(2):
        x, dy, s = f[f"pre{i}_x"], f[f"pre{i}_dy"], float(f[f"pre{i}_in_scale"])
        p = {k: _cu(f[f"pre{i}_{k}"]) for k in ("g", "v", "b")}
        u = TrainConvPre(dy.shape[1], f[f"pre{i}_v"].shape[-1])
        assert rel(u.forward(_cu(x), p, s), f[f"pre{i}_y"]) <= 2e-5
        g = u.backward(_cu(x), p, s, _cu(dy), need_dx=True)
        for k in ("dx", "dg", "dv", "db"):
            assert rel(g[k], f[f"pre{i}_{k}"]) <= 1e-4, (i, k)
        assert u.backward(_cu(x), p, s, _cu(dy))["dx"] is None
    # the detector's first layer at the training batch: 64 clips x 1 s, C = 64
    rng = np.random.default_rng(1)
    B, C, T, ks = 64, 64, 16000, 5
    x = (0.1 * rng.standard_normal((B, 1, T))).astype(np.float32)
    dy = rng.standard_normal((B, C, T)).astype(np.float32)
    p = dict(g=(0.5 + np.abs(rng.standard_normal((C, 1, 1)))).astype(np.float32), v=(0.45 * rng.standard_normal((C, 1, ks))).astype(np.float32),
             b=(0.1 * rng.standard_normal(C)).astype(np.float32))
    ref = OT.convpre_backward(x, 8.912, p["g"], p["v"], p["b"], dy)
    u = TrainConvPre(C, ks)
    pt = {k: _cu(v) for k, v in p.items()}
    assert rel(u.forward(_cu(x), pt, 8.912), ref["y"]) <= 2e-5
    g = u.backward(_cu(x), pt, 8.912, _cu(dy), need_dx=True)
    g2 = u.backward(_cu(x), pt, 8.912, _cu(dy), need_dx=True)
    for k in ("dx", "dg", "dv", "db"):
        assert rel(g[k], ref[k]) <= 1e-4 and torch.equal(g[k], g2[k]), k


def test_spec_add_gradients_vs_reference_autograd_and_oracle(golden_dir):
    from waveverify_amd import ops
    from waveverify_amd.train import TrainSpecAdd
    f = np.load(os.path.join(golden_dir, "grads_pre_spec.npz"))
    for i in range(3):
        n_fft, hop, rs, mean, std = f[f"spec{i}_meta"]
        sp = _cu(f[f"spec{i}_scale_param"]) if f"spec{i}_scale_param" in f else None
        # features from OUR STFT kernel on the fixture's waveform, as the training step would produce them
        P = ops.stft_logmag(_cu(f[f"spec{i}_wav"]), int(n_fft), int(hop), mean=float(mean), std=float(std))
        p = {k: _cu(f[f"spec{i}_{k}"]) for k in ("g", "v")}
        u = TrainSpecAdd(f[f"spec{i}_x"].shape[1], int(n_fft) // 2 + 1)
        y = u.forward(_cu(f[f"spec{i}_x"]), P, p, sp, float(rs))
        assert rel(y, f[f"spec{i}_y"]) <= 5e-5
        g = u.backward(P, p, sp, float(rs), _cu(f[f"spec{i}_dy"]))
        for k in ("dg", "dv"):
            assert rel(g[k], f[f"spec{i}_{k}"]) <= 2e-4, (i, k)
        if sp is not None:
            ref = float(f[f"spec{i}_d_scale_param"][0])
            assert abs(float(g["d_scale_param"].item()) - ref) <= 2e-4 * max(1.0, abs(ref))
    # a detector scale at the training batch against the oracle (C = 256, F = 129, T = 2000) and spec_post's shape (T = 50)
    for B, C, F, T in ((16, 256, 129, 2000), (4, 1024, 513, 50)):
        _spec_vs_oracle(B, C, F, T)


def _spec_vs_oracle(B, C, F, T):
    from waveverify_amd.train import TrainSpecAdd
    rng = np.random.default_rng(2)
    x, P, dy = (rng.standard_normal(s).astype(np.float32) for s in ((B, C, T), (B, F, T), (B, C, T)))
    p = dict(g=(0.5 + np.abs(rng.standard_normal((C, 1, 1)))).astype(np.float32), v=(rng.standard_normal((C, F, 1)) * F ** -0.5).astype(np.float32))
    spn = np.array([0.7], np.float32)
    ref = OT.spec_add_backward(x, P, p["g"], p["v"], spn, 0.5773503, dy)
    u = TrainSpecAdd(C, F)
    pt = {k: _cu(v) for k, v in p.items()}
    assert rel(u.forward(_cu(x), _cu(P), pt, _cu(spn), 0.5773503), ref["y"]) <= 2e-5
    g = u.backward(_cu(P), pt, _cu(spn), 0.5773503, _cu(dy))
    assert rel(g["dg"], ref["dg"]) <= 1e-4 and rel(g["dv"], ref["dv"]) <= 1e-4
    assert abs(float(g["d_scale_param"].item()) - ref["d_scale_param"]) <= 1e-4 * abs(ref["d_scale_param"])


def test_convpost_gradients_vs_reference_autograd_and_oracle(golden_dir):
    from waveverify_amd.train import TrainConvPost
    f = np.load(os.path.join(golden_dir, "grads_pre_spec.npz"))
    names = ("g_dw", "v_dw", "g_pw", "v_pw", "b")
    for i in range(2):
        x, dy = f[f"post{i}_x"], f[f"post{i}_dy"]
        u = TrainConvPost(x.shape[1], dy.shape[1], f[f"post{i}_v_dw"].shape[-1])
        p = {k: _cu(f[f"post{i}_{k}"]) for k in names}
        assert rel(u.forward(_cu(x), p), f[f"post{i}_y"]) <= 2e-5
        g = u.backward(_cu(x), p, _cu(dy))
        for k, ref in (("dx", "dx"), ("dg_dw", "dg_dw"), ("dv_dw", "dv_dw"), ("dg_pw", "dg_pw"), ("dv_pw", "dv_pw"), ("db", "db")):
            assert rel(g[k], f[f"post{i}_{ref}"]) <= 1e-4, (i, k, rel(g[k], f[f"post{i}_{ref}"]))
    # the detector's shape at the training batch (C = 1024 -> D = 128, 50 frames), with and without the norm; a silent column
    rng = np.random.default_rng(4)
    B, C, D, T, ks = 64, 1024, 128, 50, 5
    x = rng.standard_normal((B, C, T)).astype(np.float32)
    dy = rng.standard_normal((B, D, T)).astype(np.float32)
    p = dict(g_dw=(0.5 + np.abs(rng.standard_normal((C, 1, 1)))).astype(np.float32), v_dw=(0.45 * rng.standard_normal((C, 1, ks))).astype(np.float32),
             g_pw=(0.5 + np.abs(rng.standard_normal((D, 1, 1)))).astype(np.float32), v_pw=(rng.standard_normal((D, C, 1)) * C ** -0.5).astype(np.float32),
             b=rng.standard_normal(D).astype(np.float32))
    pt = {k: _cu(v) for k, v in p.items()}
    for l2 in (True, False):
        ref = OT.convpost_backward(x, p["g_dw"], p["v_dw"], p["g_pw"], p["v_pw"], p["b"], dy, l2norm=l2)
        u = TrainConvPost(C, D, ks, l2norm=l2)
        assert rel(u.forward(_cu(x), pt), ref["y"]) <= 2e-5
        g, g2 = u.backward(_cu(x), pt, _cu(dy)), u.backward(_cu(x), pt, _cu(dy))
        for k in ("dx", "dg_dw", "dv_dw", "dg_pw", "dv_pw", "db"):
            assert rel(g[k], ref[k]) <= 1e-4 and torch.equal(g[k], g2[k]), (l2, k, rel(g[k], ref[k]))


@pytest.mark.parametrize("B,D,O,nb,hop,N,T", [(4, 128, 32, 16, 320, 50, 16000), (3, 64, 32, 1, 32, 500, 16000), (2, 128, 32, 16, 320, 51, 16001),
                                              (2, 16, 8, 3, 5, 7, 33)])
def test_head_forward_backward_vs_torch_modules(B, D, O, nb, hop, N, T):
    """The reference's head IS torch.nn.ConvTranspose1d(D, O, hop, hop) -> [:, :, :T] -> nn.Conv1d(O, nb, 1) (detector.py:209-218,
    304-310): its CPU autograd in float64 is the oracle.  Detector, locator, a ragged length and a tiny case."""
    from waveverify_amd.train import TrainHead
    torch.manual_seed(D + hop)
    rev = torch.nn.ConvTranspose1d(D, O, hop, hop).double()
    last = torch.nn.Conv1d(O, nb, 1).double()
    z = torch.randn(B, D, N, dtype=torch.float64, requires_grad=True)
    logits = last(rev(z)[:, :, :T])
    dl = torch.randn_like(logits)
    logits.backward(dl)
    p = dict(w_rev=rev.weight.detach().float().cuda(), b_rev=rev.bias.detach().float().cuda(), w_last=last.weight.detach().float().cuda(),
             b_last=last.bias.detach().float().cuda())
    h = TrainHead(D, O, nb, hop)
    got = h.forward(z.detach().float().cuda(), p, T)
    assert rel(got, logits.detach().numpy()) <= 2e-5
    g = h.backward(z.detach().float().cuda(), p, dl.float().cuda())
    g2 = h.backward(z.detach().float().cuda(), p, dl.float().cuda())
    for k, ref in (("dz", z.grad), ("dw_rev", rev.weight.grad), ("db_rev", rev.bias.grad), ("dw_last", last.weight.grad[:, :, 0]), ("db_last", last.bias.grad)):
        assert rel(g[k], ref.numpy()) <= 1e-4, (k, rel(g[k], ref.numpy()))
        assert torch.equal(g[k], g2[k]), k


def test_upsample_unit_gradients_vs_reference_autograd_and_oracle(golden_dir):
    from waveverify_amd.train import TrainUp
    f = np.load(os.path.join(golden_dir, "grads_pre_spec.npz"))
    names = ("g_ct", "v_ct", "g_pw", "v_pw", "b")
    keys = ("dx", "dg_ct", "dv_ct", "dg_pw", "dv_pw", "db")
    for i in range(4):
        r, sc = int(f[f"up{i}_meta"][0]), float(f[f"up{i}_meta"][1])
        x, dy = f[f"up{i}_x"], f[f"up{i}_dy"]
        u = TrainUp(x.shape[1], dy.shape[1], r)
        p = {k: _cu(f[f"up{i}_{k}"]) for k in names}
        assert rel(u.forward(_cu(x), p, sc), f[f"up{i}_y"]) <= 2e-5
        g = u.backward(_cu(x), p, sc, _cu(dy))
        for k in keys:
            assert rel(g[k], f[f"up{i}_{k}"]) <= 1e-4, (i, k, rel(g[k], f[f"up{i}_{k}"]))
    # the generator's decoder shapes at a training batch: 768 -> 384 (r = 8, 50 frames) and 192 -> 96 (r = 2, 8000 -> 16000)
    # ... and the other two ratios of the per-frame ConvTranspose kernels (r = 5: scalar loads, r = 4: one 16-byte vector per frame)
    for B, K, M, T, r in ((16, 768, 384, 50, 8), (4, 192, 96, 8000, 2), (3, 384, 192, 400, 5), (2, 256, 128, 2000, 4)):
        rng = np.random.default_rng(K + r)
        x = rng.standard_normal((B, K, T)).astype(np.float32)
        dy = rng.standard_normal((B, M, T * r)).astype(np.float32)
        p = dict(g_ct=(0.5 + np.abs(rng.standard_normal((K, 1, 1)))).astype(np.float32), v_ct=(rng.standard_normal((K, 1, 2 * r)) * (2 * r) ** -0.5).astype(np.float32),
                 g_pw=(0.5 + np.abs(rng.standard_normal((M, 1, 1)))).astype(np.float32), v_pw=(rng.standard_normal((M, K, 1)) * K ** -0.5).astype(np.float32),
                 b=(0.1 * rng.standard_normal(M)).astype(np.float32))
        ref = OT.up_backward(x, 0.7071068, p["g_ct"], p["v_ct"], p["g_pw"], p["v_pw"], p["b"], dy)
        u = TrainUp(K, M, r)
        pt = {k: _cu(v) for k, v in p.items()}
        assert rel(u.forward(_cu(x), pt, 0.7071068), ref["y"]) <= 2e-5
        g, g2 = u.backward(_cu(x), pt, 0.7071068, _cu(dy)), u.backward(_cu(x), pt, 0.7071068, _cu(dy))
        for k in keys:
            assert rel(g[k], ref[k]) <= 1e-4 and torch.equal(g[k], g2[k]), (K, k, rel(g[k], ref[k]))


@pytest.mark.parametrize("B,C,Tin,T,ks", [(4, 96, 16000, 16000, 5), (2, 96, 16320, 16001, 5), (3, 8, 40, 37, 7),
                                            # the four-samples-per-thread backward with decoder samples past the clip (no gradient there)
                                            (2, 96, 16320, 16000, 5), (2, 16, 48, 40, 5)])
def test_decoder_tail_vs_torch_autograd(B, C, Tin, T, ks):
    """Scale(post) -> ELU -> causal weight-normed Conv1d(C, 1, ks) -> Scale(wav_std) -> Tanh -> [:T] (seanet.py:1166-1204) written with
    torch.nn.functional in float64 (the reference builds exactly these torch ops; SConv1d's causal padding = left pad ks-1)."""
    from waveverify_amd.train import TrainTail
    import torch.nn.functional as F
    torch.manual_seed(C + ks)
    post, wav_std = 0.7071068, 0.1122080159
    g = (0.5 + torch.rand(1, 1, 1, dtype=torch.float64)).requires_grad_(True)
    v = (torch.randn(1, C, ks, dtype=torch.float64) * (C * ks) ** -0.5).requires_grad_(True)
    b = (0.1 * torch.randn(1, dtype=torch.float64)).requires_grad_(True)
    x = torch.randn(B, C, Tin, dtype=torch.float64, requires_grad=True)
    w = g * v / v.flatten(1).norm(dim=1).view(-1, 1, 1)
    delta = torch.tanh(wav_std * F.conv1d(F.pad(F.elu(x * post), (ks - 1, 0)), w, b))[..., :T]
    dd = torch.randn_like(delta)
    delta.backward(dd)
    u = TrainTail(C, ks)
    p = dict(g=g.detach().float().cuda(), v=v.detach().float().cuda(), b=b.detach().float().cuda())
    got = u.forward(x.detach().float().cuda(), p, post, wav_std, T)
    assert rel(got, delta.detach().numpy()) <= 2e-5
    gr = u.backward(x.detach().float().cuda(), p, post, wav_std, got, dd.float().cuda())
    gr2 = u.backward(x.detach().float().cuda(), p, post, wav_std, got, dd.float().cuda())
    for k, ref in (("dx", x.grad), ("dg", g.grad), ("dv", v.grad), ("db", b.grad)):
        assert rel(gr[k], ref.numpy()) <= 1e-4, (k, rel(gr[k], ref.numpy()))
        assert torch.equal(gr[k], gr2[k])
