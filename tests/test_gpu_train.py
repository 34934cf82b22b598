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


def test_training_slice_rejects_unsupported_shapes():
    from waveverify_amd.train import TrainHalf
    half = TrainHalf(64)
    x = torch.randn(1, 64, 10, device="cuda")                     # T % 4 != 0
    p = dict(g_pw=torch.ones(64, device="cuda"), v_pw=torch.randn(64, 64, device="cuda"), g_dw=torch.ones(64, device="cuda"),
             v_dw=torch.randn(64, 5, device="cuda"), b_dw=torch.zeros(64, device="cuda"))
    with pytest.raises(RuntimeError, match="T % 4"):
        half.forward(x, p, 1.0)
