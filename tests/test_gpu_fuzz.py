"""Seeded shape fuzzing of the fused units against the numpy oracle: geometry corners the layer
shapes of the shipped nets never reach (odd channel counts, T around tile edges, every stride /
dilation / kernel-size combination the tile geometry has a branch for)."""
import numpy as np
import pytest
import torch

from oracle import wv_oracle as O
from test_gpu_ops import close, cu, rnd

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from waveverify_amd import ops as _ops
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return _ops


def _pw_dw_cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    edges = [1, 2, 3, 4, 5, 59, 60, 61, 63, 64, 65, 123, 124, 125, 127, 128, 129, 247, 248, 249, 300]
    while len(out) < n:
        stride = int(rng.choice([1, 1, 1, 2, 3, 4, 5, 8]))
        if stride == 1:
            ks, dil = int(rng.choice([1, 2, 3, 5, 5, 5, 7])), int(rng.choice([1, 1, 2, 3]))
        else:
            ks, dil = 2 * stride, 1
        if (ks - 1) * dil + 4 > 64:
            continue
        K = int(rng.choice([1, 3, 8, 17, 32, 33, 64, 80, 96, 100, 128, 130, 192, 200]))
        M = int(rng.choice([1, 4, 8, 31, 32, 33, 64, 65, 96, 97, 128, 129, 160, 192, 256, 260]))
        T = int(rng.choice(edges)) * (1 if stride == 1 else int(rng.choice([1, stride])))
        out.append((K, M, max(T, 1), ks, stride, dil, int(rng.integers(1, 4))))
    return out


@pytest.mark.parametrize("K,M,Tin,ks,stride,dil,B", _pw_dw_cases(60, 2024))
def test_pw_dw_fuzz(ops, K, M, Tin, ks, stride, dil, B):
    rng = np.random.default_rng(K * 131 + M * 17 + Tin + ks)
    X = rnd(rng, B, K, Tin)
    w_pw = rnd(rng, M, K, 1, scale=K ** -0.5)
    w_dw = rnd(rng, M, 1, ks, scale=ks ** -0.5)
    b_dw = rnd(rng, M, scale=0.1)
    pre_elu = bool(rng.integers(0, 2))
    pre = float(rng.uniform(0.5, 1.0))
    xin = X * np.float32(pre)
    h = O.sconv1d(O.elu(xin) if pre_elu else xin, w_pw, None)
    ref = O.sconv1d(h, w_dw, b_dw, stride=stride, dilation=dil, groups=M)
    kw = {}
    mode = int(rng.integers(0, 3))
    if mode == 1 and stride == 1:
        R = rnd(rng, *ref.shape)
        ref = ref * np.float32(0.61) + R
        kw = dict(resid=cu(R), out_scale=0.61)
    elif mode == 2 and M % 4 == 0:
        film = rnd(rng, B, 4, 2)
        bw = M // 4
        ref = ref * np.repeat(film[:, :, 0], bw, 1)[:, :, None] + np.repeat(film[:, :, 1], bw, 1)[:, :, None]
        kw = dict(film=cu(film), bands=4)
    got = ops.pw_dw(cu(X), w_pw, w_dw, b_dw, stride=stride, dilation=dil, pre_scale=pre, pre_elu=pre_elu, **kw)
    close(got, ref.astype(np.float32), what=f"pw_dw fuzz K={K} M={M} T={Tin} ks={ks} s={stride} d={dil}")


def _up_cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        r = int(rng.integers(1, 9))
        K = int(rng.choice([2, 6, 16, 30, 48, 64, 100, 128]))
        M = int(rng.choice([1, 8, 24, 32, 40, 64, 96, 100, 128, 192]))
        Tin = int(rng.choice([1, 2, 3, 7, 15, 16, 17, 31, 32, 33, 62, 63, 64, 65, 100]))
        out.append((K, M, Tin, r, int(rng.integers(1, 4))))
    return out


@pytest.mark.parametrize("K,M,Tin,r,B", _up_cases(40, 77))
def test_upsample_fuzz(ops, K, M, Tin, r, B):
    rng = np.random.default_rng(K * 19 + M * 3 + Tin * 7 + r)
    X = rnd(rng, B, K, Tin)
    w_ct = rnd(rng, K, 1, 2 * r, scale=(2 * r) ** -0.5)
    w_pw = rnd(rng, M, K, 1, scale=K ** -0.5)
    b = rnd(rng, M, scale=0.1)
    up = O.sconvtr1d_depthwise(O.elu(X * np.float32(0.9)), w_ct, r)
    ref = O.sconv1d(up, w_pw, b)
    got = ops.dw_pw(cu(X), w_pw, b, w_ct, mode=2, ks_or_ratio=r, pre_scale=0.9, pre_elu=True)
    close(got, ref, what=f"upsample fuzz K={K} M={M} T={Tin} r={r}")


@pytest.mark.parametrize("n_fft,hop,T", [(4, 1, 9), (6, 2, 31), (8, 3, 64), (10, 1, 129), (16, 5, 100), (32, 7, 500),
                                         (64, 1, 300), (66, 2, 257), (128, 2, 1000), (130, 9, 999), (256, 8, 2048),
                                         (512, 40, 1601), (1024, 320, 3201), (20, 20, 19), (12, 4, 1)])
def test_stft_fuzz(ops, n_fft, hop, T):
    rng = np.random.default_rng(n_fft * 5 + hop + T)
    x = rnd(rng, 2, 1, T, scale=0.3)
    mag = O.causal_stft_mag(x, n_fft, hop)
    ref = ((np.log(np.maximum(mag, np.float32(1e-5))) - np.float32(-3.0)) / np.float32(2.5)).astype(np.float32)
    got = ops.stft_logmag(cu(x), n_fft, hop, mean=-3.0, std=2.5).cpu().numpy()
    assert got.shape == ref.shape and np.isfinite(got).all()
    big = mag > 1e-3                                  # log() amplifies the error of near-zero bins
    assert np.abs(got - ref)[big].max(initial=0) <= 2e-5 * max(1.0, np.abs(ref).max())
    assert np.abs(got - ref)[~big].max(initial=0) <= 5e-3


def _net_cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    stride_sets = [[2, 2], [3, 2], [2, 3, 2], [5, 2], [4, 4], [8, 2], [2], [7, 3], [6, 5], [2, 2, 2, 2]]
    for i in range(n):
        strides = stride_sets[int(rng.integers(0, len(stride_sets)))]
        cfg = dict(strides=strides,
                   channels_enc=int(rng.choice([4, 8, 12, 16])),
                   channels_dec=int(rng.choice([4, 8, 12])),
                   dimension=int(rng.choice([8, 16, 24, 32])),
                   n_fft_base=int(rng.choice([8, 16, 32])),
                   n_residual_enc=int(rng.integers(1, 4)), n_residual_dec=int(rng.integers(1, 4)),
                   kernel_size=int(rng.choice([3, 5, 7])), last_kernel_size=int(rng.choice([3, 5, 7])),
                   residual_kernel_size=int(rng.choice([3, 5])), dilation_base=int(rng.choice([1, 1, 2])),
                   output_dim=int(rng.choice([4, 8, 16])), embedding_dim=int(rng.choice([8, 16])),
                   embedding_layers=int(rng.integers(1, 4)))
        hop = int(np.prod(strides))
        T = int(rng.choice([1, hop - 1 if hop > 1 else 1, hop, hop + 1, 3 * hop + 2, 257, 640]))
        out.append((i, cfg, max(T, 1), int(rng.integers(1, 4))))
    return out


@pytest.mark.parametrize("idx,cfgkw,T,B", _net_cases(16, 99))
def test_whole_nets_fuzz(idx, cfgkw, T, B):
    """Random (constructible) hyper-parameter sets: generator, detector and locator vs the oracle."""
    from waveverify_amd.config import default_config
    from waveverify_amd.init import random_state_dict, synthetic_clips
    from waveverify_amd.nets import HipNet
    kw = dict(cfgkw)
    nspec = len(kw["strides"]) + 1
    kw["spec_means"] = [-4.0 + 0.1 * i for i in range(nspec)]
    kw["spec_stds"] = [2.5 + 0.05 * i for i in range(nspec)]
    if (2 * kw["channels_enc"]) % 4:
        pytest.skip("FiLM bands")
    x, msg = synthetic_clips(B, T, seed=1000 + idx)
    xt, mt = torch.from_numpy(x).cuda(), torch.from_numpy(msg).cuda()
    cg = default_config("generator", **kw)
    sdg = random_state_dict(cg, 31 + idx, parametrized=bool(idx & 1))
    ref = O.generator_forward(cg, sdg, x, msg)
    got = HipNet(cg, sdg).generator(xt, mt)
    close(got, ref, tol=5e-5, what=f"generator fuzz #{idx} {cfgkw} T={T}")
    for kind in ("detector", "locator"):
        dk = {k: v for k, v in kw.items() if k not in ("channels_dec", "n_residual_dec", "embedding_dim", "embedding_layers")}
        cd = default_config(kind, **dk)
        sdd = random_state_dict(cd, 57 + idx)
        fwd = O.detector_forward if kind == "detector" else O.locator_forward
        refl = fwd(cd, sdd, x)
        net = HipNet(cd, sdd)
        gotl = net.detector(xt) if kind == "detector" else net.locator(xt)
        close(gotl, refl, tol=1e-4, what=f"{kind} fuzz #{idx} {cfgkw} T={T}")
