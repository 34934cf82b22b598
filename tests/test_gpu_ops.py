"""Kernel-level parity: every fused unit through the C ABI (wv_op_*) against the numpy oracle
on the same seeded inputs.  float32 everywhere; tolerance 2e-5 * max(1, |ref|max) (different
summation order of the f32 MFMA fmaf chain vs BLAS)."""
import numpy as np
import pytest
import torch

from oracle import wv_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from waveverify_amd import ops as _ops
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return _ops


def rnd(rng, *shape, scale=1.0):
    return (scale * rng.standard_normal(shape)).astype(np.float32)


def close(got, ref, tol=2e-5, what=""):
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    err = float(np.abs(got - ref).max()) if ref.size else 0.0
    lim = tol * max(1.0, float(np.abs(ref).max()) if ref.size else 1.0)
    assert np.isfinite(got).all(), what
    assert err <= lim, f"{what}: max|d|={err:.3e} > {lim:.3e}"


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


# (K, M, Tin, ks, stride, dil) — resblock halves, all four downsample shapes, ragged sizes
PW_DW_CASES = [
    (64, 64, 1000, 5, 1, 1), (128, 128, 777, 5, 1, 1), (96, 96, 300, 5, 1, 1),
    (192, 192, 130, 5, 1, 2), (512, 512, 50, 5, 1, 1), (8, 8, 67, 5, 1, 1), (5, 7, 1, 3, 1, 1),
    (64, 128, 1001, 4, 2, 1), (128, 256, 500, 8, 4, 1), (256, 512, 203, 10, 5, 1),
    (512, 1024, 400, 16, 8, 1), (32, 64, 37, 16, 8, 1), (768, 768, 400, 5, 1, 1),
]


@pytest.mark.parametrize("K,M,Tin,ks,stride,dil", PW_DW_CASES)
@pytest.mark.parametrize("epi", ["none", "resid", "film"])
def test_pw_dw(ops, K, M, Tin, ks, stride, dil, epi):
    if epi == "film" and M % 4:
        pytest.skip("FiLM needs channels divisible by the band count")
    if epi == "resid" and stride != 1:
        pytest.skip("residual only on stride-1 units")
    rng = np.random.default_rng(K * 7 + M + Tin)
    B = 3
    X = rnd(rng, B, K, Tin)
    w_pw = rnd(rng, M, K, 1, scale=K ** -0.5)
    w_dw = rnd(rng, M, 1, ks, scale=ks ** -0.5)
    b_dw = rnd(rng, M, scale=0.1)
    pre = 0.8660254
    h = O.sconv1d(O.elu(X * np.float32(pre)), w_pw, None)
    ref = O.sconv1d(h, w_dw, b_dw, stride=stride, dilation=dil, groups=M)
    kw = {}
    if epi == "resid":
        R = rnd(rng, *ref.shape)
        ref = ref * np.float32(0.37) + R
        kw = dict(resid=cu(R), out_scale=0.37)
    elif epi == "film":
        film = rnd(rng, B, 4, 2)
        bw = M // 4
        ref = ref * np.repeat(film[:, :, 0], bw, 1)[:, :, None] + np.repeat(film[:, :, 1], bw, 1)[:, :, None]
        kw = dict(film=cu(film), bands=4)
    got = ops.pw_dw(cu(X), w_pw, w_dw, b_dw, stride=stride, dilation=dil, pre_scale=pre, pre_elu=True, **kw)
    close(got, ref.astype(np.float32), what=f"pw_dw {epi}")


# ---- the LDS-DMA core (csrc/wv_k1.hip): M >= 128, lengths that are multiples of 4 ----------------
# (K, M, Tin, ks, stride, dil): both window widths (128 / 64 columns), every tile-edge length, M and K
# that are not multiples of the tile / chunk sizes, every strided stencil of the net.
DMA_CASES = [
    (128, 128, 8, 5, 1, 1), (128, 128, 4, 5, 1, 1), (256, 256, 60, 5, 1, 1), (256, 256, 64, 5, 1, 1),
    (128, 128, 124, 5, 1, 1), (128, 128, 128, 5, 1, 1), (128, 256, 132, 5, 1, 1), (384, 384, 2000, 5, 1, 1),
    (100, 130, 252, 5, 1, 1), (33, 128, 400, 5, 1, 1), (513, 1024, 52, 5, 1, 1), (768, 768, 400, 5, 1, 1),
    (160, 288, 1000, 5, 1, 2), (128, 128, 500, 3, 1, 1),
    (64, 128, 1000, 4, 2, 1), (128, 256, 8000, 8, 4, 1), (256, 512, 2000, 10, 5, 1), (512, 1024, 400, 16, 8, 1),
    (256, 512, 204, 10, 5, 1), (48, 136, 36, 16, 8, 1),
]


@pytest.mark.parametrize("K,M,Tin,ks,stride,dil", DMA_CASES)
@pytest.mark.parametrize("mode", ["copy", "copy+resid+act", "elu", "elu+film+act"])
def test_pw_dw_dma_core(ops, K, M, Tin, ks, stride, dil, mode):
    """pre_elu = 0: operand staged by LDS-DMA (pure copy); pre_elu = 1: through registers.  With
    act_scale the epilogue also writes ELU(act_scale * y) (the next unit's hoisted prologue)."""
    if "resid" in mode and stride != 1:
        pytest.skip("residual only on stride-1 units")
    if "film" in mode and M % 4:
        pytest.skip("FiLM needs channels divisible by the band count")
    rng = np.random.default_rng(K * 11 + M + Tin + len(mode))
    B = 2
    X = rnd(rng, B, K, Tin)
    w_pw = rnd(rng, M, K, 1, scale=K ** -0.5)
    w_dw = rnd(rng, M, 1, ks, scale=ks ** -0.5)
    b_dw = rnd(rng, M, scale=0.1)
    elu = mode.startswith("elu")
    pre = 0.8660254 if elu else 1.0
    h = O.sconv1d(O.elu(X * np.float32(pre)) if elu else X, w_pw, None)
    ref = O.sconv1d(h, w_dw, b_dw, stride=stride, dilation=dil, groups=M)
    kw = {}
    if "resid" in mode:
        R = rnd(rng, *ref.shape)
        ref = ref * np.float32(0.37) + R
        kw.update(resid=cu(R), out_scale=0.37)
    if "film" in mode:
        film = rnd(rng, B, 4, 2)
        bw = M // 4
        ref = ref * np.repeat(film[:, :, 0], bw, 1)[:, :, None] + np.repeat(film[:, :, 1], bw, 1)[:, :, None]
        kw.update(film=cu(film), bands=4)
    ref = ref.astype(np.float32)
    if "act" in mode:
        got, gact = ops.pw_dw(cu(X), w_pw, w_dw, b_dw, stride=stride, dilation=dil, pre_scale=pre, pre_elu=elu,
                              act_scale=0.7071, **kw)
        close(gact, O.elu(ref * np.float32(0.7071)), what=f"pw_dw {mode} (activated copy)")
    else:
        got = ops.pw_dw(cu(X), w_pw, w_dw, b_dw, stride=stride, dilation=dil, pre_scale=pre, pre_elu=elu, **kw)
    close(got, ref, what=f"pw_dw {mode}")


@pytest.mark.parametrize("K,M,Tin,r", [(1536, 768, 52, 8), (384, 256, 332, 4), (192, 128, 1000, 2),
                                       (768, 384, 400, 5), (66, 130, 64, 3), (96, 128, 4, 4)])
@pytest.mark.parametrize("pre_elu", [True, False])
def test_upsample_convtr_pw_dma_core(ops, K, M, Tin, r, pre_elu):
    """The upsample unit on the LDS-DMA core (weights by DMA, ConvTranspose producer through registers),
    also with a pre-activated input (pre_elu = 0) and the second, activated output."""
    rng = np.random.default_rng(K + r + 100)
    X = rnd(rng, 2, K, Tin)
    w_ct = rnd(rng, K, 1, 2 * r, scale=(2 * r) ** -0.5)
    w_pw = rnd(rng, M, K, 1, scale=K ** -0.5)
    b = rnd(rng, M, scale=0.1)
    up = O.sconvtr1d_depthwise(O.elu(X * np.float32(0.7071)) if pre_elu else X, w_ct, r)
    ref = O.sconv1d(up, w_pw, b)
    got, gact = ops.dw_pw(cu(X), w_pw, b, w_ct, mode=2, ks_or_ratio=r, pre_scale=0.7071 if pre_elu else 1.0,
                          pre_elu=pre_elu, act_scale=0.9)
    close(got, ref, what="upsample")
    close(gact, O.elu(ref * np.float32(0.9)), what="upsample (activated copy)")


@pytest.mark.parametrize("C,T", [(64, 16000), (64, 244), (64, 248), (64, 4), (64, 240), (96, 1000), (96, 236), (96, 240), (96, 128),
                                 (128, 8000), (128, 244), (128, 252), (128, 488), (192, 8000), (192, 116), (192, 120), (192, 56),
                                 (64, 112), (128, 492), (96, 8), (192, 12)])
def test_fused_resblock(ops, C, T):
    """Whole ResnetBlock in one launch, raw in / raw out (wv_rb.hip: x activated on its way into LDS, u stays in LDS) vs the
    oracle's two-unit composition; lengths around every tile edge (244 / 236 / 116 outputs per tile, 8-column halo, first-tile
    zero padding of both convs), both outputs, and bit-equality with the block run as two pw_dw units."""
    rng = np.random.default_rng(C + T)
    B = 3
    X = rnd(rng, B, C, T)
    w1, w2 = rnd(rng, C, C, 1, scale=C ** -0.5), rnd(rng, C, C, 1, scale=C ** -0.5)
    d1, d2 = rnd(rng, C, 1, 5, scale=0.45), rnd(rng, C, 1, 5, scale=0.45)
    b1, b2 = rnd(rng, C, scale=0.1), rnd(rng, C, scale=0.1)
    pre, s_out, s_act = np.float32(0.8660254), np.float32(0.41), np.float32(0.7071)
    xa = O.elu(X * pre)
    u = O.sconv1d(O.sconv1d(xa, w1, None), d1, b1, groups=C)
    y = X + s_out * O.sconv1d(O.sconv1d(O.elu(u), w2, None), d2, b2, groups=C)
    Xd = cu(X)
    got, gact = ops.resblock(Xd, w1, d1, b1, w2, d2, b2, pre_scale=float(pre), out_scale=float(s_out), act_scale=float(s_act))
    close(got, y.astype(np.float32), what="fused resblock")
    close(gact, O.elu(y.astype(np.float32) * s_act), what="fused resblock (activated copy)")
    only_raw = ops.resblock(Xd, w1, d1, b1, w2, d2, b2, pre_scale=float(pre), out_scale=float(s_out))
    only_act = ops.resblock(Xd, w1, d1, b1, w2, d2, b2, pre_scale=float(pre), out_scale=float(s_out), act_scale=float(s_act), want_raw=False)
    assert torch.equal(only_raw, got) and torch.equal(only_act, gact)
    # the same block as two K1 launches (self-activating first unit): the fused kernel keeps their arithmetic order
    _, ua = ops.pw_dw(Xd, w1, d1, b1, pre_scale=float(pre), pre_elu=True, act_scale=1.0)
    two, two_act = ops.pw_dw(ua, w2, d2, b2, resid=Xd, pre_elu=False, out_scale=float(s_out), act_scale=float(s_act))
    assert torch.equal(two, got), f"fused block differs from two launches: {float((two - got).abs().max()):.3e}"
    assert torch.equal(two_act, gact)


@pytest.mark.parametrize("B,C,T", [(6, 128, 400), (7, 128, 36), (5, 96, 200), (9, 64, 60), (3, 256, 2000), (16, 192, 12), (4, 384, 124)])
@pytest.mark.parametrize("mode", ["copy", "copy+resid+act"])
def test_pw_dw_flat_clip_time_tiling(ops, B, C, T, mode):
    """Flattened (clip, time) tiling of the k5 units on the LDS-DMA core: tiles run across clip boundaries over the
    per-clip padded axis (several short clips per tile, a clip split over many tiles).  The causal zero padding in
    front of EVERY clip and the outputs of every clip must come out exactly as with per-clip tiles."""
    rng = np.random.default_rng(B * 100 + C + T)
    X = rnd(rng, B, C, T)
    w_pw = rnd(rng, C, C, 1, scale=C ** -0.5)
    w_dw = rnd(rng, C, 1, 5, scale=0.45)
    b_dw = rnd(rng, C, scale=0.1)
    ref = O.sconv1d(O.sconv1d(X, w_pw, None), w_dw, b_dw, groups=C)
    kw = {}
    if "resid" in mode:
        R = rnd(rng, *ref.shape)
        ref = ref * np.float32(0.37) + R
        kw.update(resid=cu(R), out_scale=0.37)
    ref = ref.astype(np.float32)
    if "act" in mode:
        got, gact = ops.pw_dw(cu(X), w_pw, w_dw, b_dw, pre_elu=False, act_scale=0.7071, **kw)
        close(gact, O.elu(ref * np.float32(0.7071)), what="flat tiling (activated copy)")
    else:
        got = ops.pw_dw(cu(X), w_pw, w_dw, b_dw, pre_elu=False, **kw)
    close(got, ref, what="flat tiling")


@pytest.mark.parametrize("B,K,M,Tin", [(7, 64, 128, 400), (5, 96, 512, 400), (3, 512, 1024, 400), (16, 64, 256, 56), (2, 64, 128, 8), (9, 64, 128, 1000)])
@pytest.mark.parametrize("mode", ["plain", "film+act"])
def test_downsample_r8_flat_tiling(ops, B, K, M, Tin, mode):
    """The r = 8 downsample unit (1x1 -> depth-wise k16 / stride 8 -> FiLM) with flat tiles over the padded input axis: tiles span
    clip boundaries, every clip's junk output (the one straddling the next clip's pad) is dropped, FiLM scalars come per lane."""
    rng = np.random.default_rng(B + K + M + Tin)
    X = rnd(rng, B, K, Tin)
    w_pw = rnd(rng, M, K, 1, scale=K ** -0.5)
    w_dw = rnd(rng, M, 1, 16, scale=0.25)
    b_dw = rnd(rng, M, scale=0.1)
    ref = O.sconv1d(O.sconv1d(X, w_pw, None), w_dw, b_dw, stride=8, groups=M)
    kw = {}
    if "film" in mode:
        film = rnd(rng, B, 4, 2)
        bw = M // 4
        ref = ref * np.repeat(film[:, :, 0], bw, 1)[:, :, None] + np.repeat(film[:, :, 1], bw, 1)[:, :, None]
        kw.update(film=cu(film), bands=4)
    ref = ref.astype(np.float32)
    if "act" in mode:
        got, gact = ops.pw_dw(cu(X), w_pw, w_dw, b_dw, stride=8, pre_elu=False, act_scale=0.7071, **kw)
        close(gact, O.elu(ref * np.float32(0.7071)), what="r8 flat (activated copy)")
    else:
        got = ops.pw_dw(cu(X), w_pw, w_dw, b_dw, stride=8, pre_elu=False, **kw)
    close(got, ref, what="r8 flat")


def test_pw_dw_no_prologue_no_bias(ops):
    """decoder head: 1x1 (128->1536, no bias) -> DW k5 (seanet.py:1070-1091)."""
    rng = np.random.default_rng(5)
    X = rnd(rng, 2, 128, 50)
    w_pw = rnd(rng, 1536, 128, 1, scale=128 ** -0.5)
    w_dw = rnd(rng, 1536, 1, 5)
    ref = O.sconv1d(O.sconv1d(X, w_pw, None), w_dw, None, groups=1536)
    close(ops.pw_dw(cu(X), w_pw, w_dw, None, pre_elu=False), ref, what="dec head")


@pytest.mark.parametrize("K,M,Tin,r", [(1536, 768, 50, 8), (768, 384, 400, 5), (384, 192, 333, 4),
                                       (192, 96, 1000, 2), (16, 8, 9, 2), (24, 12, 1, 3),
                                       # every addressing mode of the ConvTranspose loader, ragged and tiny shapes
                                       (40, 24, 37, 1), (18, 130, 65, 3), (32, 64, 31, 6), (64, 32, 3, 8),
                                       (20, 20, 129, 4), (12, 100, 2, 2), (8, 8, 200, 7)])
def test_upsample_convtr_pw(ops, K, M, Tin, r):
    rng = np.random.default_rng(K + r)
    X = rnd(rng, 2, K, Tin)
    w_ct = rnd(rng, K, 1, 2 * r, scale=(2 * r) ** -0.5)
    w_pw = rnd(rng, M, K, 1, scale=K ** -0.5)
    b = rnd(rng, M, scale=0.1)
    up = O.sconvtr1d_depthwise(O.elu(X * np.float32(0.7071)), w_ct, r)
    assert up.shape[-1] == Tin * r
    ref = O.sconv1d(up, w_pw, b)
    got = ops.dw_pw(cu(X), w_pw, b, w_ct, mode=2, ks_or_ratio=r, pre_scale=0.7071, pre_elu=True)
    close(got, ref, what="upsample")


@pytest.mark.parametrize("K,M,Tin", [(1024, 128, 50), (128, 64, 500), (32, 16, 17), (64, 128, 51)])
def test_conv_post_l2norm(ops, K, M, Tin):
    rng = np.random.default_rng(K + M)
    X = rnd(rng, 3, K, Tin)
    w_dw = rnd(rng, K, 1, 5, scale=0.4)
    w_pw = rnd(rng, M, K, 1, scale=K ** -0.5)
    b = rnd(rng, M)
    h = O.sconv1d(O.sconv1d(O.elu(X), w_dw, None, groups=K), w_pw, b)
    nrm = np.sqrt((h ** 2).sum(1, keepdims=True))
    ref = (h / np.maximum(nrm, 1e-12) * np.float32(M ** 0.5)).astype(np.float32)
    got = ops.dw_pw(cu(X), w_pw, b, w_dw, mode=1, ks_or_ratio=5, pre_elu=True, l2norm=True)
    close(got, ref, what="conv_post")


@pytest.mark.parametrize("F,C,T", [(33, 64, 1000), (513, 1024, 50), (129, 256, 300), (9, 8, 33),
                                   (65, 128, 8000), (257, 512, 401), (17, 128, 3)])
def test_pointwise_accumulate(ops, F, C, T):
    """SpecBlock tail: x += scale * (W @ P) (seanet.py:497-505)."""
    rng = np.random.default_rng(F)
    P = rnd(rng, 2, F, T)
    Xa = rnd(rng, 2, C, T)
    w = rnd(rng, C, F, 1, scale=F ** -0.5)
    ref = Xa + np.float32(0.61) * O.sconv1d(P, w, None)
    acc = cu(Xa)
    ops.dw_pw(cu(P), w, None, None, mode=0, accumulate_into=acc, out_scale=0.61)
    close(acc, ref.astype(np.float32), what="spec add")


@pytest.mark.parametrize("n_fft,hop,T", [(64, 1, 1000), (128, 2, 1001), (256, 8, 4000), (512, 40, 4001),
                                         (1024, 320, 16000), (16, 1, 5), (32, 4, 1), (256, 32, 777),
                                         # frame counts that are multiples of the vector width: the LDS-DMA core
                                         (128, 2, 1000), (192, 4, 800), (16, 1, 8), (32, 4, 1024), (512, 40, 4000),
                                         (64, 2, 2024), (1024, 64, 4096), (256, 16, 16000),
                                         # the nets' finer scales at full clip length, tile edges, ragged waveforms
                                         (64, 1, 16000), (64, 1, 132), (64, 1, 260), (128, 2, 16000), (128, 2, 263), (128, 4, 16000), (128, 4, 1042),
                                         (256, 8, 16000), (256, 8, 4000), (256, 8, 1049)])
def test_stft_logmag(ops, n_fft, hop, T):
    rng = np.random.default_rng(n_fft + hop)
    wav = np.clip(rnd(rng, 2, 1, T, scale=0.1), -1, 1)
    wav[1, 0, : T // 3] = 0.0                      # silence -> exercises both clamps
    mag = O.causal_stft_mag(wav, n_fft, hop)
    ref = ((np.log(np.maximum(mag, np.float32(1e-5))) - np.float32(-4.3)) / np.float32(2.8)).astype(np.float32)
    got = ops.stft_logmag(cu(wav), n_fft, hop, mean=-4.3, std=2.8)
    # log() amplifies relative error of tiny magnitudes; compare in the log domain with a loose
    # absolute bound where mag is near the 1e-5 clamp, tight elsewhere
    got_np = got.cpu().numpy()
    assert got_np.shape == ref.shape
    big, mid = mag > 1e-2, (mag > 1e-3) & (mag <= 1e-2)
    assert np.abs(got_np - ref)[big].max(initial=0) <= 2e-5 * max(1.0, np.abs(ref).max())
    assert np.abs(got_np - ref)[mid].max(initial=0) <= 1e-4      # |d log m| = |dm| / m: f32 sums of n_fft terms over magnitudes of 1e-3 .. 1e-2
    assert np.abs(got_np - ref)[mag <= 1e-3].max(initial=0) <= 5e-3


@pytest.mark.parametrize("n_fft,hop,T", [(64, 1, 16000), (64, 1, 132), (64, 1, 1000), (64, 1, 68), (64, 1, 256), (64, 2, 2024), (64, 4, 1024),
                                         (128, 2, 16000), (128, 2, 264), (128, 2, 2000), (128, 1, 516), (128, 4, 1040), (128, 2, 263)])
def test_spec_block_in_one_launch(ops, n_fft, hop, T):
    """Whole SpecBlock in one launch (STFT -> log-magnitude -> 1x1 -> add; the spectrogram stays in LDS) for the scales whose spectrum
    is one tile (n_fft = C in {64, 128}): against the oracle's composition (silence in one clip: both clamps), every output mode, in
    place, tile-edge frame counts (128-frame tiles; ragged waveform lengths), and bit-equal to the two-kernel path it replaces."""
    rng = np.random.default_rng(n_fft + hop + T)
    C, F, Tf = n_fft, n_fft // 2 + 1, -(-T // hop)
    wav = np.clip(rnd(rng, 3, 1, T, scale=0.1), -1, 1)
    wav[1, 0, : T // 3] = 0.0
    wav[2] *= 8.0
    x = rnd(rng, 3, C, Tf)
    w = rnd(rng, C, F, 1, scale=F ** -0.5)
    s_out, s_act = np.float32(0.53), np.float32(0.7071)
    mag = O.causal_stft_mag(wav, n_fft, hop)
    P = ((np.log(np.maximum(mag, np.float32(1e-5))) - np.float32(-4.3)) / np.float32(2.8)).astype(np.float32)
    ref = (x + s_out * O.sconv1d(P, w, None)).astype(np.float32)
    wd, xd = cu(wav), cu(x)
    got, gact = ops.spec_block(wd, w, xd, n_fft, hop, mean=-4.3, std=2.8, out_scale=float(s_out), act_scale=float(s_act))
    tol = 2e-5 * max(1.0, float(np.abs(ref).max())) + 5e-3 * float(s_out) * float(np.abs(w).sum(1).max()) * float((mag <= 1e-3).any())
    assert np.isfinite(got.cpu().numpy()).all() and float(np.abs(got.cpu().numpy() - ref).max()) <= tol
    assert float(np.abs(gact.cpu().numpy() - O.elu(ref * s_act)).max()) <= tol
    assert torch.equal(ops.spec_block(wd, w, xd, n_fft, hop, mean=-4.3, std=2.8, out_scale=float(s_out)), got)
    assert torch.equal(ops.spec_block(wd, w, xd, n_fft, hop, mean=-4.3, std=2.8, out_scale=float(s_out), act_scale=float(s_act), want_raw=False), gact)
    # the two kernels it replaces: STFT -> P in HBM -> the SpecBlock add (K1, identity stencil)
    Pd = ops.stft_logmag(wd, n_fft, hop, mean=-4.3, std=2.8)
    two = xd.clone()
    if C >= 128:                                   # the op hands out the activated copy on its K1 form (M >= 128) only
        _, two_act = ops.dw_pw(Pd, w, None, None, mode=0, accumulate_into=two, out_scale=float(s_out), act_scale=float(s_act))
        assert torch.equal(two_act, gact)
        assert torch.equal(two, got), f"{float((two - got).abs().max()):.3e}"
    else:                                          # below 128 rows the op's add is the round-1 1x1 kernel: same sums, another rounding order
        ops.dw_pw(Pd, w, None, None, mode=0, accumulate_into=two, out_scale=float(s_out))
        assert float((two - got).abs().max()) <= 2e-6 * max(1.0, float(got.abs().max()))


def test_spec_block_refuses_other_shapes(ops):
    wav, x = torch.zeros(1, 1, 4000).cuda(), torch.zeros(1, 256, 500).cuda()
    with pytest.raises(RuntimeError):
        ops.spec_block(wav, np.zeros((256, 129, 1), np.float32), x, 256, 8)          # the spectrum spans several tiles
    with pytest.raises(RuntimeError):
        ops.spec_block(torch.zeros(1, 1, 60).cuda(), np.zeros((64, 33, 1), np.float32), torch.zeros(1, 64, 60).cuda(), 64, 1)   # <= 64 frames


def test_stft_with_given_basis(ops):
    rng = np.random.default_rng(3)
    wav = rnd(rng, 2, 1, 500, scale=0.1)
    basis = O.dft_basis(64) * np.float32(1.01)
    mag = O.causal_stft_mag(wav, 64, 1, basis)
    ref = np.log(np.maximum(mag, np.float32(1e-5))).astype(np.float32)
    got = ops.stft_logmag(cu(wav), 64, 1, basis=basis).cpu().numpy()
    big = mag > 1e-3
    assert np.abs(got - ref)[big].max() <= 2e-5 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("C,T,ks", [(64, 16000, 5), (32, 1001, 5), (8, 3, 7), (4, 1, 5)])
def test_conv_pre(ops, C, T, ks):
    rng = np.random.default_rng(C)
    x = rnd(rng, 3, 1, T, scale=0.1)
    w = rnd(rng, C, 1, ks)
    b = rnd(rng, C)
    ref = O.sconv1d((x * np.float32(8.912)).astype(np.float32), w, b)
    close(ops.conv_pre(cu(x), w, b, 8.912), ref, what="conv_pre")


@pytest.mark.parametrize("C,Tin,T,ks", [(96, 16000, 16000, 5), (96, 16320, 16001, 5), (8, 68, 67, 5), (8, 4, 1, 5)])
def test_tail(ops, C, Tin, T, ks):
    rng = np.random.default_rng(C + T)
    H = rnd(rng, 2, C, Tin)
    x = rnd(rng, 2, 1, T, scale=0.1)
    w = rnd(rng, 1, C, ks, scale=(C * ks) ** -0.5)
    b = rnd(rng, 1)
    y = O.sconv1d(O.elu(H * np.float32(0.7071)), w, b)
    ref = (np.tanh(y * np.float32(0.1122))[..., :T] + x).astype(np.float32)
    close(ops.tail(cu(H), w, b, cu(x), T=T, pre_scale=0.7071, out_scale=0.1122), ref, 1e-6, "tail")
    ref2 = np.tanh(y * np.float32(0.1122))[..., :T].astype(np.float32)
    close(ops.tail(cu(H), w, b, None, T=T, pre_scale=0.7071, out_scale=0.1122), ref2, 1e-6, "tail no add")


@pytest.mark.parametrize("D,O_,nb,hop,Fr,T", [(128, 32, 16, 320, 50, 16000), (128, 32, 16, 320, 51, 16001),
                                               (64, 32, 1, 32, 500, 16000), (16, 8, 16, 4, 17, 67), (8, 8, 1, 4, 1, 1)])
def test_head(ops, D, O_, nb, hop, Fr, T):
    rng = np.random.default_rng(D + hop)
    Z = rnd(rng, 2, D, Fr)
    sd = {"reverse_convolution.weight": rnd(rng, D, O_, hop, scale=D ** -0.5),
          "reverse_convolution.bias": rnd(rng, O_, scale=0.1),
          "last_layer.weight": rnd(rng, nb, O_, 1, scale=O_ ** -0.5),
          "last_layer.bias": rnd(rng, nb)}
    net = O._Net(None, sd)
    ref = O.head_forward(net, Z, T)
    logits, mean = ops.head(cu(Z), sd["reverse_convolution.weight"], sd["reverse_convolution.bias"],
                            sd["last_layer.weight"], sd["last_layer.bias"], T)
    close(logits, ref, what="head logits")
    close(mean, O.mean_probabilities(ref), 2e-6, "mean prob")
    _, mean_only = ops.head(cu(Z), sd["reverse_convolution.weight"], sd["reverse_convolution.bias"],
                            sd["last_layer.weight"], sd["last_layer.bias"], T, want_logits=False)
    assert torch.equal(mean, mean_only)            # fused reduction is deterministic


