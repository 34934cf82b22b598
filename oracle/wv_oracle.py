"""CPU oracle: a numpy (float32) restatement of the WaveVerify embed/detect forward path.

TEST INFRASTRUCTURE ONLY.  Nothing under waveverify_amd/ imports this module; it is used by
tests/, by __graft_entry__.smoke() and by bench.py's `cpu_baseline` leg as the *checker*,
never as the thing measured as the product or shipped.

Parity pinning: the reference ships no tests, golden vectors or checkpoint,
so this restatement is pinned against outputs of the reference itself, produced in the build
container by tests/golden/make_golden.py (reference modules imported from /root/reference
with a test-side `audiotools` stub) and committed as tests/golden/*.npz.
tests/test_oracle_golden.py asserts oracle == golden to <= 2e-5 on every tensor.

Every function cites the reference lines it follows (paths relative to /root/reference).
All arithmetic is float32 like the reference; tensors are [B, C, T], time innermost.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import numpy as np

F32 = np.float32


# --------------------------------------------------------------------------- weights
def fold_weight_norm(g: np.ndarray, v: np.ndarray) -> np.ndarray:
    """w = g * v / ||v||, norm over every dim but 0 (modules/conv.py:73-74 ->
    torch.nn.utils.parametrizations.weight_norm, dim=0)."""
    v = np.asarray(v, F32)
    nrm = np.sqrt((v.reshape(v.shape[0], -1) ** 2).sum(axis=1, dtype=F32)).astype(F32)
    scale = (np.asarray(g, F32).reshape(-1) / nrm).astype(F32)
    return (v * scale.reshape((-1,) + (1,) * (v.ndim - 1))).astype(F32)


def fold_state_dict(sd: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """Turn `...parametrizations.weight.original0/1` pairs into plain `...weight` keys, the
    layout checkpoints are saved in (scripts/train.py:1624-1629)."""
    out: Dict[str, np.ndarray] = {}
    tag = "parametrizations.weight.original"
    for k, val in sd.items():
        if tag in k:
            if k.endswith("original0"):
                base = k[: -len("parametrizations.weight.original0")]
                out[base + "weight"] = fold_weight_norm(val, sd[base + tag + "1"])
        else:
            out[k] = np.asarray(val, F32)
    return out


# --------------------------------------------------------------------------- elementwise
def elu(x: np.ndarray) -> np.ndarray:
    """nn.ELU(alpha=1): x if x > 0 else exp(x) - 1."""
    return np.where(x > 0, x, np.expm1(np.minimum(x, 0))).astype(F32)


def sigmoid(x: np.ndarray) -> np.ndarray:
    return (1.0 / (1.0 + np.exp(-x.astype(F32)))).astype(F32)


# --------------------------------------------------------------------------- padding math
def extra_padding_for_conv1d(length: int, kernel_size: int, stride: int, padding_total: int) -> int:
    """modules/conv.py:160-196 (note: uses kernel_size, not the dilated extent)."""
    n_frames = (length - kernel_size + padding_total) / stride + 1
    ideal = (math.ceil(n_frames) - 1) * stride + (kernel_size - padding_total)
    return max(0, ideal - length)


def sconv1d(x: np.ndarray, w: np.ndarray, b: Optional[np.ndarray], stride: int = 1,
            dilation: int = 1, groups: int = 1) -> np.ndarray:
    """Causal SConv1d.forward (modules/conv.py:715-763): left pad (k-1)*d-(s-1), right pad
    `extra` so the last frame is complete, then a plain conv1d."""
    B, Cin, T = x.shape
    Cout, Cin_g, k = w.shape
    pad_total = (k - 1) * dilation - (stride - 1)
    extra = extra_padding_for_conv1d(T, k, stride, pad_total)
    if k == 1 and groups == 1 and stride == 1:
        y = np.matmul(w[:, :, 0], x)                              # pointwise GEMM
    else:
        xp = np.pad(x, ((0, 0), (0, 0), (pad_total, extra)))
        Tout = (xp.shape[-1] - dilation * (k - 1) - 1) // stride + 1
        span = (Tout - 1) * stride + 1
        if groups == Cin and Cin_g == 1 and Cout == Cin:          # depth-wise
            y = np.zeros((B, Cout, Tout), F32)
            for i in range(k):
                y += w[None, :, 0, i, None] * xp[:, :, i * dilation: i * dilation + span: stride]
        elif groups == 1:                                         # dense k-tap conv
            y = np.zeros((B, Cout, Tout), F32)
            for i in range(k):
                y += np.matmul(w[:, :, i], xp[:, :, i * dilation: i * dilation + span: stride])
        else:
            raise NotImplementedError("only pointwise, depth-wise and dense convs occur")
    if b is not None:
        y = y + b[None, :, None]
    return y.astype(F32)


def sconvtr1d_depthwise(x: np.ndarray, w: np.ndarray, stride: int) -> np.ndarray:
    """Causal depth-wise SConvTranspose1d.forward (modules/conv.py:838-881): conv_transpose1d
    then drop the last k - s samples (trim_right_ratio = 1)."""
    B, C, L = x.shape
    k = w.shape[-1]
    full = np.zeros((B, C, (L - 1) * stride + k), F32)
    for i in range(k):
        full[:, :, i: i + (L - 1) * stride + 1: stride] += x * w[None, :, 0, i, None]
    trim = k - stride
    return full[:, :, : full.shape[-1] - trim]


# --------------------------------------------------------------------------- STFT
def hann_window_periodic(n: int) -> np.ndarray:
    """torch.hann_window(n) (periodic): 0.5 - 0.5*cos(2*pi*i/n) in float32."""
    i = np.arange(n, dtype=F32)
    return (F32(0.5) - F32(0.5) * np.cos(i * F32(2.0 * math.pi / n))).astype(F32)


def dft_basis(n_fft: int) -> np.ndarray:
    """[2F, n_fft] windowed cos / sin rows, F = n_fft//2+1 (modules/conv.py:1003-1026).
    The angle is formed in float32 exactly as the reference does, so the Nyquist sine row is
    tiny-but-nonzero like upstream."""
    n = np.arange(n_fft, dtype=F32)[None, :]
    k = np.arange(n_fft // 2 + 1, dtype=F32)[:, None]
    ang = ((F32(-2.0 * math.pi / n_fft) * k).astype(F32) * n).astype(F32)
    w = np.concatenate([np.cos(ang), np.sin(ang)], axis=0).astype(F32)
    return (w * hann_window_periodic(n_fft)[None, :]).astype(F32)


def causal_stft_mag(wav: np.ndarray, n_fft: int, hop: int, basis: Optional[np.ndarray] = None,
                    eps: float = 1e-12) -> np.ndarray:
    """CausalSTFT.forward (modules/conv.py:1036-1080): left-pad n_fft-1 zeros, strided conv
    with the DFT basis, sqrt(max(re^2+im^2, eps)).  wav [B,1,T] -> [B,F,ceil(T/hop)]."""
    if basis is None:
        basis = dft_basis(n_fft)
    B, _, T = wav.shape
    xp = np.pad(wav[:, 0, :], ((0, 0), (n_fft - 1, 0))).astype(F32)
    n_frames = (xp.shape[-1] - n_fft) // hop + 1
    st = xp.strides
    frames = np.lib.stride_tricks.as_strided(
        xp, shape=(B, n_fft, n_frames), strides=(st[0], st[1], st[1] * hop), writeable=False)
    c = np.matmul(basis, frames)                                  # [B, 2F, n_frames]
    Fq = n_fft // 2 + 1
    p = c[:, :Fq] ** 2 + c[:, Fq:] ** 2
    return np.sqrt(np.maximum(p, F32(eps))).astype(F32)


# --------------------------------------------------------------------------- blocks
class _Net:
    """Weights + hyper-parameters; `cfg` is a waveverify_amd.config.NetConfig-like object."""

    def __init__(self, cfg, sd: Dict[str, np.ndarray]):
        self.cfg = cfg
        self.sd = fold_state_dict({k: np.asarray(v) for k, v in sd.items()})

    def w(self, key: str) -> np.ndarray:
        return self.sd[key]

    def opt(self, key: str) -> Optional[np.ndarray]:
        return self.sd.get(key)


def resnet_block(net: _Net, prefix: str, x: np.ndarray, idx: int, res_scale: float,
                 dilations: List[int]) -> np.ndarray:
    """SEANetResnetBlock.forward with skip='identity' (modules/seanet.py:245-281) over
    dws_conv_block x2 (seanet.py:39-116): [ELU, 1x1 (no bias), DW k (bias)] twice."""
    pre_scale = F32((1 + idx * res_scale ** 2) ** -0.5)             # seanet.py:183
    y = x * pre_scale
    for (pw, dw), dil in zip(((1, 2), (4, 5)), dilations):
        y = elu(y)
        y = sconv1d(y, net.w(f"{prefix}.block.{pw}.conv.conv.weight"), None)
        wd = net.w(f"{prefix}.block.{dw}.conv.conv.weight")
        y = sconv1d(y, wd, net.w(f"{prefix}.block.{dw}.conv.conv.bias"), dilation=dil,
                    groups=wd.shape[0])
    scale = F32(res_scale)
    p = net.opt(f"{prefix}.res_scale_param")
    if p is not None:
        scale = F32(scale * p.reshape(-1)[0])                       # seanet.py:272-274
    return (y * scale + x).astype(F32)                               # seanet.py:277


def spec_block(net: _Net, prefix: str, x: np.ndarray, wav: np.ndarray, n_fft: int, hop: int,
               mean: float, std: float, res_scale: float) -> np.ndarray:
    """SpecBlock.forward (modules/seanet.py:463-511)."""
    basis = net.opt(f"{prefix}.spec.weight")
    if basis is not None:
        basis = basis[:, 0, :]
    y = causal_stft_mag(wav, n_fft, hop, basis)
    y = np.log(np.maximum(y, F32(1e-5))).astype(F32)                # :484
    y = ((y - F32(mean)) / F32(std)).astype(F32)                    # :494
    y = sconv1d(y, net.w(f"{prefix}.layer.conv.conv.weight"), None)
    scale = F32(res_scale)
    p = net.opt(f"{prefix}.scale_param")
    if p is not None:
        scale = F32(p.reshape(-1)[0] * scale)                       # :500-502
    return (x + y * scale).astype(F32)                               # :505


def msg_embedding(net: _Net, msg: np.ndarray) -> np.ndarray:
    """encoder.msg_embedding = [Linear, (Linear, ReLU) x embedding_layers]
    (modules/seanet.py:831-839): note no ReLU after the first Linear."""
    cfg = net.cfg
    h = msg.astype(F32)
    h = h @ net.w("encoder.msg_embedding.0.weight").T + net.w("encoder.msg_embedding.0.bias")
    for i in range(cfg.embedding_layers):
        j = 1 + 2 * i
        h = h @ net.w(f"encoder.msg_embedding.{j}.weight").T + net.w(f"encoder.msg_embedding.{j}.bias")
        h = np.maximum(h, 0)
    return h.astype(F32)


def film_params(net: _Net, emb: np.ndarray) -> np.ndarray:
    """gamma/beta of every FiLM layer (modules/seanet.py:518-550,843-846):
    returns [B, n_scales, freq_bands, 2] with [..., 0] = gamma, [..., 1] = beta."""
    cfg = net.cfg
    S = len(cfg.strides)
    out = np.zeros((emb.shape[0], S, cfg.freq_bands, 2), F32)
    for s in range(S):
        for b in range(cfg.freq_bands):
            for j, nm in enumerate(("gamma", "beta")):
                w = net.w(f"encoder.film_layers.{s}.{b}.{nm}_layer.weight")
                bb = net.w(f"encoder.film_layers.{s}.{b}.{nm}_layer.bias")
                out[:, s, b, j] = (emb @ w.T + bb)[:, 0]
    return out


def encoder_forward(net: _Net, x: np.ndarray, msg: Optional[np.ndarray],
                    taps: Optional[dict] = None) -> np.ndarray:
    """SEANetEncoder.forward (modules/seanet.py:883-976)."""
    cfg = net.cfg
    rs = cfg.res_scale_enc
    wav = x
    # conv_pre: Scale(1/wav_std) then SConv1d(1 -> C0, k)          (seanet.py:657-664)
    h = sconv1d((x * F32(1.0 / cfg.wav_std)).astype(F32),
                net.w("encoder.conv_pre.1.conv.conv.weight"),
                net.w("encoder.conv_pre.1.conv.conv.bias"))
    if taps is not None:
        taps["conv_pre"] = h
    film = None
    if msg is not None:
        if msg.shape[0] != x.shape[0]:
            # model/watermarking.py:320-329: a single message broadcasts over the batch
            reps = int(math.ceil(x.shape[0] / msg.shape[0]))
            msg = np.tile(msg, (reps, 1))[: x.shape[0]]
        film = film_params(net, msg_embedding(net, msg))
        if taps is not None:
            taps["film"] = film
    stride = 1
    mult = 1
    for s, ratio in enumerate(cfg.ratios_enc):
        for j in range(1, cfg.n_residual_enc + 1):                  # idx = j (seanet.py:684)
            h = resnet_block(net, f"encoder.blocks.{s}.{j - 1}", h, j, rs,
                             [cfg.dilation_base ** j, 1])
        h = spec_block(net, f"encoder.spec_blocks.{s}", h, wav, mult * cfg.n_fft_base, stride,
                       cfg.spec_means[s], cfg.spec_stds[s], rs)
        if taps is not None:
            taps[f"enc_scale{s}_pre_down"] = h
        stride *= ratio
        # downsample: Scale -> ELU -> 1x1 (C->2C) -> DW strided (k=2r, s=r)   (seanet.py:733-772)
        h = elu(h * F32((1 + cfg.n_residual_enc * rs ** 2) ** -0.5))
        h = sconv1d(h, net.w(f"encoder.downsample.{s}.2.conv.conv.weight"), None)
        wd = net.w(f"encoder.downsample.{s}.3.conv.conv.weight")
        h = sconv1d(h, wd, net.w(f"encoder.downsample.{s}.3.conv.conv.bias"), stride=ratio,
                    groups=wd.shape[0])
        if taps is not None:
            taps[f"enc_scale{s}_down"] = h
        if film is not None:                                        # seanet.py:928-966
            C = h.shape[1]
            bw = C // cfg.freq_bands
            g = np.repeat(film[:, s, :, 0], bw, axis=1)[:, :, None]
            bt = np.repeat(film[:, s, :, 1], bw, axis=1)[:, :, None]
            h = (h * g + bt).astype(F32)
        if taps is not None:
            taps[f"enc_scale{s}_out"] = h
        mult *= 2
    h = spec_block(net, "encoder.spec_post", h, wav, mult * cfg.n_fft_base, stride,
                   cfg.spec_means[-1], cfg.spec_stds[-1], rs)           # seanet.py:789-790: [-1]
    # conv_post: ELU -> DW k (no bias) -> 1x1 (bias) -> L2Norm      (seanet.py:797-823)
    h = elu(h)
    wd = net.w("encoder.conv_post.1.conv.conv.weight")
    h = sconv1d(h, wd, None, groups=wd.shape[0])
    h = sconv1d(h, net.w("encoder.conv_post.2.conv.conv.weight"),
                net.w("encoder.conv_post.2.conv.conv.bias"))
    # L2Norm (seanet.py:288-318): F.normalize(dim=1, eps=1e-12) * sqrt(C)
    nrm = np.sqrt((h ** 2).sum(axis=1, keepdims=True, dtype=F32))
    h = (h / np.maximum(nrm, F32(1e-12)) * F32(h.shape[1] ** 0.5)).astype(F32)
    if taps is not None:
        taps["latent"] = h
    return h


def decoder_layout(cfg):
    """Indices into decoder.model, an nn.Sequential (modules/seanet.py:1067-1204):
    [pw0, dw0] then per ratio [scale|identity, ELU, convtr, pw, resblock x n], then
    [scale, ELU, last conv, scale, tanh]."""
    n = 2
    ups = []
    mult = 2 ** len(cfg.strides)
    for r in cfg.strides:
        ups.append((n + 2, n + 3, [n + 4 + j for j in range(cfg.n_residual_dec)], r,
                    mult * cfg.channels_dec))
        n += 4 + cfg.n_residual_dec
        mult //= 2
    return 0, 1, ups, n + 2


def decoder_forward(net: _Net, z: np.ndarray, taps: Optional[dict] = None) -> np.ndarray:
    """SEANetDecoder.forward (modules/seanet.py:1212-1226; layers built :1067-1204)."""
    cfg = net.cfg
    rs = cfg.res_scale_dec
    i_pw0, i_dw0, ups, i_last = decoder_layout(cfg)
    h = sconv1d(z, net.w(f"decoder.model.{i_pw0}.conv.conv.weight"), None)
    wd = net.w(f"decoder.model.{i_dw0}.conv.conv.weight")
    h = sconv1d(h, wd, net.w(f"decoder.model.{i_dw0}.conv.conv.bias"), groups=wd.shape[0])
    post = F32((1 + cfg.n_residual_dec * rs ** 2) ** -0.5)
    for i, (ct, pw, res, r, C) in enumerate(ups):
        if i > 0:
            h = h * post                                             # seanet.py:1098-1107
        h = elu(h)
        h = sconvtr1d_depthwise(h, net.w(f"decoder.model.{ct}.convtr.convtr.weight"), r)
        h = sconv1d(h, net.w(f"decoder.model.{pw}.conv.conv.weight"),
                    net.w(f"decoder.model.{pw}.conv.conv.bias"))
        for j, ri in enumerate(res):                                 # idx = j (seanet.py:1159)
            h = resnet_block(net, f"decoder.model.{ri}", h, j, rs, [cfg.dilation_base ** j, 1])
        if taps is not None:
            taps[f"dec_scale{i}_out"] = h
    h = elu(h * post)                                                # seanet.py:1177-1179
    h = sconv1d(h, net.w(f"decoder.model.{i_last}.conv.conv.weight"),
                net.w(f"decoder.model.{i_last}.conv.conv.bias"))
    return np.tanh(h * F32(cfg.wav_std)).astype(F32)                 # :1193, final Tanh


# --------------------------------------------------------------------------- nets
def generator_forward(cfg, sd, x: np.ndarray, msg: np.ndarray, taps: Optional[dict] = None
                      ) -> np.ndarray:
    """Generator.forward (model/generator.py:360-423): delta = decode(encode(x, msg))[..., :T]."""
    net = sd if isinstance(sd, _Net) else _Net(cfg, sd)
    T = x.shape[-1]
    z = encoder_forward(net, x.astype(F32), msg, taps)
    return decoder_forward(net, z, taps)[..., :T]


def embed(cfg, sd, x: np.ndarray, msg: np.ndarray) -> np.ndarray:
    """AudioWatermarking._forward_audio_sample (model/watermarking.py:423-441): wm = G(x,msg)+x."""
    return (generator_forward(cfg, sd, x, msg) + x).astype(F32)


def head_forward(net: _Net, z: np.ndarray, T: int) -> np.ndarray:
    """reverse_convolution (ConvTranspose1d k=s=hop) -> trim to T -> last_layer 1x1
    (model/detector.py:300-310, model/locator.py:247-258)."""
    W1 = net.w("reverse_convolution.weight")                         # [D, O, hop]
    b1 = net.w("reverse_convolution.bias")
    B, D, Fr = z.shape
    O, hop = W1.shape[1], W1.shape[2]
    up = np.einsum("bdf,doj->bofj", z, W1, optimize=True).reshape(B, O, Fr * hop)
    up = (up + b1[None, :, None]).astype(F32)[:, :, :T]
    W2 = net.w("last_layer.weight")[:, :, 0]
    return (np.matmul(W2, up) + net.w("last_layer.bias")[None, :, None]).astype(F32)


def detector_forward(cfg, sd, x: np.ndarray, taps: Optional[dict] = None) -> np.ndarray:
    """Detector.forward (model/detector.py:366-391): logits [B, nbits, T]."""
    net = sd if isinstance(sd, _Net) else _Net(cfg, sd)
    z = encoder_forward(net, x.astype(F32), None, taps)
    return head_forward(net, z, x.shape[-1])


def locator_forward(cfg, sd, x: np.ndarray, taps: Optional[dict] = None) -> np.ndarray:
    """Locator.forward (model/locator.py:268-299): logits [B, 1, T]."""
    return detector_forward(cfg, sd, x, taps)


# --------------------------------------------------------------------------- decisions
def mean_probabilities(logits: np.ndarray) -> np.ndarray:
    """sigmoid -> mean over time (waveverify/core.py:577-580)."""
    return sigmoid(logits).mean(axis=2, dtype=F32).astype(F32)


def decide_bits(mean_prob: np.ndarray, threshold: float = 0.5) -> np.ndarray:
    """tensor_to_message (waveverify/utils.py:356-412): >= threshold."""
    return (mean_prob >= F32(threshold)).astype(np.int32)


def confidence(mean_prob_row: np.ndarray) -> float:
    """waveverify/core.py:583 — mean of the per-bit mean probabilities."""
    return float(mean_prob_row.mean(dtype=F32))


def ber(logits: np.ndarray, bits: np.ndarray, mask: Optional[np.ndarray] = None,
        threshold: float = 0.5, eps: float = 1e-8) -> float:
    """BER.forward (scripts/evaluate.py:442-516)."""
    probs = sigmoid(logits)
    B, W, T = probs.shape
    if mask is not None:
        m = np.broadcast_to(mask.astype(F32), (B, W, T))
        valid = m.sum(axis=2) > 0
        avg = (probs * m).sum(axis=2) / (m.sum(axis=2) + F32(eps))
    else:
        avg = probs.mean(axis=2)
        valid = np.ones((B, W), bool)
    dec = (avg >= threshold).astype(F32)
    err = ((dec != bits.astype(F32)) & valid).sum()
    tot = valid.sum()
    return float(err / tot) if tot > 0 else 0.0


def miou(pred: np.ndarray, gt: np.ndarray) -> float:
    """MIOU.forward (scripts/evaluate.py:591-665) on binary masks."""
    pred = np.asarray(pred)
    gt = np.asarray(gt)
    ufg = np.logical_or(pred == 1, gt == 1).sum()
    ifg = np.logical_and(pred == 1, gt == 1).sum()
    iou_fg = (1.0 if ifg == 0 else 0.0) if ufg == 0 else ifg / ufg
    ubg = np.logical_or(pred == 0, gt == 0).sum()
    ibg = np.logical_and(pred == 0, gt == 0).sum()
    iou_bg = (1.0 if ibg == 0 else 0.0) if ubg == 0 else ibg / ubg
    return float((iou_fg + iou_bg) / 2)
