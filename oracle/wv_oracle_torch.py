"""CPU oracle, torch flavour: the same restatement as oracle/wv_oracle.py but on
torch.nn.functional ops (ATen / oneDNN kernels, all host cores) — i.e. the arithmetic engine the
reference's own CPU path runs on.  TEST INFRASTRUCTURE ONLY: used by bench.py's `cpu_baseline` leg
(kind "port") and pinned by tests/test_oracle_golden.py against the reference's outputs.
Citations as in wv_oracle.py (paths relative to /root/reference)."""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

from .wv_oracle import decoder_layout, dft_basis, extra_padding_for_conv1d, fold_state_dict


def _t(a) -> torch.Tensor:
    return a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))


class Net:
    def __init__(self, cfg, sd: Dict[str, np.ndarray]):
        self.cfg = cfg
        self.sd = {k: _t(v).float() for k, v in fold_state_dict({k: np.asarray(v) for k, v in sd.items()}).items()}

    def w(self, k):
        return self.sd[k]

    def opt(self, k):
        return self.sd.get(k)


def sconv1d(x, w, b, stride=1, dilation=1, groups=1):
    """Causal SConv1d.forward (modules/conv.py:715-763)."""
    k = w.shape[-1]
    pad_total = (k - 1) * dilation - (stride - 1)
    extra = extra_padding_for_conv1d(x.shape[-1], k, stride, pad_total)
    return F.conv1d(F.pad(x, (pad_total, extra)), w, b, stride=stride, dilation=dilation, groups=groups)


def sconvtr1d(x, w, stride):
    """Causal depth-wise SConvTranspose1d.forward (conv.py:838-881), right trim k - s."""
    y = F.conv_transpose1d(x, w, None, stride=stride, groups=x.shape[1])
    return y[..., : y.shape[-1] - (w.shape[-1] - stride)]


def resnet_block(net, pre, x, idx, rs, dil):
    y = x * (1 + idx * rs ** 2) ** -0.5                                     # seanet.py:183
    for (pw, dw), d in zip(((1, 2), (4, 5)), dil):
        y = sconv1d(F.elu(y), net.w(f"{pre}.block.{pw}.conv.conv.weight"), None)
        wd = net.w(f"{pre}.block.{dw}.conv.conv.weight")
        y = sconv1d(y, wd, net.w(f"{pre}.block.{dw}.conv.conv.bias"), dilation=d, groups=wd.shape[0])
    p = net.opt(f"{pre}.res_scale_param")
    scale = rs * (p.reshape(-1)[0] if p is not None else 1.0)               # a tensor: differentiable for the training oracle
    return y * scale + x                                                     # seanet.py:272-277


def spec_block(net, pre, x, wav, n_fft, hop, mean, std, rs):
    basis = net.opt(f"{pre}.spec.weight")
    basis = (_t(dft_basis(n_fft))[:, None, :] if basis is None else basis).to(wav.dtype)
    c = F.conv1d(F.pad(wav, (n_fft - 1, 0)), basis, None, stride=hop)        # conv.py:1055-1068
    Fq = n_fft // 2 + 1
    y = (c[:, :Fq] ** 2 + c[:, Fq:] ** 2).clamp_min(1e-12).sqrt()
    y = (y.clamp_min(1e-5).log() - mean) / std                               # seanet.py:484,494
    y = sconv1d(y, net.w(f"{pre}.layer.conv.conv.weight"), None)
    p = net.opt(f"{pre}.scale_param")
    return x + y * (rs * (p.reshape(-1)[0] if p is not None else 1.0))


def encoder_forward(net, x, msg):
    cfg = net.cfg
    rs = cfg.res_scale_enc
    wav = x
    h = sconv1d(x * (1.0 / cfg.wav_std), net.w("encoder.conv_pre.1.conv.conv.weight"),
                net.w("encoder.conv_pre.1.conv.conv.bias"))
    film = None
    if msg is not None:
        if msg.shape[0] != x.shape[0]:
            msg = msg.repeat(int(math.ceil(x.shape[0] / msg.shape[0])), 1)[: x.shape[0]]
        e = F.linear(msg.to(x.dtype), net.w("encoder.msg_embedding.0.weight"), net.w("encoder.msg_embedding.0.bias"))
        for i in range(cfg.embedding_layers):
            j = 1 + 2 * i
            e = F.relu(F.linear(e, net.w(f"encoder.msg_embedding.{j}.weight"), net.w(f"encoder.msg_embedding.{j}.bias")))
        film = e
    stride, mult = 1, 1
    for s, r in enumerate(cfg.ratios_enc):
        for j in range(1, cfg.n_residual_enc + 1):
            h = resnet_block(net, f"encoder.blocks.{s}.{j - 1}", h, j, rs, [cfg.dilation_base ** j, 1])
        h = spec_block(net, f"encoder.spec_blocks.{s}", h, wav, mult * cfg.n_fft_base, stride,
                       cfg.spec_means[s], cfg.spec_stds[s], rs)
        stride *= r
        h = F.elu(h * (1 + cfg.n_residual_enc * rs ** 2) ** -0.5)
        h = sconv1d(h, net.w(f"encoder.downsample.{s}.2.conv.conv.weight"), None)
        wd = net.w(f"encoder.downsample.{s}.3.conv.conv.weight")
        h = sconv1d(h, wd, net.w(f"encoder.downsample.{s}.3.conv.conv.bias"), stride=r, groups=wd.shape[0])
        if film is not None:                                                 # seanet.py:928-966
            bw = h.shape[1] // cfg.freq_bands
            bands = []
            for b in range(cfg.freq_bands):
                g = F.linear(film, net.w(f"encoder.film_layers.{s}.{b}.gamma_layer.weight"),
                             net.w(f"encoder.film_layers.{s}.{b}.gamma_layer.bias")).unsqueeze(-1)
                bt = F.linear(film, net.w(f"encoder.film_layers.{s}.{b}.beta_layer.weight"),
                              net.w(f"encoder.film_layers.{s}.{b}.beta_layer.bias")).unsqueeze(-1)
                bands.append(h[:, b * bw:(b + 1) * bw] * g + bt)
            h = torch.cat(bands, dim=1)
        mult *= 2
    h = spec_block(net, "encoder.spec_post", h, wav, mult * cfg.n_fft_base, stride,
                   cfg.spec_means[-1], cfg.spec_stds[-1], rs)
    wd = net.w("encoder.conv_post.1.conv.conv.weight")
    h = sconv1d(F.elu(h), wd, None, groups=wd.shape[0])
    h = sconv1d(h, net.w("encoder.conv_post.2.conv.conv.weight"), net.w("encoder.conv_post.2.conv.conv.bias"))
    return F.normalize(h, p=2.0, dim=1, eps=1e-12) * (h.shape[1] ** 0.5)    # seanet.py:288-318


def decoder_forward(net, z):
    cfg = net.cfg
    rs = cfg.res_scale_dec
    i0, i1, ups, il = decoder_layout(cfg)
    h = sconv1d(z, net.w(f"decoder.model.{i0}.conv.conv.weight"), None)
    wd = net.w(f"decoder.model.{i1}.conv.conv.weight")
    h = sconv1d(h, wd, net.w(f"decoder.model.{i1}.conv.conv.bias"), groups=wd.shape[0])
    post = (1 + cfg.n_residual_dec * rs ** 2) ** -0.5
    for i, (ct, pw, res, r, C) in enumerate(ups):
        h = F.elu(h * post if i > 0 else h)
        h = sconvtr1d(h, net.w(f"decoder.model.{ct}.convtr.convtr.weight"), r)
        h = sconv1d(h, net.w(f"decoder.model.{pw}.conv.conv.weight"), net.w(f"decoder.model.{pw}.conv.conv.bias"))
        for j, ri in enumerate(res):
            h = resnet_block(net, f"decoder.model.{ri}", h, j, rs, [cfg.dilation_base ** j, 1])
    h = sconv1d(F.elu(h * post), net.w(f"decoder.model.{il}.conv.conv.weight"), net.w(f"decoder.model.{il}.conv.conv.bias"))
    return torch.tanh(h * cfg.wav_std)


@torch.no_grad()
def embed(net: Net, x, msg):
    """wm = G(x, msg)[..., :T] + x (generator.py:360-423, watermarking.py:423-441)."""
    x, msg = _t(x).float(), _t(msg).float()
    return decoder_forward(net, encoder_forward(net, x, msg))[..., : x.shape[-1]] + x


@torch.no_grad()
def detector_logits(net: Net, x):
    """Detector.forward / Locator.forward (detector.py:300-310, 366-391)."""
    x = _t(x).float()
    z = encoder_forward(net, x, None)
    up = F.conv_transpose1d(z, net.w("reverse_convolution.weight"), net.w("reverse_convolution.bias"),
                            stride=net.w("reverse_convolution.weight").shape[-1])[..., : x.shape[-1]]
    return F.conv1d(up, net.w("last_layer.weight"), net.w("last_layer.bias"))


def mean_probabilities(logits):
    return torch.sigmoid(logits).mean(dim=2)
