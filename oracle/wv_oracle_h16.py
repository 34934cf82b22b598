"""CPU oracle of the f16-operand / f32-accumulate MODE (csrc/wv_h16.hip), torch flavour.  TEST INFRASTRUCTURE ONLY (tests/, tools/).

The mode is a different arithmetic from the reference's (activations and weights cross HBM as f16, sums are f32), so next to the checks
against the reference's own outputs (tests/golden) it gets an oracle of ITS arithmetic: the torch port of the reference path
(oracle/wv_oracle_torch.py, pinned to the reference's outputs) with a round-to-f16 inserted at every point where a kernel of the mode
rounds -- and nowhere else.  Sums here are float64, so what separates this oracle from the GPU is f32 summation order plus the
occasional f16 value that lands on the other side of a rounding boundary.  Rounding points (kernel -> what is rounded):

  conv_pre16          the stream after conv_pre (modules/seanet.py:657-664)
  rh_kernel           per ResnetBlock (seanet.py:245-281): a' = f16(log2e * ELU(c x)), W1 / log2e and W2 / log2e as f16, u' = f16(log2e *
                      ELU(DW5(W1' a') + b1)) -- the kernel keeps both activations times log2(e) and the packer divides the weights by
                      it --, the block's output y (or its activated copy) as f16
  spec16_kernel       per SpecBlock (seanet.py:463-511): the normalised log-magnitude P as f16 (the DFT itself runs on 22-bit split operands:
                      exact here), the 1x1 weight as f16, the output ELU(c x') as f16
  conv16 / conv16s    per downsample unit (seanet.py:724-760): the COMPOSED weight W[m][i][k] = pw[m][k] dw[m][i] as f16, FiLM (seanet.py:928-966)
                      in f32 behind it, the output as f16
  conv16 (conv_post)  ELU(x') as f16, the composed weight pw[m][k] dw[k][i] as f16 (seanet.py:797-823); the latent stays f32
  l2norm_c8           the normalised latent as f16 (seanet.py:288-318)
  conv16 (dec head)   decoder.model.0/.1 composed, output ELU(.) as f16 (seanet.py:1081-1094)
  conv16u             per upsample unit (seanet.py:1147-1170, conv.py:838-881): the composed weights pw[m][k] ct[k][p] and pw[m][k] ct[k][p + r]
                      as f16, the output as f16
  tail16              nothing (f32 sums over the f16 activated stream; seanet.py:1177-1202)
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from . import wv_oracle_torch as OT
from .wv_oracle import decoder_layout, dft_basis

L2E = 1.4426950408889634


def h(t: torch.Tensor) -> torch.Tensor:
    return t.half().double()


def _w(net, k):
    return net.w(k).double()


def resnet_block(net, pre, x, idx, rs, act_scale=None):
    """-> y (f16-rounded), or f16(ELU(act_scale * y)) when act_scale is given (the last block of a decoder stage)."""
    C = x.shape[1]
    a = h(L2E * F.elu(x * (1 + idx * rs ** 2) ** -0.5))
    y1 = OT.sconv1d(a, h(_w(net, f"{pre}.block.1.conv.conv.weight") / L2E), None)
    wd = _w(net, f"{pre}.block.2.conv.conv.weight")
    y1 = OT.sconv1d(y1, wd, _w(net, f"{pre}.block.2.conv.conv.bias"), groups=C)
    u = h(L2E * F.elu(y1))
    y2 = OT.sconv1d(u, h(_w(net, f"{pre}.block.4.conv.conv.weight") / L2E), None)
    wd = _w(net, f"{pre}.block.5.conv.conv.weight")
    v = OT.sconv1d(y2, wd, _w(net, f"{pre}.block.5.conv.conv.bias"), groups=C)
    p = net.opt(f"{pre}.res_scale_param")
    y = x + v * float(np.float32(rs) * (np.float32(p.reshape(-1)[0].item()) if p is not None else np.float32(1.0)))
    return h(y) if act_scale is None else h(F.elu(y * act_scale))


def spec_block_act(net, pre, x, wav, n_fft, hop, mean, std, rs, act_scale):
    basis = net.opt(f"{pre}.spec.weight")
    basis = (OT._t(dft_basis(n_fft))[:, None, :] if basis is None else basis).double()
    c = F.conv1d(F.pad(wav, (n_fft - 1, 0)), basis, None, stride=hop)
    Fq = n_fft // 2 + 1
    mag2 = (c[:, :Fq] ** 2 + c[:, Fq:] ** 2).clamp_min(1e-10)                 # the kernels' 0.5 log(max(p, 1e-10)) form (wv_dev.h stft_logmag)
    P = h((0.5 * mag2.log() - mean) / std)
    y = OT.sconv1d(P, h(_w(net, f"{pre}.layer.conv.conv.weight")), None)
    p = net.opt(f"{pre}.scale_param")
    s = float(np.float32(rs) * (np.float32(p.reshape(-1)[0].item()) if p is not None else np.float32(1.0)))
    xs = x + y * s
    return h(F.elu(xs * act_scale))


def encoder_latent(net, x, msg):
    """SEANetEncoder.forward (seanet.py:883-976) in the mode's arithmetic -> the latent BEFORE L2Norm (f32 in the kernels)."""
    cfg = net.cfg
    rs = cfg.res_scale_enc
    wav = x
    hcur = h(OT.sconv1d(x * float(np.float32(1.0 / cfg.wav_std)), _w(net, "encoder.conv_pre.1.conv.conv.weight"), _w(net, "encoder.conv_pre.1.conv.conv.bias")))
    film = None
    if msg is not None:
        e = F.linear(msg.float(), net.w("encoder.msg_embedding.0.weight"), net.w("encoder.msg_embedding.0.bias"))
        for i in range(cfg.embedding_layers):
            j = 1 + 2 * i
            e = F.relu(F.linear(e, net.w(f"encoder.msg_embedding.{j}.weight"), net.w(f"encoder.msg_embedding.{j}.bias")))
        film = e                                                    # message MLP and FiLM scalars are f32 in the mode too
    stride, mult = 1, 1
    down_scale = (1 + cfg.n_residual_enc * rs ** 2) ** -0.5
    for s, r in enumerate(cfg.ratios_enc):
        for j in range(1, cfg.n_residual_enc + 1):
            hcur = resnet_block(net, f"encoder.blocks.{s}.{j - 1}", hcur, j, rs)
        a = spec_block_act(net, f"encoder.spec_blocks.{s}", hcur, wav, mult * cfg.n_fft_base, stride, cfg.spec_means[s], cfg.spec_stds[s], rs, down_scale)
        stride *= r
        pw = _w(net, f"encoder.downsample.{s}.2.conv.conv.weight")
        wd = _w(net, f"encoder.downsample.{s}.3.conv.conv.weight")
        wc = h((pw[:, :, 0, None].float() * wd[:, 0, None, :].float()).double())          # [M][K][2r]: the packer's f32 product, rounded to f16
        y = OT.sconv1d(a, wc, _w(net, f"encoder.downsample.{s}.3.conv.conv.bias"), stride=r)
        if film is not None:
            bw = y.shape[1] // cfg.freq_bands
            bands = []
            for b in range(cfg.freq_bands):
                g = F.linear(film, net.w(f"encoder.film_layers.{s}.{b}.gamma_layer.weight"), net.w(f"encoder.film_layers.{s}.{b}.gamma_layer.bias")).double().unsqueeze(-1)
                bt = F.linear(film, net.w(f"encoder.film_layers.{s}.{b}.beta_layer.weight"), net.w(f"encoder.film_layers.{s}.{b}.beta_layer.bias")).double().unsqueeze(-1)
                bands.append(y[:, b * bw:(b + 1) * bw] * g + bt)
            y = torch.cat(bands, dim=1)
        hcur = h(y)
        mult *= 2
    a = spec_block_act(net, "encoder.spec_post", hcur, wav, mult * cfg.n_fft_base, stride, cfg.spec_means[-1], cfg.spec_stds[-1], rs, 1.0)
    wd = _w(net, "encoder.conv_post.1.conv.conv.weight")                                  # [C][1][k] depth-wise, then the 1x1
    pw = _w(net, "encoder.conv_post.2.conv.conv.weight")
    wc = h((pw[:, :, 0, None].float() * wd[None, :, 0, :].float()).double())              # [D][C][k]
    return OT.sconv1d(a, wc, _w(net, "encoder.conv_post.2.conv.conv.bias"))


def decoder_forward(net, lat):
    cfg = net.cfg
    rs = cfg.res_scale_dec
    i0, i1, ups, il = decoder_layout(cfg)
    z = h(F.normalize(lat, p=2.0, dim=1, eps=1e-12) * (lat.shape[1] ** 0.5))
    pw, wd = _w(net, f"decoder.model.{i0}.conv.conv.weight"), _w(net, f"decoder.model.{i1}.conv.conv.weight")
    wc = h((pw[:, :, 0, None].float() * wd[:, 0, None, :].float()).double())
    post = (1 + cfg.n_residual_dec * rs ** 2) ** -0.5
    a = h(F.elu(OT.sconv1d(z, wc, _w(net, f"decoder.model.{i1}.conv.conv.bias"))))                # first upsample: no Scale in front (seanet.py:1104)
    for i, (ct, pwk, res, r, C) in enumerate(ups):
        w_ct, w_pw = _w(net, f"decoder.model.{ct}.convtr.convtr.weight"), _w(net, f"decoder.model.{pwk}.conv.conv.weight")
        bias = _w(net, f"decoder.model.{pwk}.conv.conv.bias")
        Bn, K, Lf = a.shape
        M = w_pw.shape[0]
        prev = torch.cat([torch.zeros(Bn, K, 1, dtype=a.dtype), a[:, :, :-1]], dim=2)
        y = torch.zeros(Bn, M, Lf * r, dtype=torch.float64)
        for p in range(r):                                           # out[m][r l + p] = b + sum_k W[m][k] (ct[k][p] a[k][l] + ct[k][p + r] a[k][l - 1])
            w0 = h((w_pw[:, :, 0].float() * w_ct[None, :, 0, p + r].float()).double())
            w1 = h((w_pw[:, :, 0].float() * w_ct[None, :, 0, p].float()).double())
            y[:, :, p::r] = torch.einsum("mk,bkt->bmt", w0, prev) + torch.einsum("mk,bkt->bmt", w1, a)
        y = y + bias[None, :, None]
        stage_next = post                                            # the next upsample's Scale -> ELU, or the tail's
        if not res:
            a = h(F.elu(y * stage_next))
            continue
        hcur = h(y)
        for j, ri in enumerate(res):
            last = j + 1 == len(res)
            hcur = resnet_block(net, f"decoder.model.{ri}", hcur, j, rs, stage_next if last else None)
        a = hcur
    y = OT.sconv1d(a, _w(net, f"decoder.model.{il}.conv.conv.weight"), _w(net, f"decoder.model.{il}.conv.conv.bias"))
    return torch.tanh(y * cfg.wav_std)


@torch.no_grad()
def embed(net: OT.Net, x, msg):
    """wm = G(x, msg)[..., :T] + x in the f16-operand mode's arithmetic (wv_generator_forward_f16)."""
    x, msg = OT._t(x).double(), OT._t(msg).float()
    return (decoder_forward(net, encoder_latent(net, x, msg))[..., : x.shape[-1]] + x).float()
