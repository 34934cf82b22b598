"""Gradient oracle for the detector / locator training step -- TEST INFRASTRUCTURE ONLY (tests/ import it; the product
never does).  The same torch restatement of the forward path as oracle/wv_oracle_torch.py (pinned to the reference's
outputs), made differentiable: the leaves are the reference's PARAMETRIZED state dict (weight norm g = original0,
v = original1, folded inside the graph as conv.py:47-88 does on every training forward), everything in float64, and
torch's CPU autograd supplies the gradients.  Pinned to the reference's own autograd on whole (shrunk) Detector / Locator
modules by tests/golden/netgrads_*.npz (tests/golden/make_golden_netgrads.py).

    Detector.forward / Locator.forward   /root/reference/model/detector.py:278-318,366-391, locator.py:228-299
    LocalizationLoss / DecodingLoss      /root/reference/scripts/loss.py:947-1099"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

from . import wv_oracle_torch as OTc

_W = "weight"


class LiveNet:
    """Accessor over float64 leaf tensors: w(key) folds weight norm inside the autograd graph."""

    def __init__(self, cfg, sd: Dict[str, np.ndarray]):
        self.cfg = cfg
        self.leaf = {k: torch.tensor(np.asarray(v), dtype=torch.float64, requires_grad=True) for k, v in sd.items()
                     if not k.endswith("spec.weight")}

    def w(self, k: str) -> torch.Tensor:
        if k in self.leaf:
            return self.leaf[k]
        base = k[: -len(_W)] + "parametrizations.weight."
        g, v = self.leaf[base + "original0"], self.leaf[base + "original1"]
        n = v.flatten(1).norm(dim=1).view(-1, *([1] * (v.dim() - 1)))        # norm over all dims but 0 (conv.py:73-74)
        return g * v / n

    def opt(self, k: str) -> Optional[torch.Tensor]:
        return self.leaf.get(k)


def logits_of(net: LiveNet, x: torch.Tensor) -> torch.Tensor:
    z = OTc.encoder_forward(net, x, None)
    wr = net.w("reverse_convolution.weight")
    up = F.conv_transpose1d(z, wr, net.w("reverse_convolution.bias"), stride=wr.shape[-1])[..., : x.shape[-1]]
    return F.conv1d(up, net.w("last_layer.weight"), net.w("last_layer.bias"))


def loss_and_grads(cfg, sd, x, mask, msg=None, need_dx=False):
    """-> (loss float, logits ndarray, {key: grad ndarray} for every leaf that received one, dx or None).
    msg None: LocalizationLoss(logits[B,1,T], mask); else DecodingLoss(logits[B,nb,T], mask, msg)."""
    net = LiveNet(cfg, sd)
    xt = torch.tensor(np.asarray(x), dtype=torch.float64, requires_grad=need_dx)
    logits = logits_of(net, xt)
    target = torch.tensor(np.asarray(mask), dtype=torch.float64)
    if msg is not None:
        target = torch.tensor(np.asarray(msg), dtype=torch.float64).unsqueeze(2) * target
    loss = F.binary_cross_entropy_with_logits(logits, target.expand_as(logits), reduction="mean")
    loss.backward()
    grads = {k: t.grad.numpy() for k, t in net.leaf.items() if t.grad is not None}
    return float(loss.detach()), logits.detach().numpy(), grads, (xt.grad.numpy() if need_dx else None)


def generator_loss_and_grads(cfg, sd, x, msg, target, need_dx=False):
    """wm = G(x, msg)[..., :T] + x (generator.py:360-423, watermarking.py:423-441); loss = mean((wm - target)^2) -- any
    differentiable function of wm would do: it only seeds dL/d(wm).  -> (loss, wm, {key: grad}, dx or None)."""
    net = LiveNet(cfg, sd)
    xt = torch.tensor(np.asarray(x), dtype=torch.float64, requires_grad=need_dx)
    mt = torch.tensor(np.asarray(msg), dtype=torch.float64)
    wm = OTc.decoder_forward(net, OTc.encoder_forward(net, xt, mt))[..., : xt.shape[-1]] + xt
    loss = ((wm - torch.tensor(np.asarray(target), dtype=torch.float64)) ** 2).mean()
    loss.backward()
    grads = {k: t.grad.numpy() for k, t in net.leaf.items() if t.grad is not None}
    return float(loss.detach()), wm.detach().numpy(), grads, (xt.grad.numpy() if need_dx else None)


def joint_decoding_loss_and_grads(cfgG, sdG, cfgD, sdD, x, msg, mask):
    """The watermarking objective in miniature (watermarking.py:340-441, loss.py:1020-1099): wm = G(x, msg)[..., :T] + x;
    loss = DecodingLoss(D(wm), mask, msg).  -> (loss, {G key: grad}, {D key: grad}, dL/d(wm))."""
    G, D = LiveNet(cfgG, sdG), LiveNet(cfgD, sdD)
    xt = torch.tensor(np.asarray(x), dtype=torch.float64)
    mt = torch.tensor(np.asarray(msg), dtype=torch.float64)
    wm = OTc.decoder_forward(G, OTc.encoder_forward(G, xt, mt))[..., : xt.shape[-1]] + xt
    wm.retain_grad()
    logits = logits_of(D, wm)
    target = mt.unsqueeze(2) * torch.tensor(np.asarray(mask), dtype=torch.float64)
    loss = F.binary_cross_entropy_with_logits(logits, target.expand_as(logits), reduction="mean")
    loss.backward()
    return (float(loss.detach()), {k: t.grad.numpy() for k, t in G.leaf.items() if t.grad is not None},
            {k: t.grad.numpy() for k, t in D.leaf.items() if t.grad is not None}, wm.grad.numpy())


def watermark_step_loss_and_grads(cfgG, sdG, cfgD, sdD, cfgL, sdL, x, msg, plan, seg_len, seq, lambdas):
    """The generator-update objective of the reference's step for the losses on this path (watermarking.py:340-421, train.py:1296-1344):
    wm = G(x, msg) + x; the localisation + sequence augmentation as a differentiable select (plan [B][nseg] codes 0 keep / 1 revert /
    2 zero / 3 + j clip j's original; seq = (mode, a, b, c, perm, t_out) with out[t] = in[src(t)]); D and L on the augmented audio;
    loss = l_dec DecodingLoss + l_loc LocalizationLoss + l_wav mean|wm - x|.  -> (losses dict, grads of G, D, L)."""
    G, D, L = LiveNet(cfgG, sdG), LiveNet(cfgD, sdD), LiveNet(cfgL, sdL)
    xt = torch.tensor(np.asarray(x), dtype=torch.float64)
    mt = torch.tensor(np.asarray(msg), dtype=torch.float64)
    B, _, T = xt.shape
    wm = OTc.decoder_forward(G, OTc.encoder_forward(G, xt, mt))[..., :T] + xt
    mode, a, b, c, perm, t_out = seq
    t = np.arange(t_out)
    if mode == 1:
        src = T - 1 - t
    elif mode == 2:
        src = (t - a) % T
    elif mode == 3:
        src = np.asarray(perm)[t // a] * a + t % a
    elif mode == 4:
        src = np.where((t >= a) & (t < a + c), b + (t - a), np.where((t >= b) & (t < b + c), a + (t - b), t))
    else:
        src = t
    src_t = torch.from_numpy(src.astype(np.int64))
    code = torch.from_numpy(np.asarray(plan)[:, src // seg_len].astype(np.int64))[:, None, :]           # [B,1,t_out]
    wm_s, x_s = wm[:, :, src_t], xt[:, :, src_t]
    other = torch.clamp(code - 3, min=0)[:, 0, :]                                                        # [B,t_out]
    x_other = torch.stack([xt[other[i], 0, src_t] for i in range(B)])[:, None, :]
    wm_aug = torch.where(code == 0, wm_s, torch.where(code == 1, x_s, torch.where(code == 2, torch.zeros_like(x_s), x_other)))
    mask = (code == 0).to(torch.float64)
    dec = F.binary_cross_entropy_with_logits(logits_of(D, wm_aug), (mt.unsqueeze(2) * mask), reduction="mean")
    loc = F.binary_cross_entropy_with_logits(logits_of(L, wm_aug), mask, reduction="mean")
    wav = (wm - xt).abs().mean()
    loss = lambdas["dec/loss"] * dec + lambdas["loc/loss"] * loc + lambdas["waveform/loss"] * wav
    loss.backward()
    g = lambda n: {k: t_.grad.numpy() for k, t_ in n.leaf.items() if t_.grad is not None}            # noqa: E731
    return ({"loss": float(loss.detach()), "dec/loss": float(dec.detach()), "loc/loss": float(loc.detach()), "waveform/loss": float(wav.detach())},
            g(G), g(D), g(L))
