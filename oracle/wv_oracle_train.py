"""CPU restatement (numpy float64) of the forward AND backward pass of one SEANetResnetBlock half with live
weight normalisation -- TEST INFRASTRUCTURE for the first training-step slice, pinned to
the reference's own autograd by tests/golden/grads_half_*.npz (tests/golden/make_golden_grads.py).

    a = ELU(s * x);  W = g_pw * v_pw / ||v_pw||;  h = W @ a          (1x1 conv, no bias)
    w = g_dw * v_dw / ||v_dw||;  y[m,t] = b[m] + sum_i w[m,i] * h[m, t - 4 + i]   (causal depth-wise k = 5)

Reference: modules/seanet.py:39-116 (dws_conv_block), modules/conv.py:47-88 (weight norm = torch
parametrizations.weight_norm: norm over all dims but 0), conv.py:715-763 (SConv1d causal padding).
Only tests/ import this module; the product path never does."""
from __future__ import annotations

import numpy as np


def fold(g: np.ndarray, v: np.ndarray) -> np.ndarray:
    """w = g * v / ||v||, norm over all dims but 0 (conv.py:73-74)."""
    n = np.sqrt((v.reshape(v.shape[0], -1) ** 2).sum(1)).reshape((-1,) + (1,) * (v.ndim - 1))
    return g * v / n


def fold_backward(g, v, dw):
    """(dg, dv) of w = g * v / ||v|| given dw."""
    ax = tuple(range(1, v.ndim))
    n = np.sqrt((v ** 2).sum(ax, keepdims=True))
    dot = (dw * v).sum(ax, keepdims=True)
    dg = dot / n
    dv = g / n * (dw - dot / (n * n) * v)
    return dg, dv


def half_forward(x, s, g_pw, v_pw, g_dw, v_dw, b):
    x = x.astype(np.float64)
    z = s * x
    a = np.where(z > 0, z, np.expm1(z))
    W = fold(g_pw.astype(np.float64), v_pw.astype(np.float64))[:, :, 0]          # [M, K]
    w = fold(g_dw.astype(np.float64), v_dw.astype(np.float64))[:, 0, :]          # [M, ks]
    h = np.einsum("mk,bkt->bmt", W, a)
    ks = w.shape[1]
    hp = np.pad(h, ((0, 0), (0, 0), (ks - 1, 0)))
    T = x.shape[2]
    y = b.astype(np.float64)[None, :, None] + sum(w[None, :, i, None] * hp[:, :, i:i + T] for i in range(ks))
    return y, (z, a, W, w, h)


def half_backward(x, s, g_pw, v_pw, g_dw, v_dw, b, dy):
    """-> dict(dx, dg_pw, dv_pw, dg_dw, dv_dw, db_dw) in float64."""
    y, (z, a, W, w, h) = half_forward(x, s, g_pw, v_pw, g_dw, v_dw, b)
    dy = dy.astype(np.float64)
    T = x.shape[2]
    ks = w.shape[1]
    db = dy.sum((0, 2))
    hp = np.pad(h, ((0, 0), (0, 0), (ks - 1, 0)))
    dw = np.stack([(dy * hp[:, :, i:i + T]).sum((0, 2)) for i in range(ks)], 1)            # [M, ks]
    dyp = np.pad(dy, ((0, 0), (0, 0), (0, ks - 1)))
    dh = sum(w[None, :, i, None] * dyp[:, :, ks - 1 - i:ks - 1 - i + T] for i in range(ks))  # dh[t] = sum_i w[i] dy[t + 4 - i]
    dW = np.einsum("bmt,bkt->mk", dh, a)
    da = np.einsum("mk,bmt->bkt", W, dh)
    dx = da * np.where(z > 0, 1.0, np.exp(z)) * s
    dg_pw, dv_pw = fold_backward(g_pw.astype(np.float64), v_pw.astype(np.float64), dW[:, :, None])
    dg_dw, dv_dw = fold_backward(g_dw.astype(np.float64), v_dw.astype(np.float64), dw[:, None, :])
    return dict(y=y, dx=dx, dg_pw=dg_pw, dv_pw=dv_pw, dg_dw=dg_dw, dv_dw=dv_dw, db_dw=db)


def block_forward(x, ps, res_scale_param, pre_scale, res_scale):
    """Whole SEANetResnetBlock, identity shortcut (seanet.py:245-281): y = x + s * half2(half1(pre_scale * x)),
    s = res_scale * res_scale_param (or res_scale).  ps = two dicts (g_pw, v_pw, g_dw, v_dw, b_dw)."""
    u, _ = half_forward(x, pre_scale, ps[0]["g_pw"], ps[0]["v_pw"], ps[0]["g_dw"], ps[0]["v_dw"], ps[0]["b_dw"])
    v, _ = half_forward(u, 1.0, ps[1]["g_pw"], ps[1]["v_pw"], ps[1]["g_dw"], ps[1]["v_dw"], ps[1]["b_dw"])
    s = res_scale * (1.0 if res_scale_param is None else float(np.asarray(res_scale_param).reshape(-1)[0]))
    return x.astype(np.float64) + s * v, (u, v, s)


def block_backward(x, ps, res_scale_param, pre_scale, res_scale, dy):
    """-> dict(y, dx, halves=[grads of half 1, grads of half 2], d_res_scale_param) in float64."""
    y, (u, v, s) = block_forward(x, ps, res_scale_param, pre_scale, res_scale)
    dy = dy.astype(np.float64)
    g2 = half_backward(u, 1.0, ps[1]["g_pw"], ps[1]["v_pw"], ps[1]["g_dw"], ps[1]["v_dw"], ps[1]["b_dw"], s * dy)
    g1 = half_backward(x, pre_scale, ps[0]["g_pw"], ps[0]["v_pw"], ps[0]["g_dw"], ps[0]["v_dw"], ps[0]["b_dw"], g2["dx"])
    return dict(y=y, dx=g1["dx"] + dy, halves=[g1, g2], d_res_scale_param=res_scale * float((dy * v).sum()))


def bce_logits(z, mask=None, msg=None):
    """LocalizationLoss / DecodingLoss (scripts/loss.py:947-1099): mean BCE-with-logits against
    y[b,c,t] = (msg[b,c] or 1) * (mask[b,0,t] or 1) -> (loss, dloss/dz) in float64."""
    z = z.astype(np.float64)
    y = np.ones_like(z)
    if mask is not None:
        y = y * mask.astype(np.float64)
    if msg is not None:
        y = y * msg.astype(np.float64)[:, :, None]
    loss = np.maximum(z, 0) - z * y + np.log1p(np.exp(-np.abs(z)))
    return loss.mean(), (1.0 / (1.0 + np.exp(-z)) - y) / z.size


def unit_forward(x, s, g_pw, v_pw, g_dw, v_dw, b, stride=1, elu=True):
    """The general trunk unit (SConv1d geometry, conv.py:715-763): act(s x) -> 1x1 [M,K] -> causal depth-wise conv
    (ks taps, stride, left pad ks - stride, right zero pad up to Tout = ceil(T / stride)) + bias.  Stride 1 / ks 5 is
    half_forward; M = 2K, ks = 2r, stride = r is the encoder's Downsample unit (seanet.py:733-772)."""
    x = x.astype(np.float64)
    z = s * x
    a = np.where(z > 0, z, np.expm1(z)) if elu else z
    W = fold(g_pw.astype(np.float64), v_pw.astype(np.float64))[:, :, 0]
    w = fold(g_dw.astype(np.float64), v_dw.astype(np.float64))[:, 0, :]
    h = np.einsum("mk,bkt->bmt", W, a)
    ks, T = w.shape[1], x.shape[2]
    Tout, pad = -(-T // stride), ks - stride
    hp = np.pad(h, ((0, 0), (0, 0), (pad, (Tout - 1) * stride + ks - pad - T)))
    y = b.astype(np.float64)[None, :, None] + sum(w[None, :, i, None] * hp[:, :, i:i + (Tout - 1) * stride + 1:stride] for i in range(ks))
    return y, (z, a, W, w, h, hp.shape[2], pad, Tout)


def unit_backward(x, s, g_pw, v_pw, g_dw, v_dw, b, dy, stride=1, elu=True):
    y, (z, a, W, w, h, Tp, pad, Tout) = unit_forward(x, s, g_pw, v_pw, g_dw, v_dw, b, stride, elu)
    dy = dy.astype(np.float64)
    ks, T = w.shape[1], x.shape[2]
    hp = np.pad(h, ((0, 0), (0, 0), (pad, Tp - pad - T)))
    sl = lambda i: slice(i, i + (Tout - 1) * stride + 1, stride)          # noqa: E731
    dw = np.stack([(dy * hp[:, :, sl(i)]).sum((0, 2)) for i in range(ks)], 1)
    dhp = np.zeros_like(hp)
    for i in range(ks):
        dhp[:, :, sl(i)] += w[None, :, i, None] * dy
    dh = dhp[:, :, pad:pad + T]
    dW = np.einsum("bmt,bkt->mk", dh, a)
    da = np.einsum("mk,bmt->bkt", W, dh)
    dx = da * (np.where(z > 0, 1.0, np.exp(z)) if elu else 1.0) * s
    dg_pw, dv_pw = fold_backward(g_pw.astype(np.float64), v_pw.astype(np.float64), dW[:, :, None])
    dg_dw, dv_dw = fold_backward(g_dw.astype(np.float64), v_dw.astype(np.float64), dw[:, None, :])
    return dict(y=y, dx=dx, dg_pw=dg_pw, dv_pw=dv_pw, dg_dw=dg_dw, dv_dw=dv_dw, db_dw=dy.sum((0, 2)))


def convpre_backward(x, in_scale, g, v, b, dy):
    """conv_pre (seanet.py:657-664): y = causal conv1d(in_scale * x[B,1,T], fold(g, v)[C,1,ks]) + b -> dict(y, dx, dg, dv, db)."""
    x = x.astype(np.float64)
    dy = dy.astype(np.float64)
    w = fold(g.astype(np.float64), v.astype(np.float64))[:, 0, :]          # [C, ks]
    ks, T = w.shape[1], x.shape[2]
    xp = np.pad(in_scale * x[:, 0, :], ((0, 0), (ks - 1, 0)))
    y = b.astype(np.float64)[None, :, None] + sum(w[None, :, i, None] * xp[:, None, i:i + T] for i in range(ks))
    dw = np.stack([(dy * xp[:, None, i:i + T]).sum((0, 2)) for i in range(ks)], 1)
    dyp = np.pad(dy, ((0, 0), (0, 0), (0, ks - 1)))
    dx = in_scale * sum((w[None, :, i, None] * dyp[:, :, ks - 1 - i:ks - 1 - i + T]).sum(1) for i in range(ks))
    dg, dv = fold_backward(g.astype(np.float64), v.astype(np.float64), dw[:, None, :])
    return dict(y=y, dx=dx[:, None, :], dg=dg, dv=dv, db=dy.sum((0, 2)))


def spec_add_backward(x, P, g, v, scale_param, res_scale, dy):
    """SpecBlock add (seanet.py:493-507): y = x + s * (fold(g, v)[C,F] @ P), s = res_scale * scale_param -> dict(y, dg, dv, d_scale_param)."""
    P = P.astype(np.float64)
    dy = dy.astype(np.float64)
    W = fold(g.astype(np.float64), v.astype(np.float64))[:, :, 0]
    sp = 1.0 if scale_param is None else float(np.asarray(scale_param).reshape(-1)[0])
    z = np.einsum("cf,bft->bct", W, P)
    G = np.einsum("bct,bft->cf", dy, P)
    dg, dv = fold_backward(g.astype(np.float64), v.astype(np.float64), (res_scale * sp * G)[:, :, None])
    return dict(y=x.astype(np.float64) + res_scale * sp * z, dg=dg, dv=dv, d_scale_param=res_scale * float((dy * z).sum()))


def convpost_backward(x, g_dw, v_dw, g_pw, v_pw, b, dy, l2norm=True, eps=1e-12):
    """conv_post (seanet.py:795-822): ELU -> causal depth-wise conv (no bias) -> 1x1 + bias -> L2Norm * sqrt(D)
    -> dict(y, dx, dg_dw, dv_dw, dg_pw, dv_pw, db)."""
    x = x.astype(np.float64)
    dy = dy.astype(np.float64)
    w = fold(g_dw.astype(np.float64), v_dw.astype(np.float64))[:, 0, :]
    W = fold(g_pw.astype(np.float64), v_pw.astype(np.float64))[:, :, 0]
    ks, T, D = w.shape[1], x.shape[2], W.shape[0]
    a = np.where(x > 0, x, np.expm1(x))
    ap = np.pad(a, ((0, 0), (0, 0), (ks - 1, 0)))
    h = sum(w[None, :, i, None] * ap[:, :, i:i + T] for i in range(ks))
    z = np.einsum("dc,bct->bdt", W, h) + b.astype(np.float64)[None, :, None]
    if l2norm:
        n = np.sqrt((z ** 2).sum(1, keepdims=True))
        nc = np.maximum(n, eps)
        y = z / nc * np.sqrt(D)
        dz = np.where(n > eps, np.sqrt(D) / nc * (dy - z * (dy * z).sum(1, keepdims=True) / nc ** 2), np.sqrt(D) / eps * dy)
    else:
        y, dz = z, dy
    dW = np.einsum("bdt,bct->dc", dz, h)
    dh = np.einsum("dc,bdt->bct", W, dz)
    dw = np.stack([(dh * ap[:, :, i:i + T]).sum((0, 2)) for i in range(ks)], 1)
    dhp = np.pad(dh, ((0, 0), (0, 0), (0, ks - 1)))
    da = sum(w[None, :, i, None] * dhp[:, :, ks - 1 - i:ks - 1 - i + T] for i in range(ks))
    dx = da * np.where(x > 0, 1.0, np.exp(x))
    dg_dw, dv_dw = fold_backward(g_dw.astype(np.float64), v_dw.astype(np.float64), dw[:, None, :])
    dg_pw, dv_pw = fold_backward(g_pw.astype(np.float64), v_pw.astype(np.float64), dW[:, :, None])
    return dict(y=y, dx=dx, dg_dw=dg_dw, dv_dw=dv_dw, dg_pw=dg_pw, dv_pw=dv_pw, db=dz.sum((0, 2)))


def up_backward(x, s, g_ct, v_ct, g_pw, v_pw, b, dy, elu=True):
    """The decoder's upsample unit (seanet.py:1110-1135; conv.py:838-881): act(s x) -> depth-wise ConvTranspose1d(k = 2r, stride r),
    right-trimmed to r*T -> 1x1 + bias -> dict(y, dx, dg_ct, dv_ct, dg_pw, dv_pw, db)."""
    x = x.astype(np.float64)
    dy = dy.astype(np.float64)
    wc = fold(g_ct.astype(np.float64), v_ct.astype(np.float64))[:, 0, :]         # [K, 2r]
    W = fold(g_pw.astype(np.float64), v_pw.astype(np.float64))[:, :, 0]          # [M, K]
    r, T = wc.shape[1] // 2, x.shape[2]
    z = s * x
    a = np.where(z > 0, z, np.expm1(z)) if elu else z
    full = np.zeros(a.shape[:2] + (r * T + r,))
    for j in range(2 * r):
        full[:, :, j:j + r * T:r] += a * wc[None, :, j, None]
    u = full[:, :, : r * T]
    y = np.einsum("mk,bkt->bmt", W, u) + b.astype(np.float64)[None, :, None]
    dW = np.einsum("bmt,bkt->mk", dy, u)
    du = np.pad(np.einsum("mk,bmt->bkt", W, dy), ((0, 0), (0, 0), (0, r)))
    da = sum(wc[None, :, j, None] * du[:, :, j:j + r * T:r] for j in range(2 * r))
    dwc = np.stack([(a * du[:, :, j:j + r * T:r]).sum((0, 2)) for j in range(2 * r)], 1)
    dx = da * (np.where(z > 0, 1.0, np.exp(z)) if elu else 1.0) * s
    dg_ct, dv_ct = fold_backward(g_ct.astype(np.float64), v_ct.astype(np.float64), dwc[:, None, :])
    dg_pw, dv_pw = fold_backward(g_pw.astype(np.float64), v_pw.astype(np.float64), dW[:, :, None])
    return dict(y=y, dx=dx, dg_ct=dg_ct, dv_ct=dv_ct, dg_pw=dg_pw, dv_pw=dv_pw, db=dy.sum((0, 2)))
