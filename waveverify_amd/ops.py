"""Single fused units of the hot path through the C ABI (wv_op_* of include/waveverify_hip.h).

Activations are CUDA tensors; weights are host arrays in the reference's layouts.  These are
test handles — the nets in nets.py run the same kernels with weights packed once."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import _lib


def _w(a) -> Optional[np.ndarray]:
    if a is None:
        return None
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    return np.ascontiguousarray(np.asarray(a, np.float32))


def _hp(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data


def _dp(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _dev(t: torch.Tensor) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError("activations must live on the GPU")
    return t.float().contiguous()


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def pw_dw(X, w_pw, w_dw, dw_bias=None, film=None, resid=None, stride=1, dilation=1,
          pre_scale=1.0, pre_elu=True, out_scale=1.0, bands=1, act_scale: Optional[float] = None):
    """act_scale given: also returns the second output ELU(act_scale * y) -> (Y, Yact)."""
    lib = _lib.load()
    X = _dev(X)
    B, K, Tin = X.shape
    w_pw, w_dw, dw_bias = _w(w_pw), _w(w_dw), _w(dw_bias)
    M, ks = w_dw.shape[0], w_dw.shape[-1]
    w_pw = w_pw.reshape(M, K)
    Tout = -(-Tin // stride)
    Y = torch.empty((B, M, Tout), dtype=torch.float32, device=X.device)
    film = None if film is None else _dev(film)
    resid = None if resid is None else _dev(resid)
    Yact = torch.empty_like(Y) if act_scale is not None else None
    _lib.check(lib.wv_op_pw_dw(X.data_ptr(), _hp(w_pw), _hp(w_dw), _hp(dw_bias), _dp(film), _dp(resid),
                               Y.data_ptr(), B, K, M, Tin, ks, stride, dilation, pre_scale,
                               int(pre_elu), out_scale, bands, _dp(Yact), float(act_scale or 0.0),
                               _stream()), "wv_op_pw_dw")
    return Y if act_scale is None else (Y, Yact)


def resblock(X, w_pw1, w_dw1, b1, w_pw2, w_dw2, b2, pre_scale=1.0, out_scale=1.0, act_scale: Optional[float] = None,
             want_raw: bool = True):
    """Fused SEANetResnetBlock, raw in / raw out: y = X + out_scale * half2(half1(ELU(pre_scale * X))).
    -> Y, or (Y, Yact) with act_scale given, or Yact alone with want_raw=False."""
    lib = _lib.load()
    X = _dev(X)
    B, Cc, T = X.shape
    ws = [_w(w_pw1).reshape(Cc, Cc), _w(w_dw1).reshape(Cc, -1), _w(b1), _w(w_pw2).reshape(Cc, Cc),
          _w(w_dw2).reshape(Cc, -1), _w(b2)]
    Y = torch.empty_like(X) if want_raw else None
    Yact = torch.empty_like(X) if act_scale is not None else None
    _lib.check(lib.wv_op_resblock(X.data_ptr(), float(pre_scale), *[_hp(w) for w in ws], _dp(Y), _dp(Yact),
                                  B, Cc, T, out_scale, float(act_scale or 0.0), _stream()), "wv_op_resblock")
    if Y is None:
        return Yact
    return Y if act_scale is None else (Y, Yact)


def dw_pw(X, w_pw, bias=None, w_dw=None, mode=0, ks_or_ratio=0, pre_scale=1.0, pre_elu=False,
          l2norm=False, accumulate_into: Optional[torch.Tensor] = None, out_scale=1.0,
          act_scale: Optional[float] = None):
    lib = _lib.load()
    X = _dev(X)
    B, K, Tin = X.shape
    w_pw, bias, w_dw = _w(w_pw), _w(bias), _w(w_dw)
    M = w_pw.shape[0]
    w_pw = w_pw.reshape(M, K)
    Tout = Tin * ks_or_ratio if mode == 2 else Tin
    if accumulate_into is not None:
        Y = accumulate_into
        assert Y.is_cuda and Y.is_contiguous() and tuple(Y.shape) == (B, M, Tout)
    else:
        Y = torch.empty((B, M, Tout), dtype=torch.float32, device=X.device)
    Yact = torch.empty_like(Y) if act_scale is not None else None
    _lib.check(lib.wv_op_dw_pw(X.data_ptr(), _hp(w_dw), _hp(w_pw), _hp(bias), Y.data_ptr(), B, K, M,
                               Tin, mode, ks_or_ratio, pre_scale, int(pre_elu), int(l2norm),
                               int(accumulate_into is not None), out_scale, _dp(Yact),
                               float(act_scale or 0.0), _stream()), "wv_op_dw_pw")
    return Y if act_scale is None else (Y, Yact)


def stft_logmag(wav, n_fft, hop, mean=0.0, std=1.0, basis=None) -> torch.Tensor:
    lib = _lib.load()
    wav = _dev(wav)
    B, T = wav.shape[0], wav.shape[-1]
    Tf = -(-T // hop)
    P = torch.empty((B, n_fft // 2 + 1, Tf), dtype=torch.float32, device=wav.device)
    basis = _w(basis)
    _lib.check(lib.wv_op_stft_logmag(wav.data_ptr(), _hp(basis), P.data_ptr(), B, T, n_fft, hop, mean,
                                     std, _stream()), "wv_op_stft_logmag")
    return P


def spec_block(wav, w_pw, x, n_fft, hop, mean=0.0, std=1.0, out_scale=1.0, act_scale: Optional[float] = None, want_raw: bool = True,
               basis=None):
    """Whole SpecBlock in one launch: y = x + out_scale * (W @ logmag(STFT(wav))) -> Y, (Y, Yact) or Yact alone."""
    lib = _lib.load()
    wav, x = _dev(wav), _dev(x)
    B, T = wav.shape[0], wav.shape[-1]
    M = x.shape[1]
    w_pw = _w(w_pw).reshape(M, n_fft // 2 + 1)
    Y = torch.empty_like(x) if want_raw else None
    Yact = torch.empty_like(x) if act_scale is not None else None
    _lib.check(lib.wv_op_spec_block(wav.data_ptr(), _hp(_w(basis)), _hp(w_pw), x.data_ptr(), _dp(Y), _dp(Yact), B, T, n_fft, hop, M,
                                    mean, std, out_scale, float(act_scale or 0.0), _stream()), "wv_op_spec_block")
    if Y is None:
        return Yact
    return Y if act_scale is None else (Y, Yact)


def conv_pre(x, w, bias, in_scale) -> torch.Tensor:
    lib = _lib.load()
    x = _dev(x)
    B, T = x.shape[0], x.shape[-1]
    w, bias = _w(w), _w(bias)
    Cc, ks = w.shape[0], w.shape[-1]
    Y = torch.empty((B, Cc, T), dtype=torch.float32, device=x.device)
    _lib.check(lib.wv_op_conv_pre(x.data_ptr(), _hp(w), _hp(bias), Y.data_ptr(), B, Cc, T, ks, in_scale,
                                  _stream()), "wv_op_conv_pre")
    return Y


def tail(H, w, bias, x=None, T=None, pre_scale=1.0, out_scale=1.0) -> torch.Tensor:
    lib = _lib.load()
    H = _dev(H)
    B, Cc, Tin = H.shape
    T = Tin if T is None else T
    w, bias = _w(w), _w(bias)
    ks = w.shape[-1]
    x = None if x is None else _dev(x)
    out = torch.empty((B, 1, T), dtype=torch.float32, device=H.device)
    _lib.check(lib.wv_op_tail(H.data_ptr(), _hp(w), _hp(bias), _dp(x), out.data_ptr(), B, Cc, Tin, T, ks,
                              pre_scale, out_scale, _stream()), "wv_op_tail")
    return out


def head(Z, w_rev, b_rev, w_last, b_last, T, want_logits=True, want_mean=True):
    lib = _lib.load()
    Z = _dev(Z)
    B, D, Fr = Z.shape
    w_rev, b_rev, w_last, b_last = _w(w_rev), _w(b_rev), _w(w_last), _w(b_last)
    O, hop = w_rev.shape[1], w_rev.shape[2]
    nb = w_last.shape[0]
    w_last = w_last.reshape(nb, O)
    logits = torch.empty((B, nb, T), dtype=torch.float32, device=Z.device) if want_logits else None
    mean = torch.empty((B, nb), dtype=torch.float32, device=Z.device) if want_mean else None
    _lib.check(lib.wv_op_head(Z.data_ptr(), _hp(w_rev), _hp(b_rev), _hp(w_last), _hp(b_last),
                              _dp(logits), _dp(mean), B, D, O, nb, hop, Fr, T, _stream()), "wv_op_head")
    return logits, mean


# ---- the f16 mode's units (csrc/wv_h16.hip).  A "c8" tensor is torch.float16 [B, roundup(C,16)/8, T, 8] --------------------------
def _c8(t: torch.Tensor) -> torch.Tensor:
    if not (t.is_cuda and t.dtype == torch.float16 and t.dim() == 4 and t.shape[-1] == 8 and t.is_contiguous()):
        raise RuntimeError("expected a contiguous CUDA float16 tensor [B, C/8, T, 8]")
    return t


def h16_from_f32(X, scale: float = 1.0, elu: bool = False) -> torch.Tensor:
    lib = _lib.load()
    X = _dev(X)
    B, Cc, T = X.shape
    Y = torch.empty((B, (Cc + 15) // 16 * 2, T, 8), dtype=torch.float16, device=X.device)
    _lib.check(lib.wv_h16_from_f32(X.data_ptr(), Y.data_ptr(), B, Cc, T, float(scale), int(elu), _stream()), "wv_h16_from_f32")
    return Y


def h16_to_f32(X16, channels: int) -> torch.Tensor:
    lib = _lib.load()
    X16 = _c8(X16)
    B, G, T, _ = X16.shape
    if (channels + 15) // 16 * 2 != G:
        raise ValueError("channel count does not match the tensor's groups")
    Y = torch.empty((B, channels, T), dtype=torch.float32, device=X16.device)
    _lib.check(lib.wv_h16_to_f32(X16.data_ptr(), Y.data_ptr(), B, channels, T, _stream()), "wv_h16_to_f32")
    return Y


def h16_conv_pre(x, w, bias, in_scale: float = 1.0) -> torch.Tensor:
    lib = _lib.load()
    x = _dev(x)
    B, _, T = x.shape
    w, bias = _w(w), _w(bias)
    Cc, ks = w.shape[0], w.shape[-1]
    Y = torch.empty((B, Cc // 8, T, 8), dtype=torch.float16, device=x.device)
    _lib.check(lib.wv_h16_conv_pre(x.data_ptr(), _hp(w.reshape(Cc, ks)), _hp(bias), Y.data_ptr(), B, Cc, T, ks, float(in_scale), _stream()),
               "wv_h16_conv_pre")
    return Y


def h16_resblock(X16, w_pw1, w_dw1, b1, w_pw2, w_dw2, b2, pre_scale=1.0, out_scale=1.0, act_scale: Optional[float] = None,
                 want_raw: bool = True):
    lib = _lib.load()
    X16 = _c8(X16)
    B, G, T, _ = X16.shape
    Cc = 8 * G
    ws = [_w(w_pw1).reshape(Cc, Cc), _w(w_dw1).reshape(Cc, -1), _w(b1), _w(w_pw2).reshape(Cc, Cc), _w(w_dw2).reshape(Cc, -1), _w(b2)]
    Y = torch.empty_like(X16) if want_raw else None
    Yact = torch.empty_like(X16) if act_scale is not None else None
    _lib.check(lib.wv_h16_resblock(X16.data_ptr(), float(pre_scale), *[_hp(w) for w in ws], _dp(Y), _dp(Yact), B, Cc, T, float(out_scale),
                                   float(act_scale or 0.0), _stream()), "wv_h16_resblock")
    if Y is None:
        return Yact
    return Y if act_scale is None else (Y, Yact)


def h16_conv(X16, w_pw, w_dw=None, bias=None, resid16=None, K: Optional[int] = None, ks=1, stride=1, pad=0, out_scale=1.0,
             act_scale: Optional[float] = None, want_raw: bool = True, want_f32: bool = False):
    """y = out_scale * (bias + conv(x)) + resid.  -> dict with the requested outputs: "raw" (c8 f16), "act" (c8 f16), "f32" ([B,M,Tout])."""
    lib = _lib.load()
    X16 = _c8(X16)
    B, G, Tin, _ = X16.shape
    w_pw, w_dw, bias = _w(w_pw), _w(w_dw), _w(bias)
    M = w_pw.shape[0]
    K = int(K if K is not None else w_pw.reshape(M, -1).shape[1])
    if (K + 15) // 16 * 2 != G:
        raise ValueError("weight columns do not match the tensor's channel groups")
    w_pw = w_pw.reshape(M, K)
    Tout = (Tin + stride - 1) // stride
    Gm = (M + 15) // 16 * 2
    out = {}
    if want_raw:
        out["raw"] = torch.empty((B, Gm, Tout, 8), dtype=torch.float16, device=X16.device)
    if act_scale is not None:
        out["act"] = torch.empty((B, Gm, Tout, 8), dtype=torch.float16, device=X16.device)
    if want_f32:
        out["f32"] = torch.empty((B, M, Tout), dtype=torch.float32, device=X16.device)
    if resid16 is not None:
        _c8(resid16)
    _lib.check(lib.wv_h16_conv(X16.data_ptr(), _hp(w_pw), _hp(None if w_dw is None else w_dw.reshape(M, ks)), _hp(bias), _dp(resid16),
                               _dp(out.get("raw")), _dp(out.get("act")), _dp(out.get("f32")), B, K, M, Tin, ks, stride, pad, float(out_scale),
                               float(act_scale or 0.0), _stream()), "wv_h16_conv")
    return out


def h16_spec_block(wav, w_pw, x16, n_fft, hop, mean=0.0, std=1.0, out_scale=1.0, act_scale: Optional[float] = None, want_raw: bool = True, basis=None):
    """Whole SpecBlock on the f16 pipe: y = x + out_scale * (W @ logmag(STFT(wav))) -> Y16, (Y16, Yact16) or Yact16 alone (c8 f16)."""
    lib = _lib.load()
    wav, x16 = _dev(wav), _c8(x16)
    B, T = wav.shape[0], wav.shape[-1]
    M = 8 * x16.shape[1]
    w_pw = _w(w_pw).reshape(M, n_fft // 2 + 1)
    Y = torch.empty_like(x16) if want_raw else None
    Yact = torch.empty_like(x16) if act_scale is not None else None
    _lib.check(lib.wv_h16_spec_block(wav.data_ptr(), _hp(_w(basis)), _hp(w_pw), x16.data_ptr(), _dp(Y), _dp(Yact), B, T, n_fft, hop, M,
                                     mean, std, out_scale, float(act_scale or 0.0), _stream()), "wv_h16_spec_block")
    if Y is None:
        return Yact
    return Y if act_scale is None else (Y, Yact)


def h16_upsample(X16, w_ct, w_pw, bias, ratio: int, act_scale: Optional[float] = None, want_raw: bool = True):
    """The decoder's upsample unit (ELU -> depth-wise ConvTranspose1d(2r, r), trimmed -> 1x1 + bias; seanet.py:1147-1170) as one conv on
    the f16 pipe.  X16 = the PRE-ACTIVATED input, c8 f16 [B, K/8, Tin, 8] -> c8 f16 [B, M/8, Tin * r, 8]."""
    lib = _lib.load()
    X16 = _c8(X16)
    B, G, Tin, _ = X16.shape
    K = 8 * G
    w_pw, w_ct, bias = _w(w_pw), _w(w_ct), _w(bias)
    M = w_pw.shape[0]
    w_pw, w_ct = w_pw.reshape(M, K), w_ct.reshape(K, 2 * ratio)
    Y = torch.empty((B, M // 8, Tin * ratio, 8), dtype=torch.float16, device=X16.device) if want_raw else None
    Yact = torch.empty((B, M // 8, Tin * ratio, 8), dtype=torch.float16, device=X16.device) if act_scale is not None else None
    _lib.check(lib.wv_h16_upsample(X16.data_ptr(), _hp(w_ct), _hp(w_pw), _hp(bias), _dp(Y), _dp(Yact), B, K, M, Tin, int(ratio),
                                   float(act_scale or 0.0), _stream()), "wv_h16_upsample")
    if Y is None:
        return Yact
    return Y if act_scale is None else (Y, Yact)


def h16_tail(A16, w, bias, T: int, out_scale: float, x=None) -> torch.Tensor:
    """Decoder tail on the pre-activated c8 stream: tanh(out_scale * (b + Conv1d(C -> 1, ks)(a))) (+ x) -> [B, 1, T] f32."""
    lib = _lib.load()
    A16 = _c8(A16)
    B, G, Tin, _ = A16.shape
    w, bias = _w(w), _w(bias)
    Cc, ks = w.shape[-2], w.shape[-1]
    if (Cc + 15) // 16 * 2 != G:
        raise ValueError("weight channels do not match the tensor's channel groups")
    xd = _dev(x) if x is not None else None
    out = torch.empty((B, 1, T), dtype=torch.float32, device=A16.device)
    _lib.check(lib.wv_h16_tail(A16.data_ptr(), _hp(w.reshape(Cc, ks)), _hp(bias), _dp(xd), out.data_ptr(), B, Cc, Tin, T, ks, float(out_scale), _stream()),
               "wv_h16_tail")
    return out


def h16_l2norm(lat) -> torch.Tensor:
    lib = _lib.load()
    lat = _dev(lat)
    B, D, Fr = lat.shape
    Y = torch.empty((B, (D + 15) // 16 * 2, Fr, 8), dtype=torch.float16, device=lat.device)
    _lib.check(lib.wv_h16_l2norm(lat.data_ptr(), Y.data_ptr(), B, D, Fr, _stream()), "wv_h16_l2norm")
    return Y


def h16_conv_film(X16, w_pw, w_dw, bias, film, ks, stride, pad, act_scale: Optional[float] = None, want_raw: bool = True):
    """wv_h16_conv with FiLM behind the conv: film [B, bands, 2] (gamma, beta) on the device."""
    lib = _lib.load()
    X16, film = _c8(X16), _dev(film)
    B, G, Tin, _ = X16.shape
    w_pw, w_dw, bias = _w(w_pw), _w(w_dw), _w(bias)
    M = w_pw.shape[0]
    K = w_pw.reshape(M, -1).shape[1]
    Tout = (Tin + stride - 1) // stride
    Gm = (M + 15) // 16 * 2
    Y = torch.empty((B, Gm, Tout, 8), dtype=torch.float16, device=X16.device) if want_raw else None
    Yact = torch.empty((B, Gm, Tout, 8), dtype=torch.float16, device=X16.device) if act_scale is not None else None
    _lib.check(lib.wv_h16_conv_film(X16.data_ptr(), _hp(w_pw.reshape(M, K)), _hp(w_dw.reshape(M, ks) if w_dw is not None else None), _hp(bias), film.data_ptr(),
                                    int(film.shape[1]), _dp(Y), _dp(Yact), B, K, M, Tin, ks, stride, pad, float(act_scale or 0.0), _stream()), "wv_h16_conv_film")
    if Y is None:
        return Yact
    return Y if act_scale is None else (Y, Yact)
