"""First slices of the training step on the GPU, behind the C ABI (wv_train_*,
include/waveverify_hip.h): a SEANetResnetBlock half and the whole block with live weight normalisation, forward and
backward, and the two BCE-with-logits losses with their gradients.

    half = TrainHalf(C); y = half.forward(x, params, pre_scale); grads = half.backward(x, params, pre_scale, dy)
    blk = TrainBlock(C); y, saved = blk.forward(x, [p1, p2], res_scale_param, pre_scale, res_scale)
    grads = blk.backward(x, [p1, p2], res_scale_param, pre_scale, res_scale, dy, saved)
    loss, dlogits = bce_logits(logits, mask, msg)

`params` = dict(g_pw [C], v_pw [C,C], g_dw [C], v_dw [C,5], b_dw [C]) of CUDA float32 tensors -- the live layout of
torch's weight_norm parametrization (original0 = g, original1 = v; /root/reference/modules/conv.py:47-88).
PyTorch is plumbing here (device memory, streams): every FLOP runs in libwaveverify_hip.so, and nothing falls
back to torch autograd."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np

import torch

from . import _lib


def _f(t: torch.Tensor) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError("training tensors must live on the GPU")
    return t.float().contiguous()


def _dst(into, key, like_shape, device):
    """Gradient destination: the caller's contiguous float32 view of the same size (a slice of a flat gradient arena -- the kernels then
    write it in place, no copy afterwards), else a fresh tensor."""
    t = None if into is None else into.get(key)
    if t is None:
        return torch.empty(like_shape, device=device)
    n = 1
    for d in like_shape:
        n *= int(d)
    if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != n or t.device != torch.device(device):
        raise ValueError(f"gradient destination {key}: need a contiguous float32 tensor of {n} elements on {device}")
    return t


class TrainHalf:
    def __init__(self, channels: int):
        self._lib = _lib.load()
        self.C = int(channels)
        self._h = C.c_void_p()
        if self._lib.wv_train_half_create(self.C, C.byref(self._h)) != 0:
            raise RuntimeError(f"wv_train_half_create: {self._lib.wv_train_last_error().decode()}")

    @staticmethod
    def _stream():
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def _p(self, p: Dict[str, torch.Tensor]):
        g_pw, v_pw = _f(p["g_pw"]).reshape(self.C), _f(p["v_pw"]).reshape(self.C, self.C)
        g_dw, v_dw = _f(p["g_dw"]).reshape(self.C), _f(p["v_dw"]).reshape(self.C, 5)
        return g_pw, v_pw, g_dw, v_dw, _f(p["b_dw"]).reshape(self.C)

    def forward(self, x: torch.Tensor, p: Dict[str, torch.Tensor], pre_scale: float) -> torch.Tensor:
        x = _f(x)
        B, Cc, T = x.shape
        g_pw, v_pw, g_dw, v_dw, b = self._p(p)
        y = torch.empty_like(x)
        rc = self._lib.wv_train_half_forward(self._h, x.data_ptr(), g_pw.data_ptr(), v_pw.data_ptr(), g_dw.data_ptr(),
                                             v_dw.data_ptr(), b.data_ptr(), float(pre_scale), y.data_ptr(), B, T,
                                             self._stream())
        if rc != 0:
            raise RuntimeError(f"wv_train_half_forward: {self._lib.wv_train_last_error().decode()}")
        return y

    def backward(self, x: torch.Tensor, p: Dict[str, torch.Tensor], pre_scale: float, dy: torch.Tensor):
        x, dy = _f(x), _f(dy)
        B, Cc, T = x.shape
        g_pw, v_pw, g_dw, v_dw, _ = self._p(p)
        out = dict(dx=torch.empty_like(x), dg_pw=torch.empty_like(g_pw), dv_pw=torch.empty_like(v_pw),
                   dg_dw=torch.empty_like(g_dw), dv_dw=torch.empty_like(v_dw), db_dw=torch.empty_like(g_dw))
        ws = torch.empty(int(self._lib.wv_train_half_workspace_bytes(self._h, B, T)), dtype=torch.uint8, device=x.device)
        rc = self._lib.wv_train_half_backward(
            self._h, x.data_ptr(), g_pw.data_ptr(), v_pw.data_ptr(), g_dw.data_ptr(), v_dw.data_ptr(), float(pre_scale),
            dy.data_ptr(), out["dx"].data_ptr(), out["dg_pw"].data_ptr(), out["dv_pw"].data_ptr(), out["dg_dw"].data_ptr(),
            out["dv_dw"].data_ptr(), out["db_dw"].data_ptr(), B, T, ws.data_ptr(), ws.numel(), self._stream())
        if rc != 0:
            raise RuntimeError(f"wv_train_half_backward: {self._lib.wv_train_last_error().decode()}")
        return out

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                self._lib.wv_train_half_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass


class TrainUnit:
    """The general trunk unit with live weight norm (wv_train_unit_*): act(pre_scale x) -> 1x1 [M,K] -> causal
    depth-wise conv (ks, stride) + bias.  ks = 2r, stride = r, M = 2K is the encoder's Downsample unit
    (/root/reference/modules/seanet.py:733-772).  params: g_pw [M], v_pw [M,K], g_dw [M], v_dw [M,ks], b_dw [M]."""

    def __init__(self, k_in: int, m_out: int, ks: int, stride: int):
        self._lib = _lib.load()
        self.K, self.M, self.ks, self.stride = int(k_in), int(m_out), int(ks), int(stride)
        self._h = C.c_void_p()
        if self._lib.wv_train_unit_create(self.K, self.M, self.ks, self.stride, C.byref(self._h)) != 0:
            raise RuntimeError(f"wv_train_unit_create: {self._lib.wv_train_last_error().decode()}")

    def _p(self, p):
        return (_f(p["g_pw"]).reshape(self.M), _f(p["v_pw"]).reshape(self.M, self.K), _f(p["g_dw"]).reshape(self.M),
                _f(p["v_dw"]).reshape(self.M, self.ks), _f(p["b_dw"]).reshape(self.M))

    def forward(self, x, p, pre_scale: float, pre_elu: bool = True):
        x = _f(x)
        B, _, T = x.shape
        g_pw, v_pw, g_dw, v_dw, b = self._p(p)
        y = torch.empty(B, self.M, -(-T // self.stride), device=x.device)
        rc = self._lib.wv_train_unit_forward(self._h, x.data_ptr(), g_pw.data_ptr(), v_pw.data_ptr(), g_dw.data_ptr(), v_dw.data_ptr(),
                                             b.data_ptr(), float(pre_scale), int(pre_elu), y.data_ptr(), B, T, TrainHalf._stream())
        if rc != 0:
            raise RuntimeError(f"wv_train_unit_forward: {self._lib.wv_train_last_error().decode()}")
        return y

    def backward(self, x, p, pre_scale: float, dy, pre_elu: bool = True, need_dx: bool = True, into=None):
        x, dy = _f(x), _f(dy)
        B, _, T = x.shape
        g_pw, v_pw, g_dw, v_dw, _ = self._p(p)
        dev = x.device
        out = dict(dx=torch.empty_like(x) if need_dx else None, dg_pw=_dst(into, "dg_pw", g_pw.shape, dev), dv_pw=_dst(into, "dv_pw", v_pw.shape, dev),
                   dg_dw=_dst(into, "dg_dw", g_dw.shape, dev), dv_dw=_dst(into, "dv_dw", v_dw.shape, dev), db_dw=_dst(into, "db_dw", g_dw.shape, dev))
        ws = torch.empty(int(self._lib.wv_train_unit_workspace_bytes(self._h, B, T)), dtype=torch.uint8, device=x.device)
        rc = self._lib.wv_train_unit_backward(
            self._h, x.data_ptr(), g_pw.data_ptr(), v_pw.data_ptr(), g_dw.data_ptr(), v_dw.data_ptr(), float(pre_scale), int(pre_elu),
            dy.data_ptr(), out["dx"].data_ptr() if need_dx else None, out["dg_pw"].data_ptr(), out["dv_pw"].data_ptr(),
            out["dg_dw"].data_ptr(), out["dv_dw"].data_ptr(), out["db_dw"].data_ptr(), B, T, ws.data_ptr(), ws.numel(), TrainHalf._stream())
        if rc != 0:
            raise RuntimeError(f"wv_train_unit_backward: {self._lib.wv_train_last_error().decode()}")
        return out

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                self._lib.wv_train_unit_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass


class _Handle:
    """Owner of one wv_train_* handle."""
    _create = _destroy = ""

    def _open(self, *args):
        self._lib = _lib.load()
        self._h = C.c_void_p()
        if getattr(self._lib, self._create)(*args, C.byref(self._h)) != 0:
            raise RuntimeError(f"{self._create}: {self._lib.wv_train_last_error().decode()}")

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what}: {self._lib.wv_train_last_error().decode()}")

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                getattr(self._lib, self._destroy)(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass


OPT_KEY = "wv_amd_optimizers"        # checkpoint key of this library's flat AdamW moments (not the reference's `optimizers` schema)


class StftFeatures(_Handle):
    """Normalised log-magnitude CausalSTFT features of one scale (/root/reference/modules/conv.py:1036-1086, seanet.py:479-494)
    with the DFT basis packed and uploaded once."""
    _create, _destroy = "wv_stft_plan_create", "wv_stft_plan_destroy"

    def __init__(self, n_fft: int, hop: int, mean: float, std: float, basis=None):
        """basis: the `...spec.weight` tensor of a checkpoint ([2F, 1, n_fft], a learned one when the reference trained with
        spec_learnable: true, conf/base.yml) or None = the reference's windowed DFT basis (conv.py:1003-1026)."""
        self.n_fft, self.hop, self.mean, self.std = int(n_fft), int(hop), float(mean), float(std)
        self._lib = _lib.load()
        self._h = C.c_void_p()
        bp = None
        if basis is not None:
            b = np.ascontiguousarray(np.asarray(basis, dtype=np.float32).reshape(-1))
            if b.size != (self.n_fft + 2) * self.n_fft:
                raise ValueError(f"spec.weight for n_fft={self.n_fft} must hold {(self.n_fft + 2) * self.n_fft} values, got {b.size}")
            bp = b.ctypes.data_as(C.c_void_p)
        if self._lib.wv_stft_plan_create(self.n_fft, bp, C.byref(self._h)) != 0:
            raise RuntimeError("wv_stft_plan_create failed")

    def __call__(self, wav: torch.Tensor) -> torch.Tensor:
        wav = _f(wav)
        B, T = wav.shape[0], wav.shape[-1]
        P = torch.empty(B, self.n_fft // 2 + 1, -(-T // self.hop), device=wav.device)
        if self._lib.wv_stft_plan_logmag(self._h, wav.data_ptr(), P.data_ptr(), B, T, self.hop, self.mean, self.std, TrainHalf._stream()) != 0:
            raise RuntimeError("wv_stft_plan_logmag failed")
        return P

    def backward(self, wav: torch.Tensor, dP: torch.Tensor, dwav: torch.Tensor, accumulate: bool = True) -> None:
        """dwav (+)= the gradient of <dP, features(wav)> towards the audio."""
        wav, dP = _f(wav), _f(dP)
        B, T = wav.shape[0], wav.shape[-1]
        ws = torch.empty(int(self._lib.wv_stft_plan_backward_workspace_bytes(self._h, B, T, self.hop)), dtype=torch.uint8, device=wav.device)
        if self._lib.wv_stft_plan_backward(self._h, wav.data_ptr(), dP.data_ptr(), dwav.data_ptr(), int(accumulate), B, T, self.hop, self.std,
                                           ws.data_ptr(), ws.numel(), TrainHalf._stream()) != 0:
            raise RuntimeError("wv_stft_plan_backward failed")


class TrainConvPre(_Handle):
    """conv_pre with live weight norm (/root/reference/modules/seanet.py:657-664): Scale(1/wav_std) -> causal SConv1d(1, C, ks).
    params: g [C], v [C,ks], b [C]."""
    _create, _destroy = "wv_train_convpre_create", "wv_train_convpre_destroy"

    def __init__(self, channels: int, ks: int):
        self.C, self.ks = int(channels), int(ks)
        self._open(self.C, self.ks)

    def forward(self, x, p, in_scale: float):
        x = _f(x)
        B, _, T = x.shape
        g, v, b = _f(p["g"]).reshape(self.C), _f(p["v"]).reshape(self.C, self.ks), _f(p["b"]).reshape(self.C)
        y = torch.empty(B, self.C, T, device=x.device)
        self._check(self._lib.wv_train_convpre_forward(self._h, x.data_ptr(), g.data_ptr(), v.data_ptr(), b.data_ptr(), float(in_scale),
                                                       y.data_ptr(), B, T, TrainHalf._stream()), "wv_train_convpre_forward")
        return y

    def backward(self, x, p, in_scale: float, dy, need_dx: bool = False, into=None):
        x, dy = _f(x), _f(dy)
        B, _, T = x.shape
        g, v = _f(p["g"]).reshape(self.C), _f(p["v"]).reshape(self.C, self.ks)
        d = x.device
        out = dict(dx=torch.empty_like(x) if need_dx else None, dg=_dst(into, "dg", g.shape, d), dv=_dst(into, "dv", v.shape, d), db=_dst(into, "db", g.shape, d))
        ws = torch.empty(int(self._lib.wv_train_convpre_workspace_bytes(self._h, B, T)), dtype=torch.uint8, device=x.device)
        self._check(self._lib.wv_train_convpre_backward(
            self._h, x.data_ptr(), g.data_ptr(), v.data_ptr(), float(in_scale), dy.data_ptr(), out["dx"].data_ptr() if need_dx else None,
            out["dg"].data_ptr(), out["dv"].data_ptr(), out["db"].data_ptr(), B, T, ws.data_ptr(), ws.numel(), TrainHalf._stream()),
            "wv_train_convpre_backward")
        return out


class TrainSpecAdd(_Handle):
    """SpecBlock add with live weight norm (/root/reference/modules/seanet.py:463-511): y = x + res_scale * scale_param *
    (W(g,v)[C,F] @ P), P = the normalised log-magnitude STFT features [B,F,T].  params: g [C], v [C,F]; scale_param [1] or None."""
    _create, _destroy = "wv_train_spec_create", "wv_train_spec_destroy"

    def __init__(self, channels: int, bins: int):
        self.C, self.F = int(channels), int(bins)
        self._open(self.C, self.F)

    def forward(self, x, P, p, scale_param, res_scale: float):
        x, P = _f(x), _f(P)
        B, _, T = x.shape
        g, v = _f(p["g"]).reshape(self.C), _f(p["v"]).reshape(self.C, self.F)
        sp = None if scale_param is None else _f(scale_param).reshape(1)
        y = torch.empty_like(x)
        self._check(self._lib.wv_train_spec_forward(self._h, x.data_ptr(), P.data_ptr(), g.data_ptr(), v.data_ptr(),
                                                    None if sp is None else sp.data_ptr(), float(res_scale), y.data_ptr(), B, T,
                                                    TrainHalf._stream()), "wv_train_spec_forward")
        return y

    def backward(self, P, p, scale_param, res_scale: float, dy, need_dP: bool = False, into=None):
        P, dy = _f(P), _f(dy)
        B, _, T = dy.shape
        g, v = _f(p["g"]).reshape(self.C), _f(p["v"]).reshape(self.C, self.F)
        sp = None if scale_param is None else _f(scale_param).reshape(1)
        out = dict(dg=_dst(into, "dg", g.shape, dy.device), dv=_dst(into, "dv", v.shape, dy.device),
                   d_scale_param=None if sp is None else _dst(into, "d_scale_param", (1,), dy.device),
                   dP=torch.empty_like(P) if need_dP else None)
        ws = torch.empty(int(self._lib.wv_train_spec_workspace_bytes(self._h, B, T)), dtype=torch.uint8, device=dy.device)
        self._check(self._lib.wv_train_spec_backward(
            self._h, P.data_ptr(), g.data_ptr(), v.data_ptr(), None if sp is None else sp.data_ptr(), float(res_scale), dy.data_ptr(),
            out["dg"].data_ptr(), out["dv"].data_ptr(), None if sp is None else out["d_scale_param"].data_ptr(),
            out["dP"].data_ptr() if need_dP else None, B, T, ws.data_ptr(), ws.numel(), TrainHalf._stream()), "wv_train_spec_backward")
        return out


class TrainConvPost(_Handle):
    """conv_post with live weight norm (/root/reference/modules/seanet.py:795-822): ELU -> causal depth-wise SConv1d(C, C, ks, no
    bias) -> SConv1d(C, D, 1, bias) -> L2Norm * sqrt(D).  params: g_dw [C], v_dw [C,ks], g_pw [D], v_pw [D,C], b [D]."""
    _create, _destroy = "wv_train_convpost_create", "wv_train_convpost_destroy"

    def __init__(self, channels: int, dimension: int, ks: int, l2norm: bool = True):
        self.C, self.D, self.ks, self.l2norm = int(channels), int(dimension), int(ks), bool(l2norm)
        self._open(self.C, self.D, self.ks)

    def _p(self, p):
        return (_f(p["g_dw"]).reshape(self.C), _f(p["v_dw"]).reshape(self.C, self.ks), _f(p["g_pw"]).reshape(self.D),
                _f(p["v_pw"]).reshape(self.D, self.C), _f(p["b"]).reshape(self.D))

    def forward(self, x, p):
        x = _f(x)
        B, _, T = x.shape
        g_dw, v_dw, g_pw, v_pw, b = self._p(p)
        y = torch.empty(B, self.D, T, device=x.device)
        self._check(self._lib.wv_train_convpost_forward(self._h, x.data_ptr(), g_dw.data_ptr(), v_dw.data_ptr(), g_pw.data_ptr(), v_pw.data_ptr(),
                                                        b.data_ptr(), int(self.l2norm), y.data_ptr(), B, T, TrainHalf._stream()),
                    "wv_train_convpost_forward")
        return y

    def backward(self, x, p, dy, into=None):
        x, dy = _f(x), _f(dy)
        B, _, T = x.shape
        g_dw, v_dw, g_pw, v_pw, b = self._p(p)
        d = x.device
        out = dict(dx=torch.empty_like(x), dg_dw=_dst(into, "dg_dw", g_dw.shape, d), dv_dw=_dst(into, "dv_dw", v_dw.shape, d),
                   dg_pw=_dst(into, "dg_pw", g_pw.shape, d), dv_pw=_dst(into, "dv_pw", v_pw.shape, d), db=_dst(into, "db", b.shape, d))
        ws = torch.empty(int(self._lib.wv_train_convpost_workspace_bytes(self._h, B, T)), dtype=torch.uint8, device=x.device)
        self._check(self._lib.wv_train_convpost_backward(
            self._h, x.data_ptr(), g_dw.data_ptr(), v_dw.data_ptr(), g_pw.data_ptr(), v_pw.data_ptr(), b.data_ptr(), int(self.l2norm), dy.data_ptr(),
            out["dx"].data_ptr(), out["dg_dw"].data_ptr(), out["dv_dw"].data_ptr(), out["dg_pw"].data_ptr(), out["dv_pw"].data_ptr(),
            out["db"].data_ptr(), B, T, ws.data_ptr(), ws.numel(), TrainHalf._stream()), "wv_train_convpost_backward")
        return out


class TrainHead(_Handle):
    """Detector / locator head (/root/reference/model/detector.py:209-218,278-318): ConvTranspose1d(D, O, k = s = hop) -> trim to T
    -> Conv1d(O, nb, 1), plain parameters w_rev [D,O,hop], b_rev [O], w_last [nb,O(,1)], b_last [nb]."""
    _create, _destroy = "wv_train_head_create", "wv_train_head_destroy"

    def __init__(self, dimension: int, output_dim: int, nbits: int, hop: int):
        self.D, self.O, self.nb, self.hop = int(dimension), int(output_dim), int(nbits), int(hop)
        self._open(self.D, self.O, self.nb, self.hop)

    def _p(self, p):
        return (_f(p["w_rev"]).reshape(self.D, self.O, self.hop), _f(p["b_rev"]).reshape(self.O), _f(p["w_last"]).reshape(self.nb, self.O),
                _f(p["b_last"]).reshape(self.nb))

    def _ws(self, B, N, dev):
        return torch.empty(int(self._lib.wv_train_head_workspace_bytes(self._h, B, N)), dtype=torch.uint8, device=dev)

    def forward(self, z, p, T: int):
        z = _f(z)
        B, _, N = z.shape
        w_rev, b_rev, w_last, b_last = self._p(p)
        logits = torch.empty(B, self.nb, T, device=z.device)
        ws = self._ws(B, N, z.device)
        self._check(self._lib.wv_train_head_forward(self._h, z.data_ptr(), w_rev.data_ptr(), b_rev.data_ptr(), w_last.data_ptr(), b_last.data_ptr(),
                                                    logits.data_ptr(), B, N, int(T), ws.data_ptr(), ws.numel(), TrainHalf._stream()),
                    "wv_train_head_forward")
        return logits

    def backward(self, z, p, dlogits, into=None):
        z, dl = _f(z), _f(dlogits)
        B, _, N = z.shape
        w_rev, b_rev, w_last, _ = self._p(p)
        d = z.device
        out = dict(dz=torch.empty_like(z), dw_rev=_dst(into, "dw_rev", w_rev.shape, d), db_rev=_dst(into, "db_rev", b_rev.shape, d),
                   dw_last=_dst(into, "dw_last", w_last.shape, d), db_last=_dst(into, "db_last", (self.nb,), d))
        ws = self._ws(B, N, z.device)
        self._check(self._lib.wv_train_head_backward(
            self._h, z.data_ptr(), w_rev.data_ptr(), b_rev.data_ptr(), w_last.data_ptr(), dl.data_ptr(), out["dz"].data_ptr(),
            out["dw_rev"].data_ptr(), out["db_rev"].data_ptr(), out["dw_last"].data_ptr(), out["db_last"].data_ptr(), B, N, dl.shape[2],
            ws.data_ptr(), ws.numel(), TrainHalf._stream()), "wv_train_head_backward")
        return out


class TrainUp(_Handle):
    """The decoder's upsample unit with live weight norm (/root/reference/modules/seanet.py:1110-1135): [Scale] -> ELU -> depth-wise
    SConvTranspose1d(K, K, 2r, stride r) -> SConv1d(K, M, 1, bias).  params: g_ct [K], v_ct [K,2r], g_pw [M], v_pw [M,K], b [M]."""
    _create, _destroy = "wv_train_up_create", "wv_train_up_destroy"

    def __init__(self, k_in: int, m_out: int, ratio: int):
        self.K, self.M, self.r = int(k_in), int(m_out), int(ratio)
        self._open(self.K, self.M, self.r)

    def _p(self, p):
        return (_f(p["g_ct"]).reshape(self.K), _f(p["v_ct"]).reshape(self.K, 2 * self.r), _f(p["g_pw"]).reshape(self.M),
                _f(p["v_pw"]).reshape(self.M, self.K), _f(p["b"]).reshape(self.M))

    def forward(self, x, p, pre_scale: float, pre_elu: bool = True):
        x = _f(x)
        B, _, T = x.shape
        g_ct, v_ct, g_pw, v_pw, b = self._p(p)
        y = torch.empty(B, self.M, T * self.r, device=x.device)
        self._check(self._lib.wv_train_up_forward(self._h, x.data_ptr(), g_ct.data_ptr(), v_ct.data_ptr(), g_pw.data_ptr(), v_pw.data_ptr(),
                                                  b.data_ptr(), float(pre_scale), int(pre_elu), y.data_ptr(), B, T, TrainHalf._stream()),
                    "wv_train_up_forward")
        return y

    def backward(self, x, p, pre_scale: float, dy, pre_elu: bool = True, into=None):
        x, dy = _f(x), _f(dy)
        B, _, T = x.shape
        g_ct, v_ct, g_pw, v_pw, b = self._p(p)
        d = x.device
        out = dict(dx=torch.empty_like(x), dg_ct=_dst(into, "dg_ct", g_ct.shape, d), dv_ct=_dst(into, "dv_ct", v_ct.shape, d),
                   dg_pw=_dst(into, "dg_pw", g_pw.shape, d), dv_pw=_dst(into, "dv_pw", v_pw.shape, d), db=_dst(into, "db", b.shape, d))
        ws = torch.empty(int(self._lib.wv_train_up_workspace_bytes(self._h, B, T)), dtype=torch.uint8, device=x.device)
        self._check(self._lib.wv_train_up_backward(
            self._h, x.data_ptr(), g_ct.data_ptr(), v_ct.data_ptr(), g_pw.data_ptr(), v_pw.data_ptr(), float(pre_scale), int(pre_elu),
            dy.data_ptr(), out["dx"].data_ptr(), out["dg_ct"].data_ptr(), out["dv_ct"].data_ptr(), out["dg_pw"].data_ptr(), out["dv_pw"].data_ptr(),
            out["db"].data_ptr(), B, T, ws.data_ptr(), ws.numel(), TrainHalf._stream()), "wv_train_up_backward")
        return out


class TrainTail(_Handle):
    """The decoder's tail with live weight norm (/root/reference/modules/seanet.py:1166-1204): Scale(post) -> ELU -> causal
    SConv1d(C, 1, ks) -> Scale(wav_std) -> Tanh, trimmed to the clip length.  params: g [1], v [1,C,ks], b [1]."""
    _create, _destroy = "wv_train_tail_create", "wv_train_tail_destroy"

    def __init__(self, channels: int, ks: int):
        self.C, self.ks = int(channels), int(ks)
        self._open(self.C, self.ks)

    def forward(self, x, p, post: float, wav_std: float, T: int):
        x = _f(x)
        B, _, Tin = x.shape
        g, v, b = _f(p["g"]).reshape(1), _f(p["v"]).reshape(self.C, self.ks), _f(p["b"]).reshape(1)
        delta = torch.empty(B, 1, T, device=x.device)
        self._check(self._lib.wv_train_tail_forward(self._h, x.data_ptr(), g.data_ptr(), v.data_ptr(), b.data_ptr(), float(post), float(wav_std),
                                                    delta.data_ptr(), B, Tin, int(T), TrainHalf._stream()), "wv_train_tail_forward")
        return delta

    def backward(self, x, p, post: float, wav_std: float, delta, d_delta, into=None):
        x, delta, dd = _f(x), _f(delta), _f(d_delta)
        B, _, Tin = x.shape
        g, v = _f(p["g"]).reshape(1), _f(p["v"]).reshape(self.C, self.ks)
        d = x.device
        out = dict(dx=torch.empty_like(x), dg=_dst(into, "dg", g.shape, d), dv=_dst(into, "dv", v.shape, d), db=_dst(into, "db", (1,), d))
        ws = torch.empty(int(self._lib.wv_train_tail_workspace_bytes(self._h, B)), dtype=torch.uint8, device=x.device)
        self._check(self._lib.wv_train_tail_backward(
            self._h, x.data_ptr(), g.data_ptr(), v.data_ptr(), float(post), float(wav_std), delta.data_ptr(), dd.data_ptr(), out["dx"].data_ptr(),
            out["dg"].data_ptr(), out["dv"].data_ptr(), out["db"].data_ptr(), B, Tin, delta.shape[-1], ws.data_ptr(), ws.numel(), TrainHalf._stream()),
            "wv_train_tail_backward")
        return out


class _HalfParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("g_pw", "v_pw", "g_dw", "v_dw", "bias")]


class _HalfGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("dg_pw", "dv_pw", "dg_dw", "dv_dw", "db")]


class TrainBlock:
    """Whole SEANetResnetBlock (/root/reference/modules/seanet.py:245-281, identity shortcut):
    y = x + res_scale * res_scale_param * half2(half1(pre_scale * x)); `res_scale_param` is the trainable [1]
    tensor of zero_init blocks or None.

    Contract: `backward(..., saved)` reuses the weight folds its `forward` left in the handle -- call it after THAT forward, with the
    same parameter tensors, before any optimizer step or other forward on this block (different tensors raise; changed values in the
    same tensors cannot be detected)."""

    def __init__(self, channels: int):
        self._lib = _lib.load()
        self.C = int(channels)
        self._h = C.c_void_p()
        if self._lib.wv_train_block_create(self.C, C.byref(self._h)) != 0:
            raise RuntimeError(f"wv_train_block_create: {self._lib.wv_train_last_error().decode()}")

    def _params(self, ps):
        keep, arr = [], (_HalfParams * 2)()
        for i, p in enumerate(ps):
            t = [_f(p["g_pw"]).reshape(self.C), _f(p["v_pw"]).reshape(self.C, self.C), _f(p["g_dw"]).reshape(self.C),
                 _f(p["v_dw"]).reshape(self.C, 5), _f(p["b_dw"]).reshape(self.C)]
            keep.append(t)
            arr[i] = _HalfParams(*[x.data_ptr() for x in t])
        return arr, keep

    def forward(self, x, ps, res_scale_param, pre_scale: float, res_scale: float):
        x = _f(x)
        B, _, T = x.shape
        arr, keep = self._params(ps)
        rsp = None if res_scale_param is None else _f(res_scale_param).reshape(1)
        y = torch.empty_like(x)
        saved = torch.empty(int(self._lib.wv_train_block_saved_bytes(self._h, B, T)), dtype=torch.uint8, device=x.device)
        rc = self._lib.wv_train_block_forward(self._h, x.data_ptr(), arr, None if rsp is None else rsp.data_ptr(), float(pre_scale),
                                              float(res_scale), y.data_ptr(), saved.data_ptr(), saved.numel(), B, T, TrainHalf._stream())
        if rc != 0:
            raise RuntimeError(f"wv_train_block_forward: {self._lib.wv_train_last_error().decode()}")
        return y, saved

    def backward(self, x, ps, res_scale_param, pre_scale: float, res_scale: float, dy, saved, into=None):
        """into: optional dict(halves=[{dg_pw, dv_pw, dg_dw, dv_dw, db_dw} x 2], d_res_scale_param) of destinations (see _dst)."""
        x, dy = _f(x), _f(dy)
        B, _, T = x.shape
        arr, keep = self._params(ps)
        rsp = None if res_scale_param is None else _f(res_scale_param).reshape(1)
        dev = x.device
        grads, garr = [], (_HalfGrads * 2)()
        for i in range(2):
            hi = None if into is None else into["halves"][i]
            g = dict(dg_pw=_dst(hi, "dg_pw", (self.C,), dev), dv_pw=_dst(hi, "dv_pw", (self.C, self.C), dev), dg_dw=_dst(hi, "dg_dw", (self.C,), dev),
                     dv_dw=_dst(hi, "dv_dw", (self.C, 5), dev), db_dw=_dst(hi, "db_dw", (self.C,), dev))
            grads.append(g)
            garr[i] = _HalfGrads(g["dg_pw"].data_ptr(), g["dv_pw"].data_ptr(), g["dg_dw"].data_ptr(), g["dv_dw"].data_ptr(), g["db_dw"].data_ptr())
        dx = torch.empty_like(x)
        drsp = None if rsp is None else _dst(into, "d_res_scale_param", (1,), dev)
        ws = torch.empty(int(self._lib.wv_train_block_workspace_bytes(self._h, B, T)), dtype=torch.uint8, device=dev)
        rc = self._lib.wv_train_block_backward(
            self._h, x.data_ptr(), arr, None if rsp is None else rsp.data_ptr(), float(pre_scale), float(res_scale), dy.data_ptr(),
            saved.data_ptr(), dx.data_ptr(), garr, None if drsp is None else drsp.data_ptr(), B, T, ws.data_ptr(), ws.numel(),
            TrainHalf._stream())
        if rc != 0:
            raise RuntimeError(f"wv_train_block_backward: {self._lib.wv_train_last_error().decode()}")
        return dict(dx=dx, halves=grads, d_res_scale_param=drsp)

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                self._lib.wv_train_block_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass


def bce_logits(logits: torch.Tensor, mask=None, msg=None, grad_scale: float = 1.0, want_grad: bool = True):
    """LocalizationLoss (msg None, logits [B,1,T], mask [B,1,T]) / DecodingLoss (logits [B,nb,T], msg [B,nb], mask [B,1,T])
    of /root/reference/scripts/loss.py:947-1099 -> (loss [1] tensor, d loss / d logits * grad_scale or None).
    Shape errors are ValueError, like the reference's."""
    if logits.dim() != 3:
        raise ValueError(f"detector_outputs must be 3D, got {logits.dim()}D")
    B, Cz, T = logits.shape
    if mask is not None and (mask.dim() != 3 or mask.shape[0] != B or mask.shape[1] != 1 or mask.shape[2] != T):
        raise ValueError(f"ground_truth_presence must be [batch, 1, samples] = [{B}, 1, {T}], got {tuple(mask.shape)}")
    if msg is not None and (msg.dim() != 2 or msg.shape[0] != B or msg.shape[1] != Cz):
        raise ValueError(f"ground_truth_message must be [batch, bits] = [{B}, {Cz}], got {tuple(msg.shape)}")
    if msg is None and Cz != 1:
        raise ValueError("localization loss: logits and presence mask must have the same shape")
    lib = _lib.load()
    z = _f(logits)
    m = None if mask is None else _f(mask)
    g = None if msg is None else _f(msg)
    loss = torch.empty(1, device=z.device)
    dz = torch.empty_like(z) if want_grad else None
    ws = torch.empty(int(lib.wv_train_bce_workspace_bytes()), dtype=torch.uint8, device=z.device)
    rc = lib.wv_train_bce_logits(z.data_ptr(), None if m is None else m.data_ptr(), None if g is None else g.data_ptr(), loss.data_ptr(),
                                 None if dz is None else dz.data_ptr(), float(grad_scale), B, Cz, T, ws.data_ptr(), ws.numel(),
                                 TrainHalf._stream())
    if rc != 0:
        raise RuntimeError(f"wv_train_bce_logits: {lib.wv_train_last_error().decode()}")
    return loss, dz


class FlatAdamW:
    """AdamW over one flat parameter arena, with the reference's schedule and clipping (scripts/train.py:1346-1358:
    clip_grad_norm_ -> AdamW step -> ExponentialLR step; conf/base.yml:128-130: betas (0.8, 0.99), lr 1e-4,
    gamma 0.999996; torch defaults eps 1e-8, weight_decay 0.01).  `step(p, g, max_norm)` updates p in place and returns
    the gradient norm before clipping (a [1] device tensor) when max_norm is given."""

    def __init__(self, numel: int, lr: float = 1e-4, betas=(0.8, 0.99), eps: float = 1e-8, weight_decay: float = 0.01,
                 gamma: float = 0.999996, device="cuda"):
        self._lib = _lib.load()
        self.lr0, self.betas, self.eps, self.weight_decay, self.gamma = float(lr), tuple(betas), float(eps), float(weight_decay), float(gamma)
        self.m = torch.zeros(numel, device=device)
        self.v = torch.zeros(numel, device=device)
        self.t = 0
        self._ws = torch.empty(int(self._lib.wv_train_bce_workspace_bytes()), dtype=torch.uint8, device=device)
        self._ss = torch.zeros(1, device=device)

    @property
    def lr(self) -> float:
        return self.lr0 * self.gamma ** self.t            # ExponentialLR: one decay per optimizer step

    def step(self, p: torch.Tensor, g: torch.Tensor, max_norm=None):
        if not (p.is_cuda and g.is_cuda and p.is_contiguous() and g.is_contiguous() and p.dtype == g.dtype == torch.float32):
            raise RuntimeError("FlatAdamW: contiguous float32 CUDA arenas required")
        if p.numel() != self.m.numel() or g.numel() != p.numel():
            raise ValueError("FlatAdamW: arena size mismatch")
        st = TrainHalf._stream()
        norm = None
        if max_norm is not None:
            if self._lib.wv_train_sumsq(g.data_ptr(), g.numel(), self._ss.data_ptr(), self._ws.data_ptr(), self._ws.numel(), st) != 0:
                raise RuntimeError(f"wv_train_sumsq: {self._lib.wv_train_last_error().decode()}")
            norm = self._ss.sqrt()
        lr = self.lr
        self.t += 1
        rc = self._lib.wv_train_adamw(p.data_ptr(), g.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), p.numel(), lr, self.betas[0],
                                      self.betas[1], self.eps, self.weight_decay, self.t,
                                      self._ss.data_ptr() if max_norm is not None else None, float(max_norm or 0.0), st)
        if rc != 0:
            raise RuntimeError(f"wv_train_adamw: {self._lib.wv_train_last_error().decode()}")
        return norm


class BlockTrainer:
    """The closed loop of the training slices on ONE SEANetResnetBlock (scripts/train.py:1421-1480 in miniature):
    live-weight-norm forward -> DecodingLoss on the block's output channels -> backward -> mean all-reduce of the flat
    gradient arena across ranks (one process per GPU, RCCL) -> clip_grad_norm_ + AdamW + ExponentialLR.
    Parameters, gradients and optimizer moments live in flat arenas; `self.params` are views into `self.arena`."""

    SHAPES = (("g_pw", lambda c: (c,)), ("v_pw", lambda c: (c, c)), ("g_dw", lambda c: (c,)), ("v_dw", lambda c: (c, 5)), ("b_dw", lambda c: (c,)))

    def __init__(self, channels: int, pre_scale: float = 1.0, res_scale: float = 0.5, seed: int = 0, lr: float = 1e-3,
                 max_norm: float = 1000.0, device="cuda"):
        C_ = int(channels)
        self.block = TrainBlock(C_)
        self.pre_scale, self.res_scale, self.max_norm = float(pre_scale), float(res_scale), float(max_norm)
        n = 2 * sum(int(torch.Size(f(C_)).numel()) for _, f in self.SHAPES) + 1
        gen = torch.Generator().manual_seed(seed)
        self.arena = torch.empty(n, device=device)
        self.grads = torch.zeros(n, device=device)
        self.params, self._gviews, off = [], [], 0
        for _ in range(2):
            pv, gv = {}, {}
            for name, f in self.SHAPES:
                shape = f(C_)
                k = int(torch.Size(shape).numel())
                init = torch.randn(shape, generator=gen)
                init = init.abs() + 0.5 if name.startswith("g_") else init * (0.1 if name == "b_dw" else shape[-1] ** -0.5)
                self.arena[off:off + k] = init.reshape(-1).to(device)
                pv[name], gv[name] = self.arena[off:off + k].view(shape), self.grads[off:off + k].view(shape)
                off += k
            self.params.append(pv)
            self._gviews.append(gv)
        self.arena[off] = 1.0
        self.res_scale_param, self._g_rsp = self.arena[off:off + 1], self.grads[off:off + 1]
        self.opt = FlatAdamW(n, lr=lr, device=device)

    def step(self, x: torch.Tensor, mask: torch.Tensor, msg: torch.Tensor):
        """One optimizer step on this rank's shard of the batch; returns (loss, gradient norm) as device tensors."""
        from .parallel import allreduce_mean_flat_
        y, saved = self.block.forward(x, self.params, self.res_scale_param, self.pre_scale, self.res_scale)
        loss, dz = bce_logits(y, mask, msg)
        g = self.block.backward(x, self.params, self.res_scale_param, self.pre_scale, self.res_scale, dz, saved)
        for i in range(2):
            for name, _ in self.SHAPES:
                self._gviews[i][name].copy_(g["halves"][i]["d" + name if name != "b_dw" else "db_dw"].view_as(self._gviews[i][name]))
        self._g_rsp.copy_(g["d_res_scale_param"])
        allreduce_mean_flat_(self.grads)
        norm = self.opt.step(self.arena, self.grads, self.max_norm)
        return loss, norm


class FilmMlp:
    """Message MLP + FiLM heads (/root/reference/modules/seanet.py:518-550,831-846): packs the reference's parameters
    (`encoder.msg_embedding.*`, `encoder.film_layers.{s}.{b}.{gamma,beta}_layer.*`) into the block the kernels read, and scatters
    the gradient block back by key."""

    def __init__(self, cfg):
        self._lib = _lib.load()
        self.Dm, self.E, self.L, self.S, self.bands = cfg.msg_dimension, cfg.embedding_dim, cfg.embedding_layers, len(cfg.strides), cfg.freq_bands
        self.NF = self.S * self.bands * 2
        self.keys = ["encoder.msg_embedding.0.weight", "encoder.msg_embedding.0.bias"]
        for i in range(self.L):
            self.keys += [f"encoder.msg_embedding.{1 + 2 * i}.weight", f"encoder.msg_embedding.{1 + 2 * i}.bias"]
        self.fw = [f"encoder.film_layers.{s}.{b}.{n}_layer.weight" for s in range(self.S) for b in range(self.bands) for n in ("gamma", "beta")]
        self.fb = [k[:-len("weight")] + "bias" for k in self.fw]
        self.keys += self.fw + self.fb
        self.np = int(self._lib.wv_train_film_param_count(self.Dm, self.E, self.L, self.S, self.bands))
        self._ws = None

    def pack(self, params) -> torch.Tensor:
        flat = torch.cat([params[k].reshape(-1) for k in self.keys])
        assert flat.numel() == self.np
        return flat

    def forward(self, msg: torch.Tensor, params, packed: Optional[torch.Tensor] = None) -> torch.Tensor:
        """packed: the parameters already laid out in `self.keys` order (a slice of a flat arena), else they are gathered here."""
        msg = _f(msg)
        B = msg.shape[0]
        self._packed, self._msg = (self.pack(params) if packed is None else packed), msg
        if self._packed.numel() != self.np or not self._packed.is_contiguous():
            raise ValueError("FilmMlp: packed parameter block of the wrong size / layout")
        film = torch.empty(B, self.NF, device=msg.device)
        self._ws = torch.empty(int(self._lib.wv_train_film_workspace_bytes(B, self.Dm, self.E, self.L, self.S, self.bands)), dtype=torch.uint8,
                               device=msg.device)
        if self._lib.wv_train_film_forward(msg.data_ptr(), self._packed.data_ptr(), film.data_ptr(), B, self.Dm, self.E, self.L, self.S, self.bands,
                                           self._ws.data_ptr(), self._ws.numel(), TrainHalf._stream()) != 0:
            raise RuntimeError(f"wv_train_film_forward: {self._lib.wv_train_last_error().decode()}")
        return film

    def apply(self, x, film, scale: int):
        x = _f(x)
        B, Cc, T = x.shape
        y = torch.empty_like(x)
        if self._lib.wv_train_film_apply(x.data_ptr(), film.data_ptr(), y.data_ptr(), B, Cc, T, self.bands, self.S, scale, TrainHalf._stream()) != 0:
            raise RuntimeError(f"wv_train_film_apply: {self._lib.wv_train_last_error().decode()}")
        return y

    def apply_backward(self, x, film, dy, dfilm, scale: int):
        x, dy = _f(x), _f(dy)
        B, Cc, T = x.shape
        dx = torch.empty_like(x)
        ws = torch.empty(B * Cc * 2, device=x.device)
        if self._lib.wv_train_film_apply_backward(x.data_ptr(), film.data_ptr(), dy.data_ptr(), dx.data_ptr(), dfilm.data_ptr(), B, Cc, T, self.bands,
                                                  self.S, scale, ws.data_ptr(), ws.numel() * 4, TrainHalf._stream()) != 0:
            raise RuntimeError(f"wv_train_film_apply_backward: {self._lib.wv_train_last_error().decode()}")
        return dx

    def backward(self, dfilm, gviews, dp: Optional[torch.Tensor] = None) -> None:
        """dp: the gradient block in `self.keys` order (a slice of a flat gradient arena, written in place), else the gradients are
        scattered into `gviews` afterwards."""
        B = self._msg.shape[0]
        scatter = dp is None
        if scatter:
            dp = torch.empty(self.np, device=dfilm.device)
        elif dp.numel() != self.np or not dp.is_contiguous() or dp.dtype != torch.float32:
            raise ValueError("FilmMlp: gradient block of the wrong size / layout")
        if self._lib.wv_train_film_backward(self._msg.data_ptr(), self._packed.data_ptr(), dfilm.data_ptr(), dp.data_ptr(), B, self.Dm, self.E, self.L,
                                            self.S, self.bands, self._ws.data_ptr(), self._ws.numel(), TrainHalf._stream()) != 0:
            raise RuntimeError(f"wv_train_film_backward: {self._lib.wv_train_last_error().decode()}")
        off = 0
        for k in (self.keys if scatter else ()):
            n = gviews[k].numel()
            gviews[k].copy_(dp[off:off + n].view_as(gviews[k]))
            off += n


class _NetTrainer:
    """Shared part of the net-level training steps: the flat parameter / gradient arenas keyed by the reference's PARAMETRIZED
    state-dict names, and SEANetEncoder.forward / backward on the training units (/root/reference/modules/seanet.py:883-976)."""

    def _build(self, cfg, state_dict, with_msg: bool, lr: float, max_norm: float, device):
        if cfg.dilation_base != 1:
            raise NotImplementedError("training units: dilation_base = 1 only")
        self.cfg, self.max_norm, self.with_msg = cfg, float(max_norm), with_msg
        skip = () if with_msg else ("encoder.msg_embedding.", "encoder.film_layers.")
        items = [(k, np.asarray(v, dtype=np.float32)) for k, v in state_dict.items()
                 if not (skip and k.startswith(skip)) and not k.endswith("spec.weight")]
        # the tensors this net never touches (a detector's / locator's message MLP + FiLM: grad None in the reference) are carried along
        # for state_dict(): the reference's own state dicts hold them
        self.frozen = {k: torch.from_numpy(np.array(v, dtype=np.float32)) for k, v in state_dict.items() if skip and k.startswith(skip)}
        # the DFT bases the state dict came with (`...spec.weight`: a buffer, or a learned parameter of a reference run with
        # spec_learnable: true, modules/conv.py:1023): the training forward uses THEM (as the inference nets do through
        # wv_model_set_stft_basis) and state_dict() writes them back unchanged; they are not trained here (no gradient towards the basis)
        self.spec_basis = {k: torch.from_numpy(np.array(v, dtype=np.float32)) for k, v in state_dict.items() if k.endswith("spec.weight")}
        # a state dict without them (fresh nets): the reference's own buffers (checkpoint.stft_basis, bit-equal to its CausalSTFT.weight:
        # tests/golden/dft_basis.npz), so that what the forward used is exactly what state_dict() writes and a resumed run repeats it
        from .checkpoint import stft_basis as _basis
        for s_ in range(len(cfg.ratios_enc) + 1):
            key = ("encoder.spec_post" if s_ == len(cfg.ratios_enc) else f"encoder.spec_blocks.{s_}") + ".spec.weight"
            self.spec_basis.setdefault(key, _basis((2 ** s_) * cfg.n_fft_base))
        # the message MLP + FiLM parameters sit together, in the order the FiLM kernels read them: their packed block and its gradient
        # are then plain slices of the arenas (no gather before the forward, no scatter after the backward)
        self.film = FilmMlp(cfg) if with_msg else None
        film_off = None
        if with_msg:
            fk, by = set(self.film.keys), dict(items)
            items = [(k, v) for k, v in items if k not in fk]
            film_off = sum(v.size for _, v in items)
            items += [(k, by[k]) for k in self.film.keys]
        n = sum(v.size for _, v in items)
        self.arena, self.grads = torch.empty(n, device=device), torch.zeros(n, device=device)
        self.params, self.gviews, self.ranges, off = {}, {}, {}, 0
        for k, v in items:
            self.arena[off:off + v.size] = torch.from_numpy(v.reshape(-1)).to(device)
            self.params[k], self.gviews[k] = self.arena[off:off + v.size].view(v.shape), self.grads[off:off + v.size].view(v.shape)
            self.ranges[k] = (off, off + v.size)
            off += v.size
        self._reducer = None
        rs, C = cfg.res_scale_enc, cfg.channels_enc
        self.conv_pre = TrainConvPre(C, cfg.kernel_size)
        self.scales, stride = [], 1
        for s, r in enumerate(cfg.ratios_enc):
            n_fft = (2 ** s) * cfg.n_fft_base
            self.scales.append(dict(C=C, r=r, blocks=[TrainBlock(C) for _ in range(cfg.n_residual_enc)], spec=TrainSpecAdd(C, n_fft // 2 + 1),
                                    down=TrainUnit(C, 2 * C, 2 * r, r),
                                    stft=StftFeatures(n_fft, stride, cfg.spec_means[s], cfg.spec_stds[s], self.spec_basis.get(f"encoder.spec_blocks.{s}.spec.weight"))))
            C *= 2
            stride *= r
        n_fft = (2 ** len(cfg.ratios_enc)) * cfg.n_fft_base
        self.spec_post = TrainSpecAdd(C, n_fft // 2 + 1)
        self.stft_post = StftFeatures(n_fft, stride, cfg.spec_means[-1], cfg.spec_stds[-1], self.spec_basis.get("encoder.spec_post.spec.weight"))
        self.conv_post = TrainConvPost(C, cfg.dimension, cfg.last_kernel_size)
        self._film_p = self._film_g = None
        if with_msg:
            if n - film_off != self.film.np:
                raise ValueError("message MLP / FiLM parameters do not match the configuration")
            self._film_p, self._film_g = self.arena[film_off:], self.grads[film_off:]
        self.down_scale = (1 + cfg.n_residual_enc * rs ** 2) ** -0.5
        self.opt = FlatAdamW(n, lr=lr, device=device)
        self._enc = None

    # ---- parameter access by the reference's keys ------------------------------------------------------------------------------
    def _wn(self, conv, inner=".conv.conv"):          # weight-normed SConv1d "<conv>.conv.conv" (or ".convtr.convtr")
        b = conv + inner + ".parametrizations.weight."
        return self.params[b + "original0"], self.params[b + "original1"]

    def _half(self, pre, pw, dw):
        g_pw, v_pw = self._wn(f"{pre}.{pw}")
        g_dw, v_dw = self._wn(f"{pre}.{dw}")
        return dict(g_pw=g_pw, v_pw=v_pw, g_dw=g_dw, v_dw=v_dw, b_dw=self.params[f"{pre}.{dw}.conv.conv.bias"])

    def _gwn(self, conv, inner=".conv.conv"):
        """(d original0, d original1) arena views of a weight-normed conv: destinations the kernels write in place."""
        b = conv + inner + ".parametrizations.weight."
        return self.gviews[b + "original0"], self.gviews[b + "original1"]

    def _into_half(self, pre, pw, dw):
        """Gradient-arena views of one 1x1 -> depth-wise unit: the kernels write them in place."""
        a, b = f"{pre}.{pw}.conv.conv.parametrizations.weight.", f"{pre}.{dw}.conv.conv.parametrizations.weight."
        return dict(dg_pw=self.gviews[a + "original0"], dv_pw=self.gviews[a + "original1"], dg_dw=self.gviews[b + "original0"],
                    dv_dw=self.gviews[b + "original1"], db_dw=self.gviews[f"{pre}.{dw}.conv.conv.bias"])

    def _spec_p(self, pre):
        g, v = self._wn(pre + ".layer")
        return dict(g=g, v=v), self.params.get(pre + ".scale_param")

    def _post_p(self):
        g_dw, v_dw = self._wn("encoder.conv_post.1")
        g_pw, v_pw = self._wn("encoder.conv_post.2")
        return dict(g_dw=g_dw, v_dw=v_dw, g_pw=g_pw, v_pw=v_pw, b=self.params["encoder.conv_post.2.conv.conv.bias"])

    # ---- gradient exchange overlapped with backward --------------------------------------------------------------------------------
    def begin_reduce(self, bucket_bytes: Optional[int] = None):
        """Arm the overlapped all-reduce for the backward pass that follows (one process per GPU; a no-op reducer with one rank):
        every bucket of the gradient arena goes on the wire as soon as backward has written the last gradient inside it."""
        from .parallel import DEFAULT_BUCKET_BYTES, OverlappedFlatReducer
        self._reducer = OverlappedFlatReducer(self.grads, self.ranges, bucket_bytes or DEFAULT_BUCKET_BYTES)
        return self._reducer

    def _done(self, *prefixes) -> None:
        """Backward has finished every gradient whose key starts with one of `prefixes`."""
        if self._reducer is not None and self._reducer.active:
            self._reducer.mark([k for k in self.ranges if k.startswith(prefixes)])

    def abort_reduce(self) -> None:
        """After an error inside a step: keep this rank's collectives matched with its peers' -- arm the reducer if the step had not
        got that far, launch every bucket not yet launched, wait -- then forget it."""
        try:
            if self._reducer is None:
                self.begin_reduce()
            self._reducer.wait()
        except Exception:
            pass
        self._reducer = None

    def finish_reduce(self) -> int:
        """Wait for the exchange (launching what backward did not mark) and take the mean.  Returns the number of collectives."""
        if self._reducer is None:
            from .parallel import allreduce_mean_flat_
            return allreduce_mean_flat_(self.grads)
        n = self._reducer.wait()
        self._reducer = None
        return n

    # ---- the trained net in the reference's checkpoint layouts ---------------------------------------------------------------------
    def _fold(self, g: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
        """g * v / ||v|| by the training units' own device fold (wv_train_fold_weight): exactly the weights the forward passes use."""
        lib = _lib.load()
        M, K = int(v.shape[0]), int(v.numel() // v.shape[0])
        w, inv = torch.empty_like(v), torch.empty(M, device=v.device)
        if lib.wv_train_fold_weight(g.data_ptr(), v.data_ptr(), w.data_ptr(), inv.data_ptr(), M, K, TrainHalf._stream()) != 0:
            raise RuntimeError(f"wv_train_fold_weight: {lib.wv_train_last_error().decode()}")
        return w

    def optimizer_state(self) -> Dict[str, object]:
        """AdamW's state by parameter key (CPU tensors): {step, exp_avg{key}, exp_avg_sq{key}} -- independent of the arena's order."""
        return {"step": int(self.opt.t),
                "exp_avg": {k: self.opt.m[lo:hi].detach().cpu().clone() for k, (lo, hi) in self.ranges.items()},
                "exp_avg_sq": {k: self.opt.v[lo:hi].detach().cpu().clone() for k, (lo, hi) in self.ranges.items()}}

    def load_optimizer_state(self, st) -> bool:
        """Restore `optimizer_state()` when it names exactly this trainer's parameters (same keys and sizes); else leave the fresh
        optimizer as it is and return False (e.g. a checkpoint written in the stripped layout: its moments belong to other tensors)."""
        try:
            ok = set(st["exp_avg"]) == set(self.ranges) and all(st["exp_avg"][k].numel() == hi - lo == st["exp_avg_sq"][k].numel()
                                                                 for k, (lo, hi) in self.ranges.items())
        except (KeyError, TypeError, AttributeError):
            ok = False
        if not ok:
            return False
        for k, (lo, hi) in self.ranges.items():
            self.opt.m[lo:hi] = st["exp_avg"][k].reshape(-1).to(self.opt.m.device)
            self.opt.v[lo:hi] = st["exp_avg_sq"][k].reshape(-1).to(self.opt.v.device)
        self.opt.t = int(st["step"])
        return True

    def state_dict(self, parametrized: bool = False) -> Dict[str, torch.Tensor]:
        """The net as the reference's `state_dict()` (CPU tensors, every key of /root/reference/modules/seanet.py's modules incl. the
        DFT buffers and, for a detector / locator, the unused message MLP + FiLM tensors it was built with).
        parametrized=False: the STRIPPED layout the reference writes (scripts/train.py:1624-1650 removes the weight-norm
        parametrizations before saving; `...conv.conv.weight` = g v / ||v||, folded on the device like every training forward);
        parametrized=True: the live layout (`...parametrizations.weight.original0/1`)."""
        from .params import param_specs
        from .checkpoint import stft_basis
        out: Dict[str, torch.Tensor] = {}
        for key, shape, role in param_specs(self.cfg):
            if role == "wn":
                base = key[: -len("weight")] + "parametrizations.weight.original"
                g, v = self.params[base + "0"], self.params[base + "1"]
                if parametrized:
                    out[base + "0"], out[base + "1"] = g.detach().cpu().clone(), v.detach().cpu().clone()
                else:
                    out[key] = self._fold(g, v).cpu()
            elif key in self.params:
                out[key] = self.params[key].detach().cpu().clone()
            elif key in self.frozen:
                out[key] = self.frozen[key].clone()
            else:
                raise KeyError(f"{key}: not in the state dict this trainer was built from")
        for s in range(len(self.cfg.ratios_enc) + 1):
            pre = "encoder.spec_post" if s == len(self.cfg.ratios_enc) else f"encoder.spec_blocks.{s}"
            n_fft = (2 ** s) * self.cfg.n_fft_base
            kept = self.spec_basis.get(pre + ".spec.weight")
            out[pre + ".spec.weight"] = kept.reshape(n_fft + 2, 1, n_fft).clone() if kept is not None else stft_basis(n_fft)
        return out

    def _block_fwd(self, blk, pre, h, pre_scale, rs):
        ps = [self._half(pre + ".block", 1, 2), self._half(pre + ".block", 4, 5)]
        y, saved = blk.forward(h, ps, self.params.get(pre + ".res_scale_param"), pre_scale, rs)
        return y, (h, saved, pre_scale)

    def _block_bwd(self, blk, pre, rec, dh, rs):
        h_in, saved, pre_scale = rec
        ps = [self._half(pre + ".block", 1, 2), self._half(pre + ".block", 4, 5)]
        rsp = self.params.get(pre + ".res_scale_param")
        into = dict(halves=[self._into_half(pre + ".block", 1, 2), self._into_half(pre + ".block", 4, 5)])
        if rsp is not None:
            into["d_res_scale_param"] = self.gviews[pre + ".res_scale_param"]
        return blk.backward(h_in, ps, rsp, pre_scale, rs, dh, saved, into)["dx"]

    # ---- the encoder -----------------------------------------------------------------------------------------------------------------
    def encoder_forward(self, x: torch.Tensor, msg: Optional[torch.Tensor]) -> torch.Tensor:
        cfg, rs = self.cfg, self.cfg.res_scale_enc
        sv = dict(x=x, scales=[], film=None)
        if self.with_msg:
            msg = _f(msg)
            if msg.shape[0] != x.shape[0]:                                         # one message for the batch (watermarking.py:320-329)
                msg = msg.repeat(-(-x.shape[0] // msg.shape[0]), 1)[: x.shape[0]].contiguous()
            sv["film"] = self.film.forward(msg, self.params, self._film_p)
        g, v = self._wn("encoder.conv_pre.1")
        h = self.conv_pre.forward(x, dict(g=g, v=v, b=self.params["encoder.conv_pre.1.conv.conv.bias"]), 1.0 / cfg.wav_std)
        for s, sc in enumerate(self.scales):
            rec = dict(blocks=[])
            for j, blk in enumerate(sc["blocks"]):
                h, r_ = self._block_fwd(blk, f"encoder.blocks.{s}.{j}", h, (1 + (j + 1) * rs ** 2) ** -0.5, rs)   # idx = j + 1 (seanet.py:183,684)
                rec["blocks"].append(r_)
            P = sc["stft"](x)
            sp, scp = self._spec_p(f"encoder.spec_blocks.{s}")
            h = sc["spec"].forward(h, P, sp, scp, rs)
            rec["P"], rec["down_in"] = P, h
            h = sc["down"].forward(h, self._half(f"encoder.downsample.{s}", 2, 3), self.down_scale, True)
            if self.with_msg:
                rec["film_in"] = h
                h = self.film.apply(h, sv["film"], s)
            sv["scales"].append(rec)
        P = self.stft_post(x)
        sp, scp = self._spec_p("encoder.spec_post")
        h = self.spec_post.forward(h, P, sp, scp, rs)
        sv["P_post"], sv["post_in"] = P, h
        self._enc = sv
        return self.conv_post.forward(h, self._post_p())

    def encoder_backward(self, dz: torch.Tensor, need_dx: bool = False):
        """Fills the encoder's gradients; with need_dx returns dL/d(audio): through conv_pre AND through every scale's spectrogram
        branch (SpecBlock 1x1 -> normalise -> log -> |STFT|)."""
        cfg, rs, sv = self.cfg, self.cfg.res_scale_enc, self._enc
        if sv is None:
            raise RuntimeError("backward before forward")
        (gd0, gd1), (gp0, gp1) = self._gwn("encoder.conv_post.1"), self._gwn("encoder.conv_post.2")
        g = self.conv_post.backward(sv["post_in"], self._post_p(), dz,
                                    dict(dg_dw=gd0, dv_dw=gd1, dg_pw=gp0, dv_pw=gp1, db=self.gviews["encoder.conv_post.2.conv.conv.bias"]))
        dh = g["dx"]
        self._done("encoder.conv_post.")

        dx_spec = torch.zeros_like(sv["x"]) if need_dx else None

        def spec_back(unit, stft, pre, P, dy):
            sp, scp = self._spec_p(pre)
            b = pre + ".layer.conv.conv.parametrizations.weight."
            into = dict(dg=self.gviews[b + "original0"], dv=self.gviews[b + "original1"])
            if scp is not None:
                into["d_scale_param"] = self.gviews[pre + ".scale_param"]
            gs = unit.backward(P, sp, scp, rs, dy, need_dx, into)
            if need_dx:
                stft.backward(sv["x"], gs["dP"], dx_spec, True)
        spec_back(self.spec_post, self.stft_post, "encoder.spec_post", sv["P_post"], dh)
        self._done("encoder.spec_post.")
        dfilm = torch.zeros_like(sv["film"]) if self.with_msg else None
        for s in reversed(range(len(self.scales))):
            sc, rec = self.scales[s], sv["scales"][s]
            if self.with_msg:
                dh = self.film.apply_backward(rec["film_in"], sv["film"], dh, dfilm, s)
            gd = sc["down"].backward(rec["down_in"], self._half(f"encoder.downsample.{s}", 2, 3), self.down_scale, dh, True, True,
                                     self._into_half(f"encoder.downsample.{s}", 2, 3))
            dh = gd["dx"]
            spec_back(sc["spec"], sc["stft"], f"encoder.spec_blocks.{s}", rec["P"], dh)          # the add passes dh through unchanged
            for j in reversed(range(len(sc["blocks"]))):
                dh = self._block_bwd(sc["blocks"][j], f"encoder.blocks.{s}.{j}", rec["blocks"][j], dh, rs)
            self._done(f"encoder.downsample.{s}.", f"encoder.spec_blocks.{s}.", f"encoder.blocks.{s}.")
        if self.with_msg:
            self.film.backward(dfilm, self.gviews, self._film_g)
            self._done("encoder.msg_embedding.", "encoder.film_layers.")
        gv = self._wn("encoder.conv_pre.1")
        g0, g1 = self._gwn("encoder.conv_pre.1")
        gp = self.conv_pre.backward(sv["x"], dict(g=gv[0], v=gv[1], b=self.params["encoder.conv_pre.1.conv.conv.bias"]), 1.0 / cfg.wav_std,
                                    dh, need_dx, dict(dg=g0, dv=g1, db=self.gviews["encoder.conv_pre.1.conv.conv.bias"]))
        self._enc = None
        self._done("encoder.conv_pre.")
        return gp["dx"] + dx_spec if need_dx else None

    def _optimizer_step(self):
        self.finish_reduce()
        return self.opt.step(self.arena, self.grads, self.max_norm)


class EncoderNetTrainer(_NetTrainer):
    """The training step of the Detector or the Locator on the HIP units: SEANetEncoder (msg = None) + head under the reference's
    BCE loss (/root/reference/model/detector.py:278-318, locator.py:228-299, modules/seanet.py:883-976, scripts/loss.py:947-1099,
    scripts/train.py:1346-1358).  `state_dict` is the reference's PARAMETRIZED layout (weight norm original0 / original1); every
    tensor that receives a gradient in the reference lives in one flat arena (`self.arena`, views in `self.params`), its gradient
    in `self.grads`; the message MLP and FiLM layers of the encoder are unused without a message and stay out, as they stay
    untouched in the reference (grad None).  One process per GPU: `step` all-reduces the gradient arena over RCCL.
    Clip lengths must keep every ResnetBlock stage a multiple of 4 samples (T = 16000 does for both nets)."""

    def __init__(self, cfg, state_dict, lr: float = 1e-4, max_norm: float = 1000.0, device="cuda"):
        if cfg.kind not in ("detector", "locator"):
            raise ValueError("EncoderNetTrainer: detector or locator")
        self._build(cfg, state_dict, False, lr, max_norm, device)
        self.nb = cfg.nbits if cfg.kind == "detector" else 1
        self.head = TrainHead(cfg.dimension, cfg.output_dim, self.nb, cfg.hop_length)
        self._z = None

    def _head_p(self):
        return dict(w_rev=self.params["reverse_convolution.weight"], b_rev=self.params["reverse_convolution.bias"],
                    w_last=self.params["last_layer.weight"], b_last=self.params["last_layer.bias"])

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = _f(x)
        self._z = self.encoder_forward(x, None)
        return self.head.forward(self._z, self._head_p(), x.shape[-1])

    def backward(self, dlogits: torch.Tensor, need_dx: bool = False):
        """Fills `self.grads` (every view of `self.gviews`); returns dL/dx through conv_pre when asked."""
        if self._z is None:
            raise RuntimeError("backward before forward")
        g = self.head.backward(self._z, self._head_p(), dlogits,
                               dict(dw_rev=self.gviews["reverse_convolution.weight"], db_rev=self.gviews["reverse_convolution.bias"],
                                    dw_last=self.gviews["last_layer.weight"], db_last=self.gviews["last_layer.bias"]))
        self._z = None
        self._done("reverse_convolution.", "last_layer.")
        return self.encoder_backward(g["dz"], need_dx)

    def step(self, x: torch.Tensor, mask: torch.Tensor, msg: Optional[torch.Tensor] = None):
        """One optimizer step on this rank's clips: LocalizationLoss (locator: msg None) or DecodingLoss (detector) ->
        backward -> mean all-reduce of the gradient arena -> clip + AdamW + ExponentialLR.  Returns (loss, gradient norm)."""
        logits = self.forward(x)
        loss, dz = bce_logits(logits, mask, msg)
        self.begin_reduce()
        self.backward(dz)
        return loss, self._optimizer_step()


class GeneratorTrainer(_NetTrainer):
    """Forward and backward of the Generator on the HIP units (/root/reference/model/generator.py:290-423, modules/seanet.py:883-976
    with the message path, :1067-1226 decoder; model/watermarking.py:423-441: wm = G(x, msg)[..., :T] + x).  `forward(x, msg)` returns
    the watermarked audio; `backward(d_wm)` takes the gradient of ANY loss on it (the reference's waveform / spectral / adversarial
    losses stay on PyTorch-ROCm) and fills `self.grads`; `apply_gradients()` all-reduces and steps AdamW."""

    def __init__(self, cfg, state_dict, lr: float = 1e-4, max_norm: float = 1000.0, device="cuda"):
        from .params import decoder_layout
        if cfg.kind != "generator":
            raise ValueError("GeneratorTrainer: generator")
        self._build(cfg, state_dict, True, lr, max_norm, device)
        self.i_pw0, self.i_dw0, self.ups, self.i_last = decoder_layout(cfg)
        Ctop = (2 ** len(cfg.strides)) * cfg.channels_dec
        self.dec_in = TrainUnit(cfg.dimension, Ctop, cfg.kernel_size, 1)
        self.dec_ups = [dict(up=TrainUp(C, C // 2, r), blocks=[TrainBlock(C // 2) for _ in res]) for _, _, res, r, C in self.ups]
        self.tail = TrainTail(cfg.channels_dec, cfg.last_kernel_size)
        self.post = (1 + cfg.n_residual_dec * cfg.res_scale_dec ** 2) ** -0.5
        self._dec = None

    def _in_p(self):
        g_pw, v_pw = self._wn(f"decoder.model.{self.i_pw0}")
        g_dw, v_dw = self._wn(f"decoder.model.{self.i_dw0}")
        return dict(g_pw=g_pw, v_pw=v_pw, g_dw=g_dw, v_dw=v_dw, b_dw=self.params[f"decoder.model.{self.i_dw0}.conv.conv.bias"])

    def _up_p(self, ct, pw):
        g_ct, v_ct = self._wn(f"decoder.model.{ct}", ".convtr.convtr")
        g_pw, v_pw = self._wn(f"decoder.model.{pw}")
        return dict(g_ct=g_ct, v_ct=v_ct, g_pw=g_pw, v_pw=v_pw, b=self.params[f"decoder.model.{pw}.conv.conv.bias"])

    def _tail_p(self):
        g, v = self._wn(f"decoder.model.{self.i_last}")
        return dict(g=g, v=v, b=self.params[f"decoder.model.{self.i_last}.conv.conv.bias"])

    def forward(self, x: torch.Tensor, msg: torch.Tensor) -> torch.Tensor:
        cfg, rs = self.cfg, self.cfg.res_scale_dec
        x = _f(x)
        z = self.encoder_forward(x, msg)
        sv = dict(z=z, ups=[])
        h = self.dec_in.forward(z, self._in_p(), 1.0, False)
        for i, ((ct, pw, res, r, C), du) in enumerate(zip(self.ups, self.dec_ups)):
            rec = dict(up_in=h, blocks=[])
            h = du["up"].forward(h, self._up_p(ct, pw), self.post if i > 0 else 1.0, True)
            for j, (ri, blk) in enumerate(zip(res, du["blocks"])):
                h, r_ = self._block_fwd(blk, f"decoder.model.{ri}", h, (1 + j * rs ** 2) ** -0.5, rs)               # idx = j (seanet.py:1143-1161)
                rec["blocks"].append(r_)
            sv["ups"].append(rec)
        sv["tail_in"] = h
        delta = self.tail.forward(h, self._tail_p(), self.post, cfg.wav_std, x.shape[-1])
        sv["delta"] = delta
        self._dec = sv
        return delta + x

    def backward(self, d_wm: torch.Tensor, need_dx: bool = False):
        """d_wm = dLoss/d(watermarked audio) [B,1,T].  Returns the gradient towards the input audio through the generator's conv_pre
        path plus the identity (wm = delta + x) when asked."""
        cfg, rs, sv = self.cfg, self.cfg.res_scale_dec, self._dec
        if sv is None:
            raise RuntimeError("backward before forward")
        d_wm = _f(d_wm)
        g0, g1 = self._gwn(f"decoder.model.{self.i_last}")
        g = self.tail.backward(sv["tail_in"], self._tail_p(), self.post, cfg.wav_std, sv["delta"], d_wm,
                               dict(dg=g0, dv=g1, db=self.gviews[f"decoder.model.{self.i_last}.conv.conv.bias"]))
        dh = g["dx"]
        self._done(f"decoder.model.{self.i_last}.")
        for i in reversed(range(len(self.ups))):
            (ct, pw, res, r, C), du, rec = self.ups[i], self.dec_ups[i], sv["ups"][i]
            for j in reversed(range(len(res))):
                dh = self._block_bwd(du["blocks"][j], f"decoder.model.{res[j]}", rec["blocks"][j], dh, rs)
                self._done(f"decoder.model.{res[j]}.")
            (c0, c1), (p0, p1) = self._gwn(f"decoder.model.{ct}", ".convtr.convtr"), self._gwn(f"decoder.model.{pw}")
            gu = du["up"].backward(rec["up_in"], self._up_p(ct, pw), self.post if i > 0 else 1.0, dh, True,
                                   dict(dg_ct=c0, dv_ct=c1, dg_pw=p0, dv_pw=p1, db=self.gviews[f"decoder.model.{pw}.conv.conv.bias"]))
            dh = gu["dx"]
            self._done(f"decoder.model.{ct}.", f"decoder.model.{pw}.")
        (p0, p1), (d0, d1) = self._gwn(f"decoder.model.{self.i_pw0}"), self._gwn(f"decoder.model.{self.i_dw0}")
        gi = self.dec_in.backward(sv["z"], self._in_p(), 1.0, dh, False, True,
                                  dict(dg_pw=p0, dv_pw=p1, dg_dw=d0, dv_dw=d1, db_dw=self.gviews[f"decoder.model.{self.i_dw0}.conv.conv.bias"]))
        self._dec = None
        self._done(f"decoder.model.{self.i_pw0}.", f"decoder.model.{self.i_dw0}.")
        dx = self.encoder_backward(gi["dx"], need_dx)
        return None if dx is None else dx + d_wm

    def apply_gradients(self):
        """Mean all-reduce of the gradient arena over the ranks, clip_grad_norm_, AdamW, ExponentialLR (scripts/train.py:1346-1358)."""
        return self._optimizer_step()


def l1_loss(a: torch.Tensor, b: torch.Tensor, grad_scale: float = 1.0, want_grad: bool = True):
    """mean |a - b| (the reference's waveform loss, scripts/train.py:1322) -> (loss [1], grad_scale * dloss/da or None)."""
    a, b = _f(a), _f(b)
    if a.shape != b.shape:
        raise ValueError(f"shape mismatch: {tuple(a.shape)} vs {tuple(b.shape)}")
    lib = _lib.load()
    loss = torch.empty(1, device=a.device)
    da = torch.empty_like(a) if want_grad else None
    ws = torch.empty(int(lib.wv_train_bce_workspace_bytes()), dtype=torch.uint8, device=a.device)
    if lib.wv_train_l1(a.data_ptr(), b.data_ptr(), loss.data_ptr(), None if da is None else da.data_ptr(), float(grad_scale), a.numel(),
                       ws.data_ptr(), ws.numel(), TrainHalf._stream()) != 0:
        raise RuntimeError(f"wv_train_l1: {lib.wv_train_last_error().decode()}")
    return loss, da


class WatermarkTrainer:
    """The generator-update step of the reference's training loop on the HIP units (/root/reference/model/watermarking.py:340-421
    `_forward_train`, scripts/train.py:1296-1358 `_update_generator`), for the losses that live on this path:

        wm = G(x, msg) + x  ->  localisation + sequence augmentation (one launch, reference RNG order)  ->  D, L on the augmented audio
        loss = lambda_dec * DecodingLoss(D) + lambda_loc * LocalizationLoss(L) + lambda_wav * mean|wm - x|      (conf/base.yml:141-150)
        backward through D and L to the audio (conv_pre and every spectrogram branch), through the augmentation's select, through G;
        mean all-reduce of the three gradient arenas; clip_grad_norm_ on the GENERATOR's parameters only (train.py:1351-1353); AdamW
        on all three nets.

    Not on this path (they stay on PyTorch-ROCm): the audio effects between augmentation and detection (identity here),
    the mel / multi-scale-STFT losses and the discriminator; `extra_d_wm` is where their gradient towards the watermarked audio enters."""

    LAMBDAS = {"waveform/loss": 1000.0, "loc/loss": 100.0, "dec/loss": 10000.0}

    def __init__(self, cfgG, sdG, cfgD, sdD, cfgL, sdL, lr: float = 1e-4, max_norm: float = 1000.0, sample_rate: int = 16000,
                 window_duration: float = 0.1, device="cuda", effect_scheduler=None, apply_effect=None, effect_backward=None):
        """effect_scheduler: a waveverify_amd.effect_scheduler.EffectScheduler (watermarking.py:266-271); every step then selects
        effects as `_apply_adaptive_effects` does (watermarking.py:537: select_effects(batch size), i.e. at most one per known effect,
        applied to the FIRST clips of the batch) and feeds per-clip BER / mIoU back (`_update_effect_metrics`, watermarking.py:697-752).
        apply_effect(name, params, audio [1,1,T], mask [1,1,T]) -> (audio, mask) runs the non-identity effects; without it only
        'identity' can be scheduled.  effect_backward(name, params, d_out [1,1,T]) -> d_in carries the loss gradient back through an
        effect: the reference differentiates through its julius filters and torchaudio resampler (plain torch ops,
        effect_augmentation.py:1451-1501,1684-1870), so their gradient is the transposed filter (`effects.apply_effect_backward`);
        its SoX / codec / quantisation effects are straight-through Functions (:462-500) whose gradient is the identity -- which is
        also what a missing `effect_backward` means for every effect."""
        from .augment import TemporalAugmenter
        from .metrics import BER, MIOU
        self.G = GeneratorTrainer(cfgG, sdG, lr, max_norm, device)
        self.D = EncoderNetTrainer(cfgD, sdD, lr, max_norm, device)
        self.L = EncoderNetTrainer(cfgL, sdL, lr, max_norm, device)
        self.aug = TemporalAugmenter(sample_rate, window_duration)
        self.lambdas = dict(self.LAMBDAS)
        self.effect_scheduler, self.apply_effect, self.effect_backward = effect_scheduler, apply_effect, effect_backward
        self.ber_calculator, self.miou_calculator = BER(threshold=0.5), MIOU()
        self.effect_update_count = 0

    def state_dicts(self, parametrized: bool = False) -> Dict[str, Dict[str, torch.Tensor]]:
        """{generator, detector, locator: state dict} in the reference's stripped (default) or live weight-norm layout."""
        return {"generator": self.G.state_dict(parametrized), "detector": self.D.state_dict(parametrized), "locator": self.L.state_dict(parametrized)}

    def save_checkpoint(self, save_path, tag: str = "latest", step: Optional[int] = None, parametrized: bool = False):
        """Write the three nets as the reference's atomic checkpoint <save_path>/<tag>.pth (scripts/train.py:1589-1676: weight-norm
        parametrizations removed, temporary file renamed into place) -- the file `WaveVerify(checkpoint=<save_path>)` reads
        (waveverify/core.py:324-426), here and in the reference.  INFERENCE-compatible with the reference, not resume-compatible: its
        training resume (scripts/train.py:654-657,674,773-774) also wants `models.discriminator`, `schedulers`, `tracker` and torch
        AdamW state dicts under `optimizers`, none of which this path has.  Our flat AdamW moments therefore go under the private key
        `wv_amd_optimizers` ({net: {step, exp_avg, exp_avg_sq}} keyed by parameter name), so the reference sees NO optimizer state
        rather than a malformed one.  `parametrized=True` (the live g / v layout) is the only layout `from_checkpoint` restores the
        moments for, and the reference's INFERENCE loader does not read it (strict=False: every weight would stay at its initial
        value) -- write the default stripped layout for files the reference should load."""
        from .checkpoint import argbind_config, save_atomic_checkpoint
        opts = {k: n.optimizer_state() for k, n in (("generator", self.G), ("detector", self.D), ("locator", self.L))}
        cfgs = {"generator": self.G.cfg, "detector": self.D.cfg, "locator": self.L.cfg}
        return save_atomic_checkpoint(save_path, tag, self.state_dicts(parametrized), self.G.opt.t if step is None else step,
                                      argbind_config(cfgs), {OPT_KEY: opts})

    @classmethod
    def from_checkpoint(cls, path, **kwargs) -> "WatermarkTrainer":
        """Resume from an atomic (or legacy) checkpoint -- the reference's or one `save_checkpoint` wrote.  Stripped weights are put
        back under weight norm as torch does it (original0 = ||w||, original1 = w: the same function, `checkpoint.to_parametrized`);
        the optimizer moments are restored when the file holds them for the live layout (`save_checkpoint(parametrized=True)`),
        else AdamW starts fresh at the saved step count."""
        from .checkpoint import find_atomic_checkpoint_file, is_atomic_checkpoint, load_checkpoint, to_parametrized, _load
        sds, cfgs = load_checkpoint(path)
        missing = [k for k in ("generator", "detector", "locator") if k not in sds]
        if missing:
            raise FileNotFoundError(f"checkpoint {path} has no {', '.join(missing)} weights")
        live = {k: {n: t.numpy() for n, t in to_parametrized(sds[k], cfgs[k]).items()} for k in sds}
        tr = cls(cfgs["generator"], live["generator"], cfgs["detector"], live["detector"], cfgs["locator"], live["locator"], **kwargs)
        if is_atomic_checkpoint(path):
            ck = _load(find_atomic_checkpoint_file(path))
            step, opts = int(ck.get("step", 0) or 0), ck.get(OPT_KEY) or {}
            for k, net in (("generator", tr.G), ("detector", tr.D), ("locator", tr.L)):
                was_live = any("parametrizations.weight.original" in n for n in sds[k])     # moments of (g, v) fit these (g, v) only
                if not (was_live and isinstance(opts, dict) and net.load_optimizer_state(opts.get(k))):
                    net.opt.t = step
        return tr

    def _effects(self, wm_aug, mask):
        """-> (audio, mask, effects_applied): the straight-through effects on the first clips (watermarking.py:521-612)."""
        applied = self.effect_scheduler.select_effects(wm_aug.shape[0])
        out = wm_aug
        for i, (name, params) in enumerate(applied):
            if str(name) == "identity":
                continue
            if self.apply_effect is None:
                raise NotImplementedError(f"effect '{name}' was scheduled but no apply_effect callable is attached (only 'identity' runs here)")
            if out is wm_aug:
                out, mask = wm_aug.clone(), mask.clone()
            a_i, m_i = self.apply_effect(str(name), params, wm_aug[i:i + 1], mask[i:i + 1])
            if a_i.shape != wm_aug[i:i + 1].shape:
                raise RuntimeError("apply_effect must keep the clip length (the reference adjusts lengths back, effect_augmentation.py:118-232)")
            out[i:i + 1], mask[i:i + 1] = a_i, m_i
        return out, mask, applied

    def _update_effect_metrics(self, logits_d, logits_l, msg, mask, applied) -> None:
        loc_bin = (logits_l > 0.5).float()                       # the RAW locator output at 0.5, as the reference does (watermarking.py:717)
        for i, (name, params) in enumerate(applied):
            ber = self.ber_calculator(logits_d[i:i + 1], msg[i:i + 1], mask[i:i + 1])
            miou = self.miou_calculator(loc_bin[i:i + 1], mask[i:i + 1])
            self.effect_scheduler.update_effect_metrics(name, params, float(ber), float(miou))
            self.effect_update_count += 1

    def step(self, x: torch.Tensor, msg: torch.Tensor, extra_d_wm: Optional[torch.Tensor] = None, augment: bool = True):
        x, msg = _f(x), _f(msg)
        if msg.dim() == 1:
            msg = msg[None]
        if msg.shape[0] != x.shape[0]:                                 # one (or a shorter list of) message(s) for the batch: repeated as the
            msg = msg.repeat(-(-x.shape[0] // msg.shape[0]), 1)[: x.shape[0]].contiguous()   # reference's forward does (watermarking.py:320-329)
        lam = self.lambdas
        wm = self.G.forward(x, msg)
        if augment:
            sig, mask, _, stats = self.aug.forward(x, wm)
            wm_aug = sig.audio_data
        else:
            wm_aug, mask, stats = wm, torch.ones_like(wm), {}
        applied = []
        if self.effect_scheduler is not None:
            wm_aug, mask, applied = self._effects(wm_aug, mask)
        logits_d = self.D.forward(wm_aug)
        dec, dzD = bce_logits(logits_d, mask, msg, grad_scale=lam["dec/loss"])
        # each net's gradient exchange starts inside its own backward (bucket by bucket) and runs under everything that follows:
        # the detector's under the locator's passes and the generator's backward, the locator's under the generator's backward.
        # (Measured under gloo only so far: the RCCL overlap has not run on hardware)
        try:
            self.D.begin_reduce()
            d_aug = self.D.backward(dzD, need_dx=True)
            self.D._reducer.flush()
            logits_l = self.L.forward(wm_aug)
            loc, dzL = bce_logits(logits_l, mask, None, grad_scale=lam["loc/loss"])
            self.L.begin_reduce()
            d_aug = d_aug + self.L.backward(dzL, need_dx=True)
            self.L._reducer.flush()
            if applied:
                self._update_effect_metrics(logits_d, logits_l, msg, mask, applied)
                stats = dict(stats, selected_effects=applied)
                if self.effect_backward is not None:                      # back through each clip's effect (identity when there is no hook)
                    for i, (name, params) in enumerate(applied):
                        if str(name) != "identity":
                            d_aug[i:i + 1] = self.effect_backward(str(name), params, d_aug[i:i + 1].clone())
            d_wm = self.aug.backward(d_aug) if augment else d_aug
            wav, d_wav = l1_loss(wm, x, grad_scale=lam["waveform/loss"])
            d_wm = d_wm + d_wav
            if extra_d_wm is not None:
                d_wm = d_wm + _f(extra_d_wm)
            self.G.begin_reduce()
            self.G.backward(d_wm)
        except BaseException:
            # a rank that raises between the first begin_reduce and the last finish_reduce (a user hook, an augmentation) must not leave its
            # peers blocked in finish_reduce on collectives it never issues: it still issues EVERY bucket of the step, in the step's order
            # (D, L, G), waits for them, drops the reducers, and only then re-raises -- the peers' step completes (with this rank's
            # unfinished gradients in the mean: the step is lost either way) and the next step starts clean on every rank
            for net in (self.D, self.L, self.G):
                net.abort_reduce()
            raise
        for net in (self.G, self.D, self.L):
            net.finish_reduce()
        norm = self.G.opt.step(self.G.arena, self.G.grads, self.G.max_norm)          # clipping: the generator only
        self.D.opt.step(self.D.arena, self.D.grads, None)
        self.L.opt.step(self.L.arena, self.L.grads, None)
        total = lam["dec/loss"] * dec + lam["loc/loss"] * loc + lam["waveform/loss"] * wav
        return {"loss": total, "dec/loss": dec, "loc/loss": loc, "waveform/loss": wav, "grad_norm": norm, "stats": stats}
