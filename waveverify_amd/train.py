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
from typing import Dict

import torch

from . import _lib


def _f(t: torch.Tensor) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError("training tensors must live on the GPU")
    return t.float().contiguous()


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

    def backward(self, x, p, pre_scale: float, dy, pre_elu: bool = True, need_dx: bool = True):
        x, dy = _f(x), _f(dy)
        B, _, T = x.shape
        g_pw, v_pw, g_dw, v_dw, _ = self._p(p)
        out = dict(dx=torch.empty_like(x) if need_dx else None, dg_pw=torch.empty_like(g_pw), dv_pw=torch.empty_like(v_pw),
                   dg_dw=torch.empty_like(g_dw), dv_dw=torch.empty_like(v_dw), db_dw=torch.empty_like(g_dw))
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

    def backward(self, x, p, in_scale: float, dy, need_dx: bool = False):
        x, dy = _f(x), _f(dy)
        B, _, T = x.shape
        g, v = _f(p["g"]).reshape(self.C), _f(p["v"]).reshape(self.C, self.ks)
        out = dict(dx=torch.empty_like(x) if need_dx else None, dg=torch.empty_like(g), dv=torch.empty_like(v), db=torch.empty_like(g))
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

    def backward(self, P, p, scale_param, res_scale: float, dy):
        P, dy = _f(P), _f(dy)
        B, _, T = dy.shape
        g, v = _f(p["g"]).reshape(self.C), _f(p["v"]).reshape(self.C, self.F)
        sp = None if scale_param is None else _f(scale_param).reshape(1)
        out = dict(dg=torch.empty_like(g), dv=torch.empty_like(v), d_scale_param=None if sp is None else torch.empty(1, device=dy.device))
        ws = torch.empty(int(self._lib.wv_train_spec_workspace_bytes(self._h, B, T)), dtype=torch.uint8, device=dy.device)
        self._check(self._lib.wv_train_spec_backward(
            self._h, P.data_ptr(), g.data_ptr(), v.data_ptr(), None if sp is None else sp.data_ptr(), float(res_scale), dy.data_ptr(),
            out["dg"].data_ptr(), out["dv"].data_ptr(), None if sp is None else out["d_scale_param"].data_ptr(), B, T, ws.data_ptr(),
            ws.numel(), TrainHalf._stream()), "wv_train_spec_backward")
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

    def backward(self, x, p, dy):
        x, dy = _f(x), _f(dy)
        B, _, T = x.shape
        g_dw, v_dw, g_pw, v_pw, b = self._p(p)
        out = dict(dx=torch.empty_like(x), dg_dw=torch.empty_like(g_dw), dv_dw=torch.empty_like(v_dw), dg_pw=torch.empty_like(g_pw),
                   dv_pw=torch.empty_like(v_pw), db=torch.empty_like(b))
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

    def backward(self, z, p, dlogits):
        z, dl = _f(z), _f(dlogits)
        B, _, N = z.shape
        w_rev, b_rev, w_last, _ = self._p(p)
        out = dict(dz=torch.empty_like(z), dw_rev=torch.empty_like(w_rev), db_rev=torch.empty_like(b_rev), dw_last=torch.empty_like(w_last),
                   db_last=torch.empty(self.nb, device=z.device))
        ws = self._ws(B, N, z.device)
        self._check(self._lib.wv_train_head_backward(
            self._h, z.data_ptr(), w_rev.data_ptr(), b_rev.data_ptr(), w_last.data_ptr(), dl.data_ptr(), out["dz"].data_ptr(),
            out["dw_rev"].data_ptr(), out["db_rev"].data_ptr(), out["dw_last"].data_ptr(), out["db_last"].data_ptr(), B, N, dl.shape[2],
            ws.data_ptr(), ws.numel(), TrainHalf._stream()), "wv_train_head_backward")
        return out


class _HalfParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("g_pw", "v_pw", "g_dw", "v_dw", "bias")]


class _HalfGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("dg_pw", "dv_pw", "dg_dw", "dv_dw", "db")]


class TrainBlock:
    """Whole SEANetResnetBlock (/root/reference/modules/seanet.py:245-281, identity shortcut):
    y = x + res_scale * res_scale_param * half2(half1(pre_scale * x)); `res_scale_param` is the trainable [1]
    tensor of zero_init blocks or None."""

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

    def backward(self, x, ps, res_scale_param, pre_scale: float, res_scale: float, dy, saved):
        x, dy = _f(x), _f(dy)
        B, _, T = x.shape
        arr, keep = self._params(ps)
        rsp = None if res_scale_param is None else _f(res_scale_param).reshape(1)
        dev = x.device
        grads, garr = [], (_HalfGrads * 2)()
        for i in range(2):
            g = dict(dg_pw=torch.empty(self.C, device=dev), dv_pw=torch.empty(self.C, self.C, device=dev), dg_dw=torch.empty(self.C, device=dev),
                     dv_dw=torch.empty(self.C, 5, device=dev), db_dw=torch.empty(self.C, device=dev))
            grads.append(g)
            garr[i] = _HalfGrads(g["dg_pw"].data_ptr(), g["dv_pw"].data_ptr(), g["dg_dw"].data_ptr(), g["dv_dw"].data_ptr(), g["db_dw"].data_ptr())
        dx = torch.empty_like(x)
        drsp = None if rsp is None else torch.empty(1, device=dev)
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
