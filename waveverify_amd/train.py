"""First slice of the training step on the GPU: one SEANetResnetBlock half with live weight
normalisation, forward and backward, behind the C ABI (wv_train_half_*, include/waveverify_hip.h).

    half = TrainHalf(C); y = half.forward(x, params, pre_scale); grads = half.backward(x, params, pre_scale, dy)

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
