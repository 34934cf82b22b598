"""Host-side handles on the HIP nets: Generator / Detector / Locator behind the C ABI.

PyTorch is plumbing here (device memory, streams); every FLOP of the forward pass runs in
libwaveverify_hip.so.  The call surface mirrors the reference modules
(/root/reference/model/generator.py:360-423, detector.py:366-391, locator.py:268-299) on plain
tensors: generator(x[B,1,T], msg[B|1,16]) -> delta, detector(x) -> logits[B,nbits,T],
locator(x) -> logits[B,1,T].
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Mapping, Optional, Tuple

import numpy as np
import torch

from . import _lib
from .config import NetConfig

_TAG0 = "parametrizations.weight.original0"
_TAG1 = "parametrizations.weight.original1"


def _np32(v) -> np.ndarray:
    if isinstance(v, torch.Tensor):
        v = v.detach().cpu().numpy()
    return np.ascontiguousarray(np.asarray(v, dtype=np.float32))


def _fill_config(cfg: NetConfig) -> _lib.WvConfig:
    c = _lib.WvConfig()
    c.kind = _lib.WV_KIND[cfg.kind]
    for f in ("dimension", "msg_dimension", "channels_enc", "channels_dec", "n_fft_base",
              "n_residual_enc", "n_residual_dec", "kernel_size", "last_kernel_size",
              "residual_kernel_size", "dilation_base", "nbits", "output_dim", "embedding_dim",
              "embedding_layers", "freq_bands"):
        setattr(c, f, int(getattr(cfg, f)))
    c.zero_init = int(bool(cfg.zero_init))
    c.n_strides = len(cfg.strides)
    if c.n_strides > _lib.WV_MAX_STRIDES:
        raise ValueError("too many strides")
    for i, s in enumerate(cfg.strides):
        c.strides[i] = int(s)
    c.res_scale_enc = cfg.res_scale_enc
    c.res_scale_dec = cfg.res_scale_dec
    c.wav_std = cfg.wav_std
    for i in range(c.n_strides):
        c.spec_means[i] = cfg.spec_means[i]
        c.spec_stds[i] = cfg.spec_stds[i]
    c.spec_means[_lib.WV_MAX_STRIDES] = cfg.spec_means[-1]     # spec_post uses [-1] (seanet.py:789)
    c.spec_stds[_lib.WV_MAX_STRIDES] = cfg.spec_stds[-1]
    return c


class HipNet:
    """One WaveVerify net resident on one GPU."""


    def __init__(self, cfg: NetConfig, state_dict: Mapping[str, object], device="cuda",
                 strict: bool = False):
        self.cfg = cfg
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("waveverify_amd runs on an MI355X only: device must be 'cuda[:i]'")
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device visible (torch.cuda.is_available() is False)")
        self._lib = _lib.load()
        self._h = C.c_void_p()
        ccfg = _fill_config(cfg)
        _lib.check(self._lib.wv_model_create(C.byref(ccfg), C.byref(self._h)), "wv_model_create")
        self._ws: Dict[int, torch.Tensor] = {}          # per-stream workspaces
        self._retired: list = []                        # outgrown buffers, kept alive (see _workspace)
        self.unexpected_keys = []
        with torch.cuda.device(self.device):
            self._load(state_dict, strict)
            _lib.check(self._lib.wv_model_finalize(self._h), "wv_model_finalize")

    # ------------------------------------------------------------------ weights
    def param_table(self) -> Dict[str, Tuple[Tuple[int, ...], bool]]:
        n = self._lib.wv_model_num_params(self._h)
        out = {}
        name = C.create_string_buffer(256)
        shape = (C.c_int64 * 4)()
        nd, wn = C.c_int(), C.c_int()
        for i in range(n):
            _lib.check(self._lib.wv_model_param_info(self._h, i, name, 256, shape, C.byref(nd),
                                                     C.byref(wn)))
            out[name.value.decode()] = (tuple(shape[: nd.value]), bool(wn.value))
        return out

    def _load(self, sd: Mapping[str, object], strict: bool) -> None:
        """load_state_dict for both key layouts (waveverify/core.py:324-426 strict=False on the
        stripped layout; `parametrizations.weight.original0/1` pairs are folded in the library)."""
        table = self.param_table()
        for k, v in sd.items():
            kb = k.encode()
            if k.endswith(_TAG1):
                continue
            if k.endswith(_TAG0):
                base = k[: -len(_TAG0)]
                g, vv = _np32(v), _np32(sd[base + _TAG1])
                _lib.check(self._lib.wv_model_set_param_wn(
                    self._h, (base + "weight").encode(), g.ctypes.data, g.size, vv.ctypes.data,
                    vv.size), f"set_param_wn({base}weight)")
            elif k.endswith("spec.weight"):
                a = _np32(v)
                _lib.check(self._lib.wv_model_set_stft_basis(self._h, kb, a.ctypes.data, a.size),
                           f"set_stft_basis({k})")
            elif k in table:
                a = _np32(v)
                _lib.check(self._lib.wv_model_set_param(self._h, kb, a.ctypes.data, a.size),
                           f"set_param({k})")
            else:
                self.unexpected_keys.append(k)
        if strict and self.unexpected_keys:
            raise RuntimeError(f"unexpected keys in state dict: {self.unexpected_keys[:5]} ...")

    # ------------------------------------------------------------------ plumbing
    def reserve(self, B: int, T: int) -> None:
        """Pre-allocate the current stream's workspace for batches up to (B, T) (e.g. before capturing
        a forward into a HIP graph outside torch.cuda.graph, where a capture must not allocate)."""
        self._workspace(B, T)

    def _workspace(self, B: int, T: int) -> torch.Tensor:
        """One workspace per (net, stream): calls on different streams never share scratch memory.
        A buffer that is outgrown is RETIRED, not freed -- a captured graph (or a launch still queued
        on the stream) may hold its address -- and stays alive as long as the net does."""
        need = int(self._lib.wv_workspace_bytes(self._h, B, T))
        key = int(torch.cuda.current_stream(self.device).cuda_stream)
        ws = self._ws.get(key)
        if ws is None or ws.numel() < need:
            if ws is not None:
                self._retired.append(ws)
            ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            self._ws[key] = ws
        return ws

    def _prep(self, x: torch.Tensor) -> torch.Tensor:
        if x.dim() == 2:
            x = x.unsqueeze(1)
        if x.dim() != 3 or x.shape[1] != 1:
            raise ValueError(f"expected audio of shape [B,1,T], got {tuple(x.shape)}")
        return x.to(self.device, torch.float32).contiguous()

    @staticmethod
    def _stream() -> C.c_void_p:
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    @property
    def hop_length(self) -> int:
        return self.cfg.hop_length

    # ------------------------------------------------------------------ forward passes
    def generator(self, x: torch.Tensor, msg: torch.Tensor, add_input: bool = False, precision: str = "f32") -> torch.Tensor:
        """delta = G(x, msg) [B,1,T]; with add_input the watermarked audio delta + x
        (model/watermarking.py:423-441).  precision="f16": the f16-operand / f32-accumulate throughput mode (csrc/wv_h16.hip)."""
        if self.cfg.kind != "generator":
            raise RuntimeError("not a generator")
        if precision not in ("f32", "f16"):
            raise ValueError("precision must be 'f32' or 'f16'")
        x = self._prep(x)
        B, _, T = x.shape
        msg = msg.to(self.device).float().contiguous()          # seanet.py:909 casts to float
        if msg.dim() != 2 or msg.shape[1] != self.cfg.msg_dimension:
            raise ValueError(f"msg must be [B,{self.cfg.msg_dimension}], got {tuple(msg.shape)}")
        if msg.shape[0] not in (1, B):                          # watermarking.py:320-329
            reps = -(-B // msg.shape[0])
            msg = msg.repeat(reps, 1)[:B].contiguous()
        out = torch.empty_like(x)
        with torch.cuda.device(self.device):
            ws = self._workspace(B, T)
            fn = self._lib.wv_generator_forward_f16 if precision == "f16" else self._lib.wv_generator_forward
            _lib.check(fn(
                self._h, x.data_ptr(), msg.data_ptr(), msg.shape[0], out.data_ptr(),
                int(add_input), B, T, ws.data_ptr(), ws.numel(), self._stream()),
                "wv_generator_forward" + ("_f16" if precision == "f16" else ""))
        return out

    def _head(self, x, want_logits: bool, want_mean: bool, precision: str = "f32"):
        x = self._prep(x)
        B, _, T = x.shape
        nb = self.cfg.head_bits
        logits = torch.empty((B, nb, T), dtype=torch.float32, device=self.device) if want_logits else None
        mean = torch.empty((B, nb), dtype=torch.float32, device=self.device) if want_mean else None
        with torch.cuda.device(self.device):
            ws = self._workspace(B, T)
            if precision not in ("f32", "f16"):
                raise ValueError("precision must be 'f32' or 'f16'")
            if self.cfg.kind == "detector":
                fn = self._lib.wv_detector_forward_f16 if precision == "f16" else self._lib.wv_detector_forward
                _lib.check(fn(
                    self._h, x.data_ptr(), logits.data_ptr() if want_logits else None,
                    mean.data_ptr() if want_mean else None, B, T, ws.data_ptr(), ws.numel(),
                    self._stream()), "wv_detector_forward" + ("_f16" if precision == "f16" else ""))
            elif self.cfg.kind == "locator":
                fn = self._lib.wv_locator_forward_f16 if precision == "f16" else self._lib.wv_locator_forward
                _lib.check(fn(
                    self._h, x.data_ptr(), logits.data_ptr(), B, T, ws.data_ptr(), ws.numel(),
                    self._stream()), "wv_locator_forward" + ("_f16" if precision == "f16" else ""))
            else:
                raise RuntimeError("generator has no detection head")
        return logits, mean

    def detector(self, x: torch.Tensor, precision: str = "f32") -> torch.Tensor:
        """Detector.forward: logits [B, nbits, T].  precision="f16": the f16-operand / f32-accumulate throughput mode (csrc/wv_h16.hip)."""
        return self._head(x, True, False, precision)[0]

    def detector_mean_prob(self, x: torch.Tensor, precision: str = "f32") -> torch.Tensor:
        """mean_t sigmoid(logits) [B, nbits] without materialising the logits (core.py:577-580)."""
        if self.cfg.kind != "detector":
            raise RuntimeError("not a detector")
        return self._head(x, False, True, precision)[1]

    def locator(self, x: torch.Tensor, precision: str = "f32") -> torch.Tensor:
        """Locator.forward: logits [B, 1, T].  precision="f16": the f16-operand / f32-accumulate throughput mode."""
        if self.cfg.kind != "locator":
            raise RuntimeError("not a locator")
        return self._head(x, True, False, precision)[0]

    def encoder(self, x: torch.Tensor, msg: Optional[torch.Tensor] = None) -> torch.Tensor:
        """SEANetEncoder.forward: latent [B, dimension, ceil(T/hop)]."""
        x = self._prep(x)
        B, _, T = x.shape
        Fr = -(-T // self.hop_length)
        lat = torch.empty((B, self.cfg.dimension, Fr), dtype=torch.float32, device=self.device)
        mp, rows = None, 0
        if msg is not None:
            msg = msg.to(self.device).float().contiguous()
            mp, rows = msg.data_ptr(), msg.shape[0]
        with torch.cuda.device(self.device):
            ws = self._workspace(B, T)
            _lib.check(self._lib.wv_encoder_forward(self._h, x.data_ptr(), mp, rows, lat.data_ptr(),
                                                    B, T, ws.data_ptr(), ws.numel(), self._stream()),
                       "wv_encoder_forward")
        return lat

    def film(self, msg: torch.Tensor, B: int) -> torch.Tensor:
        msg = msg.to(self.device).float().contiguous()
        out = torch.empty((B, len(self.cfg.strides), self.cfg.freq_bands, 2), dtype=torch.float32,
                          device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.wv_model_film(self._h, msg.data_ptr(), msg.shape[0], out.data_ptr(),
                                               B, self._stream()), "wv_model_film")
        return out

    def __call__(self, *a, **kw):
        return {"generator": self.generator, "detector": self.detector,
                "locator": self.locator}[self.cfg.kind](*a, **kw)

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                self._lib.wv_model_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass
