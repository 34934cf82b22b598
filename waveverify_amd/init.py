"""Seeded random initialisation of WaveVerify state dicts (no checkpoint is shipped upstream:
/root/reference/waveverify/utils.py:45-52 has an empty download URL).

This is the analogue of constructing the reference modules with fresh weights
(/root/reference/modules/conv.py:397-399 kaiming-normal convs; seanet.py:849-857), with two
deliberate differences so that every branch of the forward pass carries signal:
  * `res_scale_param` / `scale_param` are drawn from U(0.5, 1.5) instead of 0
    (zero would switch the residual and spectrogram branches off, seanet.py:233-235,452-454);
  * FiLM gamma biases are drawn around 1 and the last_layer bias is wide, so the message
    actually modulates the features and bit decisions have comfortable margins.
numpy Philox streams are used so the same (config, seed) gives the same weights everywhere.
"""
from __future__ import annotations

import zlib
from typing import Dict

import numpy as np

from .config import NetConfig
from .params import param_specs


def _rng(seed: int, key: str) -> np.random.Generator:
    return np.random.Generator(np.random.Philox(key=[seed, zlib.crc32(key.encode())]))


def random_state_dict(cfg: NetConfig, seed: int = 0, parametrized: bool = False
                      ) -> Dict[str, np.ndarray]:
    """Return {key: float32 ndarray}.  With parametrized=True weight-normed tensors are
    emitted as `...parametrizations.weight.original0` (g) / `original1` (v) pairs, the layout
    torch's weight_norm parametrization keeps in a live model."""
    sd: Dict[str, np.ndarray] = {}
    for key, shape, role in param_specs(cfg):
        r = _rng(seed, f"{cfg.kind}/{key}")
        if role == "wn":
            fan_in = int(np.prod(shape[1:]))
            v = r.normal(0.0, 1.0 / np.sqrt(fan_in), size=shape).astype(np.float32)
            nrm = np.sqrt((v.reshape(shape[0], -1).astype(np.float64) ** 2).sum(1))
            g = (nrm * r.uniform(0.8, 1.25, size=shape[0])).astype(np.float32)
            if parametrized:
                base = key[: -len("weight")]
                sd[base + "parametrizations.weight.original0"] = g.reshape(shape[0], 1, 1)
                sd[base + "parametrizations.weight.original1"] = v
            else:
                w = v * (g / nrm.astype(np.float32)).reshape(shape[0], 1, 1)
                sd[key] = w.astype(np.float32)
        elif role == "plain":
            fan_in = int(np.prod(shape[1:])) if not key.startswith("reverse_convolution") \
                else shape[0]
            std = 1.0 / np.sqrt(fan_in)
            if "film_layers" in key:
                std = 0.25 / np.sqrt(fan_in)
            sd[key] = r.normal(0.0, std, size=shape).astype(np.float32)
        elif role == "bias":
            if "gamma_layer" in key:
                val = r.uniform(0.6, 1.4, size=shape)
            elif key == "last_layer.bias":
                # keep every bit's time-averaged probability well away from the 0.5 threshold
                val = r.uniform(1.0, 2.5, size=shape) * r.choice([-1.0, 1.0], size=shape)
            elif "conv_post.2" in key:
                val = r.normal(0.0, 1.0, size=shape)        # seanet.py:825-828
            else:
                val = r.normal(0.0, 0.05, size=shape)
            sd[key] = val.astype(np.float32)
        elif role == "scalar":
            sd[key] = r.uniform(0.5, 1.5, size=shape).astype(np.float32)
        else:  # pragma: no cover
            raise AssertionError(role)
    return sd


def synthetic_clips(B: int, T: int, seed: int = 1234, nbits: int = 16):
    """SURVEY.md section 8(d) synthetic inputs: x = clip(0.1*N(0,1), -1, 1), msg ~ Bernoulli(0.5)."""
    r = np.random.default_rng(seed)
    x = np.clip(0.1 * r.standard_normal((B, 1, T), dtype=np.float32), -1.0, 1.0)
    msg = r.integers(0, 2, size=(B, nbits)).astype(np.float32)
    return x.astype(np.float32), msg
