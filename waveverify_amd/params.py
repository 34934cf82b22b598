"""State-dict key grammar of the reference nets, derived from a NetConfig.

Keys are the *stripped* ones the reference's checkpoints hold after
`parametrize.remove_parametrizations` (/root/reference/scripts/train.py:1624-1629,
/root/reference/waveverify/core.py:116-139): `...conv.conv.weight` etc.  Module indices follow
the nn.Sequential layouts in /root/reference/modules/seanet.py (encoder :657-846,
decoder :1067-1204) and the heads in model/detector.py:209-218 / model/locator.py:174-183.

Each entry: (key, shape, role) with role in
  'wn'      weight-normed conv / conv-transpose weight (may arrive as g,v pair)
  'plain'   plain weight (Linear, reverse_convolution, last_layer)
  'bias'    bias vector
  'scalar'  1-element learnable scale (only when zero_init)
"""
from __future__ import annotations

from typing import List, Tuple

from .config import NetConfig

Spec = Tuple[str, Tuple[int, ...], str]


def _resblock(prefix: str, dim: int, k: int, zero_init: bool) -> List[Spec]:
    out: List[Spec] = []
    if zero_init:
        out.append((f"{prefix}.res_scale_param", (1,), "scalar"))
    for pw, dw in ((1, 2), (4, 5)):
        out.append((f"{prefix}.block.{pw}.conv.conv.weight", (dim, dim, 1), "wn"))
        out.append((f"{prefix}.block.{dw}.conv.conv.bias", (dim,), "bias"))
        out.append((f"{prefix}.block.{dw}.conv.conv.weight", (dim, 1, k), "wn"))
    return out


def encoder_specs(cfg: NetConfig) -> List[Spec]:
    C0 = cfg.channels_enc
    out: List[Spec] = [
        ("encoder.conv_pre.1.conv.conv.bias", (C0,), "bias"),
        ("encoder.conv_pre.1.conv.conv.weight", (C0, 1, cfg.kernel_size), "wn"),
    ]
    mult = 1
    for s, _ in enumerate(cfg.ratios_enc):
        for j in range(cfg.n_residual_enc):
            out += _resblock(f"encoder.blocks.{s}.{j}", mult * C0, cfg.residual_kernel_size,
                             cfg.zero_init)
        mult *= 2
    mult = 1
    for s, _ in enumerate(cfg.ratios_enc):
        n_fft = mult * cfg.n_fft_base
        if cfg.zero_init:
            out.append((f"encoder.spec_blocks.{s}.scale_param", (1,), "scalar"))
        out.append((f"encoder.spec_blocks.{s}.layer.conv.conv.weight",
                    (mult * C0, n_fft // 2 + 1, 1), "wn"))
        mult *= 2
    mult = 1
    for s, r in enumerate(cfg.ratios_enc):
        C = mult * C0
        out.append((f"encoder.downsample.{s}.2.conv.conv.weight", (2 * C, C, 1), "wn"))
        out.append((f"encoder.downsample.{s}.3.conv.conv.bias", (2 * C,), "bias"))
        out.append((f"encoder.downsample.{s}.3.conv.conv.weight", (2 * C, 1, 2 * r), "wn"))
        mult *= 2
    C = mult * C0
    n_fft = mult * cfg.n_fft_base
    if cfg.zero_init:
        out.append(("encoder.spec_post.scale_param", (1,), "scalar"))
    out.append(("encoder.spec_post.layer.conv.conv.weight", (C, n_fft // 2 + 1, 1), "wn"))
    out.append(("encoder.conv_post.1.conv.conv.weight", (C, 1, cfg.last_kernel_size), "wn"))
    out.append(("encoder.conv_post.2.conv.conv.bias", (cfg.dimension,), "bias"))
    out.append(("encoder.conv_post.2.conv.conv.weight", (cfg.dimension, C, 1), "wn"))
    # message MLP + FiLM exist in every SEANetEncoder (seanet.py:831-846); only the
    # generator's forward uses them (msg is None for detector/locator, seanet.py:907).
    E = cfg.embedding_dim
    idx = [0] + [1 + 2 * i for i in range(cfg.embedding_layers)]
    dims = [cfg.msg_dimension] + [E] * cfg.embedding_layers
    for i, d in zip(idx, dims):
        out.append((f"encoder.msg_embedding.{i}.weight", (E, d), "plain"))
        out.append((f"encoder.msg_embedding.{i}.bias", (E,), "bias"))
    for s in range(len(cfg.strides)):
        for b in range(cfg.freq_bands):
            for nm in ("gamma", "beta"):
                out.append((f"encoder.film_layers.{s}.{b}.{nm}_layer.weight", (1, E), "plain"))
                out.append((f"encoder.film_layers.{s}.{b}.{nm}_layer.bias", (1,), "bias"))
    return out


def decoder_layout(cfg: NetConfig):
    """Indices into decoder.model (modules/seanet.py:1067-1204): returns
    (idx_pw0, idx_dw0, [(idx_convtr, idx_pw, [idx_res...], ratio, C_in)], idx_last)."""
    n = 2
    ups = []
    mult = 2 ** len(cfg.strides)
    for r in cfg.strides:
        ct, pw = n + 2, n + 3
        res = [n + 4 + j for j in range(cfg.n_residual_dec)]
        ups.append((ct, pw, res, r, mult * cfg.channels_dec))
        n += 4 + cfg.n_residual_dec
        mult //= 2
    return 0, 1, ups, n + 2


def decoder_specs(cfg: NetConfig) -> List[Spec]:
    Cd = cfg.channels_dec
    i_pw0, i_dw0, ups, i_last = decoder_layout(cfg)
    Ctop = (2 ** len(cfg.strides)) * Cd
    out: List[Spec] = [
        (f"decoder.model.{i_pw0}.conv.conv.weight", (Ctop, cfg.dimension, 1), "wn"),
        (f"decoder.model.{i_dw0}.conv.conv.bias", (Ctop,), "bias"),
        (f"decoder.model.{i_dw0}.conv.conv.weight", (Ctop, 1, cfg.kernel_size), "wn"),
    ]
    for ct, pw, res, r, C in ups:
        out.append((f"decoder.model.{ct}.convtr.convtr.weight", (C, 1, 2 * r), "wn"))
        out.append((f"decoder.model.{pw}.conv.conv.bias", (C // 2,), "bias"))
        out.append((f"decoder.model.{pw}.conv.conv.weight", (C // 2, C, 1), "wn"))
        for i in res:
            out += _resblock(f"decoder.model.{i}", C // 2, cfg.residual_kernel_size,
                             cfg.zero_init)
    out.append((f"decoder.model.{i_last}.conv.conv.bias", (1,), "bias"))
    out.append((f"decoder.model.{i_last}.conv.conv.weight", (1, Cd, cfg.last_kernel_size), "wn"))
    return out


def head_specs(cfg: NetConfig) -> List[Spec]:
    hop = cfg.hop_length
    return [
        ("reverse_convolution.weight", (cfg.dimension, cfg.output_dim, hop), "plain"),
        ("reverse_convolution.bias", (cfg.output_dim,), "bias"),
        ("last_layer.weight", (cfg.head_bits, cfg.output_dim, 1), "plain"),
        ("last_layer.bias", (cfg.head_bits,), "bias"),
    ]


def param_specs(cfg: NetConfig) -> List[Spec]:
    out = encoder_specs(cfg)
    if cfg.has_decoder:
        out += decoder_specs(cfg)
    else:
        out += head_specs(cfg)
    return out


def param_count(cfg: NetConfig) -> int:
    n = 0
    for _, shape, role in param_specs(cfg):
        p = 1
        for d in shape:
            p *= d
        n += p
        if role == "wn":           # the g vector of the (g, v) parametrization
            n += shape[0]
    return n
