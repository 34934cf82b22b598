"""Hyper-parameters of the three WaveVerify nets on the embed/detect hot path.

Field names and defaults follow the reference constructors:
  Generator  /root/reference/model/generator.py:63-104
  Detector   /root/reference/model/detector.py:82-114
  Locator    /root/reference/model/locator.py:84-115
Only the configuration the reference can actually construct is supported
("Config / flags" row): weight_norm, causal, ELU, skip='identity',
spec='stft' with log compression, inout_norm, encoder_l2norm, bias=True.  Everything else
in the reference constructors selects code that the shipped config never reaches.
"""
from __future__ import annotations

from dataclasses import dataclass, field, asdict
from typing import List

WAV_STD = 0.1122080159                       # modules/seanet.py:631
RES_SCALE = 0.5773502691896258               # model/generator.py:73
SPEC_MEANS = [-4.554, -4.315, -4.021, -3.726, -3.477]   # modules/seanet.py:632
SPEC_STDS = [2.830, 2.837, 2.817, 2.796, 2.871]         # modules/seanet.py:633

KINDS = ("generator", "detector", "locator")


@dataclass
class NetConfig:
    kind: str = "generator"
    sample_rate: int = 16000
    dimension: int = 128
    msg_dimension: int = 16
    channels_enc: int = 64
    channels_dec: int = 96
    n_fft_base: int = 64
    n_residual_enc: int = 2
    n_residual_dec: int = 3
    res_scale_enc: float = RES_SCALE
    res_scale_dec: float = RES_SCALE
    strides: List[int] = field(default_factory=lambda: [8, 5, 4, 2])
    kernel_size: int = 5
    last_kernel_size: int = 5
    residual_kernel_size: int = 5
    dilation_base: int = 1
    zero_init: bool = True
    nbits: int = 16
    output_dim: int = 32
    embedding_dim: int = 64
    embedding_layers: int = 2
    freq_bands: int = 4
    wav_std: float = WAV_STD
    spec_means: List[float] = field(default_factory=lambda: list(SPEC_MEANS))
    spec_stds: List[float] = field(default_factory=lambda: list(SPEC_STDS))

    def __post_init__(self):
        if self.kind not in KINDS:
            raise ValueError(f"kind must be one of {KINDS}, got {self.kind!r}")
        if len(self.strides) + 1 > len(self.spec_means):
            raise ValueError("spec_means/spec_stds must hold len(strides)+1 entries")

    # ---- derived quantities -------------------------------------------------
    @property
    def ratios_enc(self) -> List[int]:
        """Encoder walks the strides reversed (modules/seanet.py:646)."""
        return list(reversed(self.strides))

    @property
    def hop_length(self) -> int:
        h = 1
        for s in self.strides:
            h *= s
        return h

    @property
    def has_decoder(self) -> bool:
        return self.kind == "generator"

    @property
    def head_bits(self) -> int:
        """Channels of the detector/locator last_layer."""
        return self.nbits if self.kind == "detector" else 1

    def to_dict(self):
        return asdict(self)


def generator_config(**kw) -> NetConfig:
    return NetConfig(kind="generator", **kw)


def detector_config(**kw) -> NetConfig:
    return NetConfig(kind="detector", **kw)


def locator_config(**kw) -> NetConfig:
    """Locator defaults: model/locator.py:84-115."""
    base = dict(dimension=64, channels_enc=32, n_residual_enc=1, strides=[8, 4])
    base.update(kw)
    return NetConfig(kind="locator", **base)


def default_config(kind: str, **kw) -> NetConfig:
    return {"generator": generator_config, "detector": detector_config,
            "locator": locator_config}[kind](**kw)
