"""WaveVerify: the reference's package API on the MI355X-native path.

Same constructor, methods, return types and error wrapping as
/root/reference/waveverify/core.py:51-729 — `WaveVerify(checkpoint, device)`,
`.embed/.detect/.locate/.verify` — plus batched tensor entry points (`embed_batch`,
`detect_batch`, `locate_batch`) that the file-based methods are thin wrappers of.  The forward
passes run in libwaveverify_hip.so (see nets.py); there is no CPU fallback.
"""
from __future__ import annotations

import logging
from pathlib import Path
from types import SimpleNamespace
from typing import Dict, Mapping, Optional, Tuple, Union

import numpy as np
import torch

from .checkpoint import load_checkpoint
from .config import NetConfig, default_config
from .init import random_state_dict
from .nets import HipNet
from .utils import load_audio, message_to_tensor, save_audio, tensor_to_message
from .watermark_id import WatermarkID

logger = logging.getLogger(__name__)


class WaveVerify:
    DEFAULT_SAMPLE_RATE: int = 16000
    DEFAULT_WATERMARK_BITS: int = 16

    def __init__(self, checkpoint: Union[str, Path, Mapping] = "base", device: str = "auto") -> None:
        """checkpoint: "base" (the reference's download URL is empty upstream, utils.py:45-52, so
        this fails exactly as it does there), a path to an atomic .pth / checkpoint directory /
        legacy directory, or a mapping {"generator"|"detector"|"locator": state_dict}."""
        try:
            self.device = self._setup_device(device)
            if isinstance(checkpoint, Mapping):
                sds = {k: dict(v) for k, v in checkpoint.items()}
                from .checkpoint import infer_config
                cfgs = {k: infer_config(k, sd) for k, sd in sds.items()}
            else:
                if checkpoint == "base":
                    raise FileNotFoundError(
                        "the pre-trained 'base' checkpoint is not distributed (empty download URL in the "
                        "reference, waveverify/utils.py:45-52); pass a checkpoint path")
                sds, cfgs = load_checkpoint(Path(checkpoint))
            self._build(sds, cfgs)
            self.sample_rate = self.DEFAULT_SAMPLE_RATE
            self.watermark_bits = self.DEFAULT_WATERMARK_BITS
        except Exception as e:
            logger.error(f"Failed to initialize WaveVerify: {str(e)}")
            raise RuntimeError(f"WaveVerify initialization failed: {str(e)}") from e

    @classmethod
    def random_init(cls, seed: int = 0, device: str = "auto",
                    configs: Optional[Dict[str, NetConfig]] = None) -> "WaveVerify":
        """Seeded random weights (benchmarks, tests): no trained checkpoint ships upstream."""
        cfgs = configs or {k: default_config(k) for k in ("generator", "detector", "locator")}
        self = cls.__new__(cls)
        self.device = self._setup_device(device)
        self._build({k: random_state_dict(c, seed) for k, c in cfgs.items()}, cfgs)
        self.sample_rate, self.watermark_bits = cls.DEFAULT_SAMPLE_RATE, cls.DEFAULT_WATERMARK_BITS
        return self

    def _build(self, sds, cfgs) -> None:
        nets = {k: HipNet(cfgs[k], sds[k], self.device) for k in sds}
        self.configs = cfgs
        # NOT in the reference: "f16" routes detect / detect_batch / verify through the detector's f16-operand / f32-accumulate mode
        # (csrc/wv_h16.hip; the same bits on every fixture, ~3x the clips per second).  Opt-in only; the default is the exact path.
        self.detector_precision = "f32"
        # ... and the same switch for embed / embed_batch (`wm` stays within 1e-4 of the exact path's and the reference's) and locate /
        # locate_batch.  `set_precision("f16")` flips all three; `value` of the headline benchmark is always the exact path.
        self.generator_precision = "f32"
        self.locator_precision = "f32"
        # .model.generator / .detector / .locator like the reference's AudioWatermarking
        self.model = SimpleNamespace(generator=nets.get("generator"), detector=nets.get("detector"),
                                     locator=nets.get("locator"))

    def _setup_device(self, device: str) -> torch.device:
        if device == "auto":
            if not torch.cuda.is_available():
                raise RuntimeError("no MI355X visible and waveverify_amd has no CPU path")
            return torch.device("cuda", torch.cuda.current_device())
        return torch.device(device)

    def set_precision(self, precision: str) -> None:
        """NOT in the reference: "f32" (default, exact) or "f16" (the f16-operand / f32-accumulate throughput mode) for all three nets."""
        if precision not in ("f32", "f16"):
            raise ValueError("precision must be 'f32' or 'f16'")
        self.generator_precision = self.detector_precision = self.locator_precision = precision

    def _need(self, name: str) -> HipNet:
        net = getattr(self.model, name)
        if net is None:
            raise RuntimeError(f"checkpoint holds no {name} weights")
        return net

    # ------------------------------------------------------------------ batched tensor API
    @torch.no_grad()
    def embed_batch(self, audio: torch.Tensor, message: torch.Tensor) -> torch.Tensor:
        """audio [B,1,T] (or [B,T]); message [B,16] or [1,16] (0/1) -> watermarked [B,1,T]."""
        return self._need("generator").generator(audio, message, add_input=True, precision=self.generator_precision)

    @torch.no_grad()
    def detect_batch(self, audio: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """-> (bits [B,16] int32, mean_prob [B,16]); bits = time-averaged sigmoid >= 0.5."""
        mp = self._need("detector").detector_mean_prob(audio, precision=self.detector_precision)
        return (mp >= 0.5).to(torch.int32), mp

    @torch.no_grad()
    def locate_batch(self, audio: torch.Tensor) -> torch.Tensor:
        """-> sigmoid(locator logits) [B, T]."""
        return torch.sigmoid(self._need("locator").locator(audio, precision=self.locator_precision)).squeeze(1)

    # ------------------------------------------------------------------ reference file API
    def embed(self, audio_path: Union[str, Path], watermark_id: Union[WatermarkID, str, int],
              output_path: Optional[Union[str, Path]] = None) -> Tuple[np.ndarray, int, WatermarkID]:
        try:
            watermark_id = self._validate_watermark_id(watermark_id)
            audio, _ = load_audio(audio_path, self.sample_rate)
            msg = message_to_tensor(watermark_id.to_bits(), self.watermark_bits)
            wm = self.embed_batch(audio.unsqueeze(0), msg).squeeze(0)          # [1, T]
            if output_path:
                save_audio(wm, output_path, self.sample_rate)
            return wm.cpu().numpy().squeeze(), self.sample_rate, watermark_id
        except Exception as e:
            logger.error(f"Embedding failed: {str(e)}")
            raise RuntimeError(f"Failed to embed watermark: {str(e)}") from e

    def detect(self, audio_path: Union[str, Path]) -> Tuple[WatermarkID, float]:
        try:
            audio, _ = load_audio(audio_path, self.sample_rate)
            _, mp = self.detect_batch(audio.unsqueeze(0))
            confidence = mp.mean().item()              # mean of per-bit mean probabilities (core.py:583)
            detected = WatermarkID.custom(tensor_to_message(mp))
            return detected, confidence
        except Exception as e:
            logger.error(f"Detection failed: {str(e)}")
            raise RuntimeError(f"Failed to detect watermark: {str(e)}") from e

    def locate(self, audio_path: Union[str, Path]) -> np.ndarray:
        try:
            audio, _ = load_audio(audio_path, self.sample_rate)
            mask = self.locate_batch(audio.unsqueeze(0)).squeeze()
            n = audio.shape[-1]
            if mask.dim() == 1 and mask.shape[0] != n:          # core.py:638-644 (never hit: same length)
                mask = torch.nn.functional.interpolate(mask[None, None], size=n, mode="linear",
                                                       align_corners=False).squeeze()
            return mask.cpu().numpy()
        except Exception as e:
            logger.error(f"Localization failed: {str(e)}")
            raise RuntimeError(f"Failed to locate watermark: {str(e)}") from e

    def verify(self, audio_path: Union[str, Path],
               expected_watermark: Union[WatermarkID, str, int]) -> bool:
        try:
            expected = self._validate_watermark_id(expected_watermark)
            detected, _ = self.detect(audio_path)
            return detected == expected
        except Exception as e:
            logger.error(f"Verification failed: {str(e)}")
            raise RuntimeError(f"Failed to verify watermark: {str(e)}") from e

    def _validate_watermark_id(self, watermark_id) -> WatermarkID:
        if not isinstance(watermark_id, WatermarkID):
            try:
                watermark_id = WatermarkID.custom(watermark_id)
            except (ValueError, TypeError) as e:
                raise ValueError(
                    f"Invalid watermark_id: {e}. Use WatermarkID.for_creator(), .for_timestamp(), etc. "
                    f"or provide a 16-bit binary string, int (0-65535), or 2 bytes.")
        return watermark_id
