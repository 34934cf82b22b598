"""BER and MIoU, the two metrics parity is reported in.

Restated from /root/reference/scripts/evaluate.py: BER.forward :442-516 (mask-weighted
time-averaged sigmoid, >= threshold, errors over valid bits) and MIOU.forward :591-665
(mean of foreground and background IoU on binary masks; an empty union counts as IoU 1).
Note model/watermarking.py:717,797 binarises the *raw* locator output at 0.5 before MIOU.
"""
from __future__ import annotations

from typing import Optional, Union

import numpy as np
import torch


class BER:
    def __init__(self, threshold: float = 0.5, eps: float = 1e-8):
        self.threshold, self.eps = threshold, eps

    def __call__(self, decoded_logits: torch.Tensor, original_bits: torch.Tensor,
                 presence_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        B, W, T = decoded_logits.shape
        if tuple(original_bits.shape) != (B, W):
            raise RuntimeError("BER computation failed")        # reference wraps the ValueError
        probs = torch.sigmoid(decoded_logits)
        if presence_mask is not None:
            if presence_mask.shape[0] != B or presence_mask.shape[2] != T:
                raise RuntimeError("BER computation failed")
            mask = presence_mask.expand(-1, W, -1)
            valid = mask.sum(dim=2) > 0
            avg = (probs * mask).sum(dim=2) / (mask.sum(dim=2) + self.eps)
        else:
            avg = probs.mean(dim=2)
            valid = torch.ones((B, W), dtype=torch.bool, device=decoded_logits.device)
        decoded = (avg >= self.threshold).float()
        errors = ((decoded != original_bits.float()) * valid).sum()
        total = valid.sum()
        if total > 0:
            return errors / total
        return torch.tensor(0.0, device=decoded_logits.device)


class MIOU:
    def __call__(self, predicted_mask: Union[torch.Tensor, np.ndarray],
                 ground_truth_mask: Union[torch.Tensor, np.ndarray]) -> float:
        p = predicted_mask.detach().cpu().numpy() if torch.is_tensor(predicted_mask) else np.asarray(predicted_mask)
        g = ground_truth_mask.detach().cpu().numpy() if torch.is_tensor(ground_truth_mask) else np.asarray(ground_truth_mask)
        if p.shape != g.shape or not np.isin(np.unique(p), [0, 1]).all() or not np.isin(np.unique(g), [0, 1]).all():
            raise RuntimeError("MIOU computation failed")
        out = []
        for cls in (1, 0):
            inter = np.logical_and(p == cls, g == cls).sum()
            union = np.logical_or(p == cls, g == cls).sum()
            out.append((1.0 if inter == 0 else 0.0) if union == 0 else inter / union)
        return float(sum(out) / 2)
