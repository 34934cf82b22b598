"""Temporal augmentations of the training step, on the GPU.

Host-side mirror of the reference's two modules and of the place that chains them:

    LocalizationAugmentation   /root/reference/utils/localization_augmentation.py:64-321
    SequenceAugmentation       /root/reference/utils/seq_augmentation.py:42-273
    TemporalAugmenter.forward  /root/reference/model/watermarking.py:487-519  (_apply_augmentations)

Same constructor arguments, `forward` arguments, return tuples, statistics and exceptions.  What differs is where
the work happens: the reference mutates three tensors segment by segment in Python; here the host only DRAWS the
plan -- with the reference's random-number calls in the reference's order (numpy's global generator, and torch's
for the shuffle permutation), so that the same seeds give the same augmentation -- and one HIP launch applies it
(csrc/wv_aug.hip through the C ABI, include/waveverify_hip.h).  There is no CPU fallback.

The reference wraps the augmented audio in audiotools' AudioSignal; `Signal` below carries the two attributes
its callers read (`audio_data`, `sample_rate`)."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import _lib

ORIGINAL_REVERT_PROB = 0.33          # localization_augmentation.py:35-37
ZERO_REPLACE_PROB = 0.66
TARGET_AUGMENTATION_RATIO = 0.20
REVERSE_PROBABILITY = 0.3            # seq_augmentation.py:29-35
CIRCULAR_SHIFT_PROBABILITY = 0.4
SHUFFLE_PROBABILITY = 0.3
DEFAULT_SEGMENT_DURATION = 0.5
DEFAULT_CHUNK_DIVISIONS = 4

SEQ_IDENTITY, SEQ_REVERSE, SEQ_ROLL, SEQ_PERMUTE, SEQ_CHUNK_SWAP = range(5)
KEEP, REVERT, ZERO, CROSS = 0, 1, 2, 3      # plan codes; CROSS + j = take clip j's original


@dataclass
class Signal:
    audio_data: torch.Tensor
    sample_rate: int


@dataclass
class SeqMap:
    """out[t] = in[src(t)] on the time axis; `t_out` is the output length."""
    mode: int = SEQ_IDENTITY
    a: int = 0
    b: int = 0
    c: int = 0
    perm: Optional[np.ndarray] = None
    t_out: int = 0


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(t: torch.Tensor) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError("augmentation tensors must live on the GPU (no CPU fallback)")
    return t.float().contiguous()


def _launch(original, watermarked, plan, seg_len, sm: SeqMap):
    lib = _lib.load()
    B, Cc, T = watermarked.shape
    outs = [torch.empty(B, Cc, sm.t_out, dtype=torch.float32, device=watermarked.device) for _ in range(3)]
    plan_d = perm_d = None
    nseg = 0
    if plan is not None:
        plan = np.ascontiguousarray(plan, dtype=np.int32)
        nseg = plan.shape[1]
        if plan.shape[0] != B or plan.min() < 0 or plan.max() >= CROSS + B:
            raise ValueError("augmentation plan out of range")
        plan_d = torch.from_numpy(plan).to(watermarked.device)
    if sm.mode == SEQ_PERMUTE:
        perm = np.ascontiguousarray(sm.perm, dtype=np.int32)
        if perm.size * sm.a != sm.t_out or perm.min() < 0 or (int(perm.max()) + 1) * sm.a > T:
            raise ValueError("segment permutation out of range")
        perm_d = torch.from_numpy(perm).to(watermarked.device)
    rc = lib.wv_aug_localize_sequence(
        original.data_ptr(), watermarked.data_ptr(), plan_d.data_ptr() if plan_d is not None else None, nseg, int(seg_len),
        sm.mode, sm.a, sm.b, sm.c, perm_d.data_ptr() if perm_d is not None else None,
        outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(), B, Cc, T, sm.t_out, _stream())
    if rc != 0:
        raise RuntimeError(f"wv_aug_localize_sequence failed ({rc})")
    return outs


def inverse_map(sm: SeqMap, t_in: int) -> SeqMap:
    """The map that undoes `sm` on the output axis (out[t] = in[src(t)]  <=>  t = inverse.src(ts))."""
    if sm.mode == SEQ_ROLL:
        return SeqMap(SEQ_ROLL, a=t_in - sm.a, t_out=sm.t_out)
    if sm.mode == SEQ_PERMUTE:
        inv = np.empty_like(np.asarray(sm.perm))
        inv[np.asarray(sm.perm)] = np.arange(len(sm.perm), dtype=inv.dtype)
        return SeqMap(SEQ_PERMUTE, a=sm.a, perm=inv, t_out=sm.t_out)
    return SeqMap(sm.mode, sm.a, sm.b, sm.c, None, sm.t_out)          # identity, reverse, chunk swap: self-inverse


def backward_to_watermarked(d_out: torch.Tensor, plan, seg_len: int, sm: SeqMap, t_in: int) -> torch.Tensor:
    """Gradient of the augmented audio towards the watermarked input of `_launch` (the select passes it where the sample was kept)."""
    lib = _lib.load()
    d_out = _dev(d_out)
    B, Cc, T_out = d_out.shape
    inv = inverse_map(sm, t_in)
    plan_d = perm_d = None
    nseg = 0
    if plan is not None:
        plan = np.ascontiguousarray(plan, dtype=np.int32)
        nseg = plan.shape[1]
        plan_d = torch.from_numpy(plan).to(d_out.device)
    if inv.mode == SEQ_PERMUTE:
        perm_d = torch.from_numpy(np.ascontiguousarray(inv.perm, dtype=np.int32)).to(d_out.device)
    d_wm = torch.empty(B, Cc, t_in, dtype=torch.float32, device=d_out.device)
    rc = lib.wv_aug_backward(d_out.data_ptr(), plan_d.data_ptr() if plan_d is not None else None, nseg, int(seg_len), inv.mode, inv.a, inv.b, inv.c,
                             perm_d.data_ptr() if perm_d is not None else None, d_wm.data_ptr(), B, Cc, t_in, T_out, _stream())
    if rc != 0:
        raise RuntimeError(f"wv_aug_backward failed ({rc})")
    return d_wm


def apply_sequence_map(tensors: List[Optional[torch.Tensor]], sm: SeqMap) -> List[Optional[torch.Tensor]]:
    """The sequence map alone on up to three [B,C,T] tensors (one launch)."""
    lib = _lib.load()
    ref = next(t for t in tensors if t is not None)
    B, Cc, T = ref.shape
    ins = [None if t is None else _dev(t) for t in tensors] + [None] * (3 - len(tensors))
    outs = [None if t is None else torch.empty(B, Cc, sm.t_out, dtype=torch.float32, device=ref.device) for t in ins]
    perm_d = None
    if sm.mode == SEQ_PERMUTE:
        perm = np.ascontiguousarray(sm.perm, dtype=np.int32)
        if perm.size * sm.a != sm.t_out or perm.min() < 0 or (int(perm.max()) + 1) * sm.a > T:
            raise ValueError("segment permutation out of range")
        perm_d = torch.from_numpy(perm).to(ref.device)
    ptr = lambda t: None if t is None else t.data_ptr()          # noqa: E731
    rc = lib.wv_aug_sequence(ptr(ins[0]), ptr(ins[1]), ptr(ins[2]), ptr(outs[0]), ptr(outs[1]), ptr(outs[2]),
                             sm.mode, sm.a, sm.b, sm.c, ptr(perm_d), B * Cc, T, sm.t_out, _stream())
    if rc != 0:
        raise RuntimeError(f"wv_aug_sequence failed ({rc})")
    return outs[:len(tensors)]


class LocalizationAugmentation:
    """Reverts / zeroes / cross-substitutes about 20 % of the fixed-length segments of every clip and returns the
    presence mask (localization_augmentation.py:64-321)."""

    def __init__(self, sample_rate: int = 16000, window_duration: float = 0.1):
        if sample_rate <= 0:
            raise ValueError(f"Sample rate must be positive, got {sample_rate}")
        if window_duration <= 0:
            raise ValueError(f"Window duration must be positive, got {window_duration}")
        self.sample_rate = sample_rate
        self.window_duration = window_duration
        self.segment_length = int(sample_rate * window_duration)
        self._reset_stats()

    def _reset_stats(self) -> None:
        self.stats = {"original_revert": 0, "zero_replace": 0, "cross_substitute": 0, "unchanged": 0}

    def draw_plan(self, batch_size: int, num_samples: int) -> np.ndarray:
        """plan[B][nseg] with the reference's draws in the reference's order (:269-303): per clip one
        choice-without-replacement of segment starts, then per chosen segment one uniform and, for a
        cross-substitution, one choice of the other clip.  Updates `self.stats` (sample counts)."""
        L = self.segment_length
        total_segments = int(np.ceil(num_samples / L))
        segments_to_modify = int(total_segments * TARGET_AUGMENTATION_RATIO)
        plan = np.zeros((batch_size, total_segments), np.int32)
        self._reset_stats()
        for b in range(batch_size):
            starts = np.random.choice(np.arange(0, num_samples, L), segments_to_modify, replace=False)
            for start in starts:
                n = min(start + L, num_samples) - start
                u = np.random.rand()
                if u < ORIGINAL_REVERT_PROB:
                    plan[b, start // L] = REVERT
                    self.stats["original_revert"] += n
                elif u < ZERO_REPLACE_PROB:
                    plan[b, start // L] = ZERO
                    self.stats["zero_replace"] += n
                elif batch_size >= 2:
                    other = np.random.choice([j for j in range(batch_size) if j != b])
                    plan[b, start // L] = CROSS + int(other)
                    self.stats["cross_substitute"] += n
        return plan

    def _finish_stats(self, batch_size: int, num_samples: int) -> Dict[str, float]:
        total = batch_size * num_samples
        self.stats["unchanged"] = total - (self.stats["original_revert"] + self.stats["zero_replace"] +
                                           self.stats["cross_substitute"])
        for k in self.stats:
            self.stats[k] = float((self.stats[k] / total) * 100)
        return self.stats

    def forward(self, original: torch.Tensor, watermarked: torch.Tensor):
        if original.shape != watermarked.shape:
            raise ValueError(f"Shape mismatch: original {original.shape} != watermarked {watermarked.shape}")
        original, watermarked = _dev(original), _dev(watermarked)
        B, _, T = watermarked.shape
        plan = self.draw_plan(B, T)
        wm, upd, mask = _launch(original, watermarked, plan, self.segment_length, SeqMap(t_out=T))
        return Signal(wm, self.sample_rate), mask, upd, self._finish_stats(B, T)

    __call__ = forward


class SequenceAugmentation:
    """One sequence-level transform for the whole batch: reverse (30 %), circular shift (40 %) or a permutation of
    0.5 s segments (30 %), applied alike to the audio, the original and the mask (seq_augmentation.py:42-273)."""

    VALID = ["reverse", "circular_shift", "shuffle", "chunk_shuffle"]

    def __init__(self, sample_rate: int, methods: Optional[List[str]] = None):
        if sample_rate <= 0:
            raise ValueError(f"Sample rate must be positive, got {sample_rate}")
        self.sample_rate = sample_rate
        if methods is None:
            self.methods = list(self.VALID)
        else:
            invalid = set(methods) - set(self.VALID)
            if invalid:
                raise ValueError(f"Invalid augmentation methods: {invalid}. Valid methods: {self.VALID}")
            self.methods = methods
        self.stats = {m: 0 for m in self.methods}
        self.stats["unchanged"] = 0

    def draw(self, batch_size: int, num_samples: int) -> Tuple[str, SeqMap]:
        """The method and its index map, with the reference's draws (:153-206).  As in the reference the selection
        ignores `methods` (they only name the statistics) and 'chunk_shuffle' is never drawn; a clip too short
        for two segments counts as 'shuffle' in the statistics but is returned unchanged."""
        self.stats = {k: 0 for k in self.stats}
        self.stats["unchanged"] = 0
        u = np.random.rand()
        if u < REVERSE_PROBABILITY:
            self.stats["reverse"] += batch_size
            return "reverse", SeqMap(SEQ_REVERSE, t_out=num_samples)
        if u < REVERSE_PROBABILITY + CIRCULAR_SHIFT_PROBABILITY:
            shift = int(np.random.randint(1, num_samples))
            self.stats["circular_shift"] += batch_size
            return "circular_shift", SeqMap(SEQ_ROLL, a=shift, t_out=num_samples)
        if u < REVERSE_PROBABILITY + CIRCULAR_SHIFT_PROBABILITY + SHUFFLE_PROBABILITY:
            seg = int(DEFAULT_SEGMENT_DURATION * self.sample_rate)
            if num_samples >= 2 * seg:
                n = num_samples // seg
                perm = torch.randperm(n).numpy()
                self.stats["shuffle"] += batch_size
                return "shuffle", SeqMap(SEQ_PERMUTE, a=seg, perm=perm, t_out=n * seg)
            self.stats["shuffle"] += batch_size
            return "unchanged", SeqMap(t_out=num_samples)
        self.stats["unchanged"] += batch_size
        return "unchanged", SeqMap(t_out=num_samples)

    def chunk_swap_map(self, num_samples: int) -> Optional[SeqMap]:
        """The 'chunk_shuffle' branch (:212-247), which the reference's own selection never reaches: two
        non-overlapping chunks of T/4 samples exchanged.  None when no placement is found."""
        n = num_samples // DEFAULT_CHUNK_DIVISIONS
        if not (n > 0 and num_samples > 2 * n):
            return None
        c1 = int(np.random.randint(0, num_samples - n))
        c2 = int(np.random.randint(0, num_samples - n))
        attempts = 0
        while abs(c1 - c2) < n and attempts < 100:
            c2 = int(np.random.randint(0, num_samples - n))
            attempts += 1
        return SeqMap(SEQ_CHUNK_SWAP, a=c1, b=c2, c=n, t_out=num_samples) if attempts < 100 else None

    def _finish_stats(self, batch_size: int) -> Dict[str, float]:
        for k in self.stats:
            self.stats[k] = float((self.stats[k] / batch_size) * 100)
        return self.stats

    def forward(self, updated_original: torch.Tensor, watermarked: torch.Tensor, ground_truth_presence: torch.Tensor):
        try:
            if not (updated_original.shape == watermarked.shape == ground_truth_presence.shape):
                raise ValueError(
                    f"Input tensors must have the same shape. Got: updated_original={updated_original.shape}, "
                    f"watermarked={watermarked.shape}, ground_truth_presence={ground_truth_presence.shape}")
            B, _, T = watermarked.shape
            method, sm = self.draw(B, T)
            wm, upd, gt = apply_sequence_map([watermarked, updated_original, ground_truth_presence], sm)
            return Signal(wm, self.sample_rate), upd, gt, self._finish_stats(B), method
        except Exception as e:
            raise RuntimeError(f"Failed to apply augmentation: {str(e)}") from e

    __call__ = forward


class TemporalAugmenter:
    """AudioWatermarking._apply_augmentations (model/watermarking.py:487-519): the localisation augmentation, then
    the sequence augmentation, as ONE launch over the batch (the plan lookup composed with the index map)."""

    def __init__(self, sample_rate: int = 16000, window_duration: float = 0.1):
        self.localization_augmenter = LocalizationAugmentation(sample_rate, window_duration)
        self.seq_augmenter = SequenceAugmentation(sample_rate)
        self.sample_rate = sample_rate

    def forward(self, original: torch.Tensor, watermarked: torch.Tensor):
        if original.shape != watermarked.shape:
            raise ValueError(f"Shape mismatch: original {original.shape} != watermarked {watermarked.shape}")
        original, watermarked = _dev(original), _dev(watermarked)
        B, _, T = watermarked.shape
        loc, seq = self.localization_augmenter, self.seq_augmenter
        plan = loc.draw_plan(B, T)
        stats_loc = dict(loc._finish_stats(B, T))
        try:
            _, sm = seq.draw(B, T)
        except Exception as e:
            raise RuntimeError(f"Failed to apply augmentation: {str(e)}") from e
        stats_seq = dict(seq._finish_stats(B))
        wm, upd, mask = _launch(original, watermarked, plan, loc.segment_length, sm)
        self.last = (plan, loc.segment_length, sm, T)                # what backward() needs
        return Signal(wm, self.sample_rate), mask, upd, {**stats_loc, **stats_seq}

    def backward(self, d_augmented: torch.Tensor) -> torch.Tensor:
        """dL/d(watermarked) from dL/d(augmented watermarked) of the last forward."""
        plan, seg_len, sm, T = self.last
        return backward_to_watermarked(d_augmented, plan, seg_len, sm, T)

    __call__ = forward
