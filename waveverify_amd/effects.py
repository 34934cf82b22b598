"""Sinc-filter and resample effects on the GPU (SURVEY.md section 8f-3 and the resample front-end of 8f-4).

Mirror of the four AudioEffects the reference implements with third-party arithmetic
(/root/reference/utils/effect_augmentation.py:1451-1501 resample, :1684-1870 high / low / band-pass) plus `identity` (:1364), with
the reference's own wrapper behaviour: cutoffs are clamped to [0, nyquist - 1e-5] and handed over as cutoff / NYQUIST, every method
returns (tensor, mask), low / high-pass return the input unchanged when the library would raise, band-pass raises ValueError.

PARITY UNPINNED: `julius` (0.2.7 in the reference's requirements) and `torchaudio` are not in this image and the reference holds no
output of theirs.  The filters are restated from the libraries' published algorithms --
  julius.lowpass.LowPassFilters: half_size = int(zeros / min_cutoff / 2), zeros = 8; filter = 2 c hann(2h+1) sinc(2 c pi t), t = -h..h,
      normalised to sum 1; replicate padding of h samples; cutoff in cycles per SAMPLE, must be <= 0.5; highpass = x - lowpass;
      bandpass = lowpass(high) - lowpass(low), both with the half_size of the lower cutoff;
  torchaudio.functional.resample (sinc_interp_hann, lowpass_filter_width 6, rolloff 0.99): see `resample_kernels`
-- and checked here against an independent float64 restatement (oracle/wv_oracle_fx.py) only.  Note the reference's cutoff / nyquist
convention doubles the cutoff julius sees (3000 Hz at 16 kHz -> 0.375 cycles per sample = 6 kHz), and cutoffs above nyquist / 2 make the
library raise -- both are kept, not corrected.  The convolutions run in csrc/wv_fx.hip; there is no CPU fallback."""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib

DEFAULT_SAMPLE_RATE = 16000
EPSILON = 1e-5


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(t: torch.Tensor) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError("effects run on the GPU (no CPU fallback)")
    return t.float().contiguous()


# ---- taps, as the libraries publish them ----------------------------------------------------------------------------------------
def lowpass_taps(cutoffs, zeros: float = 8) -> Tuple[np.ndarray, int]:
    """julius.lowpass.LowPassFilters.__init__: one windowed-sinc filter per cutoff (cycles per sample), float32 arithmetic as torch's
    defaults give it.  -> (taps [n, 2h+1] float32, h)."""
    cutoffs = [float(c) for c in cutoffs]
    if min(cutoffs) < 0:
        raise ValueError("Minimum cutoff must be larger than zero.")
    if max(cutoffs) > 0.5:
        raise ValueError("A cutoff above 0.5 does not make sense.")
    half = int(zeros / min(c for c in cutoffs if c > 0) / 2)
    window = torch.hann_window(2 * half + 1, periodic=False)
    time = torch.arange(-half, half + 1)
    filters = []
    for c in cutoffs:
        if c == 0:
            f = torch.zeros_like(time).float()
        else:
            xx = 2 * c * math.pi * time
            sinc = torch.where(xx == 0, torch.tensor(1.0), torch.sin(xx) / xx)
            f = 2 * c * window * sinc
            f = f / f.sum()
        filters.append(f)
    return torch.stack(filters).numpy().astype(np.float32), half


def resample_kernels(orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    """torchaudio.functional.functional._get_sinc_resample_kernel (sinc_interp_hann): float64 index arithmetic, float32 kernels.
    -> (kernels [new, 2 width + orig] float32, width, orig, new) with orig / new divided by their gcd."""
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    idx = np.arange(-width, width + orig, dtype=np.float64)[None, :] / orig
    t = np.arange(0, -new, -1, dtype=np.float64)[:, None] / new + idx
    t = np.clip(t * base, -lowpass_filter_width, lowpass_filter_width)
    window = np.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    with np.errstate(invalid="ignore", divide="ignore"):
        k = np.where(t == 0, 1.0, np.sin(t) / t)
    k = k * window * (base / orig)
    return k.astype(np.float32), width, orig, new


# ---- device convolutions -----------------------------------------------------------------------------------------------------------
def _fir(x: torch.Tensor, taps: np.ndarray, half: int) -> torch.Tensor:
    """x [..., T] -> [n_filters, ..., T]: julius' replicate-padded 'same' convolution."""
    lib = _lib.load()
    shape = list(x.shape)
    xr = _dev(x).reshape(-1, shape[-1])
    nf, L = taps.shape
    td = torch.from_numpy(np.ascontiguousarray(taps)).to(xr.device)
    y = torch.empty(xr.shape[0], nf, shape[-1], device=xr.device)
    if lib.wv_fx_fir_bank(xr.data_ptr(), td.data_ptr(), y.data_ptr(), xr.shape[0], shape[-1], nf, L, 1, half, half, 1, 0, _stream()) != 0:
        raise RuntimeError("wv_fx_fir_bank failed")
    return y.permute(1, 0, 2).reshape([nf] + shape)


def lowpass(x: torch.Tensor, cutoff: float) -> torch.Tensor:
    """julius.lowpass_filter(x, cutoff)."""
    taps, half = lowpass_taps([cutoff])
    return _fir(x, taps, half)[0]


def highpass(x: torch.Tensor, cutoff: float) -> torch.Tensor:
    """julius.highpass_filter(x, cutoff) = x - lowpass(x)."""
    return _dev(x) - lowpass(x, cutoff)


def bandpass(x: torch.Tensor, cutoff_low: float, cutoff_high: float) -> torch.Tensor:
    """julius.bandpass_filter: lowpass(high) - lowpass(low), one filter bank (the half width of the lower cutoff)."""
    if cutoff_low > cutoff_high:
        raise ValueError(f"Lower cutoff {cutoff_low} should be less than higher cutoff {cutoff_high}.")
    taps, half = lowpass_taps([cutoff_low, cutoff_high])
    lows = _fir(x, taps, half)
    return lows[1] - lows[0]


def resampled_length(T: int, orig_freq: int, new_freq: int) -> int:
    """Length of `resample_waveform`'s output: ceil(new * T / orig) with the two rates divided by their gcd."""
    g = math.gcd(int(orig_freq), int(new_freq))
    return int(math.ceil((int(new_freq) // g) * T / (int(orig_freq) // g)))


def resample_waveform(x: torch.Tensor, orig_freq: int, new_freq: int) -> torch.Tensor:
    """torchaudio.transforms.Resample(orig_freq, new_freq)(x) on [..., T]."""
    if int(orig_freq) == int(new_freq):
        return _dev(x)
    lib = _lib.load()
    k, width, orig, new = resample_kernels(orig_freq, new_freq)
    shape = list(x.shape)
    xr = _dev(x).reshape(-1, shape[-1])
    T = shape[-1]
    t_out = int(math.ceil(new * T / orig))
    kd = torch.from_numpy(np.ascontiguousarray(k)).to(xr.device)
    y = torch.empty(xr.shape[0], t_out, device=xr.device)
    if lib.wv_fx_resample(xr.data_ptr(), kd.data_ptr(), y.data_ptr(), xr.shape[0], T, orig, new, k.shape[1], width, t_out, _stream()) != 0:
        raise RuntimeError("wv_fx_resample failed")
    return y.reshape(shape[:-1] + [t_out])


# ---- adjoints: the gradient of a loss through these effects -------------------------------------------------------------------------
# In the reference the four effects are plain differentiable torch ops (julius' FFT / direct convolutions, torchaudio's strided conv1d;
# effect_augmentation.py:1451-1501,1684-1870 -- not the straight-through Function classes of :462-500), so the generator's gradient
# crosses them through the TRANSPOSED filter.
def _fir_adjoint(dy: torch.Tensor, taps: np.ndarray, half: int) -> torch.Tensor:
    """Transpose of `_fir` for ONE filter: dy [..., T] -> dx [..., T] (time-reversed taps over the zero-padded gradient, then the
    transpose of the replicate padding)."""
    lib = _lib.load()
    shape = list(dy.shape)
    T = shape[-1]
    dr = _dev(dy).reshape(-1, T)
    L = taps.shape[1]
    rev = torch.from_numpy(np.ascontiguousarray(taps[:1, ::-1])).to(dr.device)
    dxp = torch.empty(dr.shape[0], 1, T + L - 1, device=dr.device)               # gradient towards the replicate-padded signal
    if lib.wv_fx_fir_bank(dr.data_ptr(), rev.data_ptr(), dxp.data_ptr(), dr.shape[0], T, 1, L, 1, L - 1, L - 1, 0, 0, _stream()) != 0:
        raise RuntimeError("wv_fx_fir_bank failed")
    dx = torch.empty_like(dr)
    if lib.wv_fx_fold_replicate(dxp.data_ptr(), dx.data_ptr(), dr.shape[0], T, half, half, _stream()) != 0:
        raise RuntimeError("wv_fx_fold_replicate failed")
    return dx.reshape(shape)


def lowpass_adjoint(dy: torch.Tensor, cutoff: float) -> torch.Tensor:
    taps, half = lowpass_taps([cutoff])
    return _fir_adjoint(dy, taps, half)


def highpass_adjoint(dy: torch.Tensor, cutoff: float) -> torch.Tensor:
    return _dev(dy) - lowpass_adjoint(dy, cutoff)


def bandpass_adjoint(dy: torch.Tensor, cutoff_low: float, cutoff_high: float) -> torch.Tensor:
    taps, half = lowpass_taps([cutoff_low, cutoff_high])
    return _fir_adjoint(dy, taps[1:2], half) - _fir_adjoint(dy, taps[0:1], half)


def resample_waveform_adjoint(dy: torch.Tensor, orig_freq: int, new_freq: int, t_in: int) -> torch.Tensor:
    """Transpose of `resample_waveform(x [..., t_in], orig_freq, new_freq)`: dy [..., t_out] -> dx [..., t_in]."""
    if int(orig_freq) == int(new_freq):
        return _dev(dy)
    lib = _lib.load()
    k, width, orig, new = resample_kernels(orig_freq, new_freq)
    shape = list(dy.shape)
    dr = _dev(dy).reshape(-1, shape[-1])
    kd = torch.from_numpy(np.ascontiguousarray(k)).to(dr.device)
    dx = torch.empty(dr.shape[0], t_in, device=dr.device)
    if lib.wv_fx_resample_adjoint(dr.data_ptr(), kd.data_ptr(), dx.data_ptr(), dr.shape[0], t_in, orig, new, k.shape[1], width, shape[-1], _stream()) != 0:
        raise RuntimeError("wv_fx_resample_adjoint failed")
    return dx.reshape(shape[:-1] + [t_in])


DIFFERENTIABLE = ("identity", "highpass_filter", "lowpass_filter", "bandpass_filter", "resample")


def apply_effect_backward(name: str, params: dict, d_out: torch.Tensor, sample_rate: int = DEFAULT_SAMPLE_RATE) -> torch.Tensor:
    """Gradient towards the input of `apply_effect(name, params, audio)` given the gradient towards its output (same length): the
    transposed operator for the effects the reference differentiates through (`DIFFERENTIABLE`); anything else is one of the
    reference's straight-through effects (effect_augmentation.py:462-500) and passes the gradient unchanged."""
    if name not in DIFFERENTIABLE or name == "identity":
        return d_out
    A = AudioEffects
    T = d_out.shape[-1]
    if name == "lowpass_filter" or name == "highpass_filter":
        c = A._cutoff(params.get("cutoff_freq", 3000 if name == "lowpass_filter" else 500), sample_rate)
        try:
            return (lowpass_adjoint if name == "lowpass_filter" else highpass_adjoint)(d_out, c)
        except ValueError:                                   # the forward passed the input through
            return d_out
    if name == "bandpass_filter":
        nyquist = sample_rate / 2.0
        lo = max(0.0, min(params.get("cutoff_freq_low", 300), nyquist - EPSILON)) / nyquist
        hi = max(0.0, min(params.get("cutoff_freq_high", 8000), nyquist - EPSILON)) / nyquist
        return bandpass_adjoint(d_out, lo, hi)
    new_sr = int(params["new_sample_rate"])                  # resample: down, up, then cropped / zero-padded back to T
    t_mid = resampled_length(T, sample_rate, new_sr)
    t_up = resampled_length(t_mid, new_sr, sample_rate)
    d_up = d_out[..., :t_up] if t_up <= T else torch.nn.functional.pad(d_out, (0, t_up - T))     # transpose of the crop / zero pad
    d_mid = resample_waveform_adjoint(d_up.contiguous(), new_sr, sample_rate, t_mid)
    return resample_waveform_adjoint(d_mid, sample_rate, new_sr, T)


# ---- the reference's effect wrappers -------------------------------------------------------------------------------------------------
class AudioEffects:
    """identity / highpass_filter / lowpass_filter / bandpass_filter / resample with the reference's signatures and conventions
    (effect_augmentation.py:1364-1379,1451-1501,1684-1870)."""

    @staticmethod
    def identity(tensor, mask=None, **kwargs):
        return tensor, mask

    @staticmethod
    def _cutoff(freq: float, sample_rate: int) -> float:
        nyquist = sample_rate / 2
        return max(0.0, min(freq, nyquist - EPSILON)) / nyquist

    @staticmethod
    def highpass_filter(tensor, cutoff_freq: float = 500, sample_rate: int = DEFAULT_SAMPLE_RATE, mask=None, **kwargs):
        try:
            return highpass(tensor, AudioEffects._cutoff(cutoff_freq, sample_rate)), mask
        except ValueError:                                   # the reference catches the library's error and passes the input through
            return tensor, mask

    @staticmethod
    def lowpass_filter(tensor, cutoff_freq: float = 3000, sample_rate: int = DEFAULT_SAMPLE_RATE, mask=None, **kwargs):
        try:
            return lowpass(tensor, AudioEffects._cutoff(cutoff_freq, sample_rate)), mask
        except ValueError:
            return tensor, mask

    @staticmethod
    def bandpass_filter(tensor, cutoff_freq_low: float = 300, cutoff_freq_high: float = 8000, sample_rate: int = DEFAULT_SAMPLE_RATE,
                        mask=None, **kwargs):
        if cutoff_freq_low < 0:
            raise ValueError(f"Low cutoff frequency must be non-negative, got {cutoff_freq_low} Hz")
        if cutoff_freq_high < 0:
            raise ValueError(f"High cutoff frequency must be non-negative, got {cutoff_freq_high} Hz")
        nyquist = sample_rate / 2.0
        lo = max(0.0, min(cutoff_freq_low, nyquist - EPSILON))
        hi = max(0.0, min(cutoff_freq_high, nyquist - EPSILON))
        if lo >= hi:
            raise ValueError(f"Low cutoff {lo} Hz must be less than high cutoff {hi} Hz")
        nl, nh = lo / nyquist, hi / nyquist
        if not (0.0 < nl < 1.0) or not (0.0 < nh < 1.0):
            raise ValueError(f"Normalized cutoffs must be between 0 and 1. Got low: {nl}, high: {nh}")
        return bandpass(tensor, nl, nh), mask                # a cutoff above 0.5 cycles per sample raises ValueError, as in the reference

    @staticmethod
    def resample(tensor, new_sample_rate: int, sample_rate: int = DEFAULT_SAMPLE_RATE, mask=None, **kwargs):
        if not isinstance(new_sample_rate, int) or new_sample_rate <= 0:
            raise ValueError(f"new_sample_rate must be positive int, got {new_sample_rate}")
        down = resample_waveform(tensor, sample_rate, new_sample_rate)
        return resample_waveform(down, new_sample_rate, sample_rate), mask


def apply_effect(name: str, params: dict, audio: torch.Tensor, mask: Optional[torch.Tensor] = None, sample_rate: int = DEFAULT_SAMPLE_RATE):
    """Dispatcher with the (name, params, audio, mask) -> (audio, mask) shape WatermarkTrainer's `apply_effect` hook expects.  Effects
    that change the length (resample rounding) are cropped / zero-padded back to the input length, as the reference's
    AudioProcessor.adjust_audio_length does for its straight-through effects."""
    fn = getattr(AudioEffects, name, None)
    if fn is None or name.startswith("_"):
        raise NotImplementedError(f"effect '{name}' is not available on the GPU path")
    out, mask = fn(audio, sample_rate=sample_rate, mask=mask, **params)
    T = audio.shape[-1]
    if out.shape[-1] > T:
        out = out[..., :T]
    elif out.shape[-1] < T:
        out = torch.nn.functional.pad(out, (0, T - out.shape[-1]))
    return out, mask
