"""Audio file I/O and bit <-> tensor helpers of the package API.

Mirrors /root/reference/waveverify/utils.py:170-412 (load_audio, save_audio, message_to_tensor,
tensor_to_message) in names, argument meaning and error behaviour.  The reference delegates file
decoding and resampling to torchaudio, which is not part of this image: RIFF/WAV (PCM 8/16/24/32
and IEEE float32) is read and written natively here; when torchaudio is importable it is used
for every other container.  Resampling runs on the GPU (waveverify_amd/effects.py: torchaudio's
published polyphase-sinc algorithm restated; the library itself is absent, so parity with it is
UNPINNED for non-16 kHz inputs -- feed 16 kHz mono to stay on pinned ground).
"""
from __future__ import annotations

import struct
from pathlib import Path
from typing import List, Tuple, Union

import numpy as np
import torch

DEFAULT_SAMPLE_RATE = 16000
DEFAULT_BITS = 16
DECISION_THRESHOLD = 0.5
AUDIO_CLAMP_MIN, AUDIO_CLAMP_MAX = -1.0, 1.0


def _read_wav(path: Path) -> Tuple[np.ndarray, int]:
    with open(path, "rb") as f:
        data = f.read()
    if len(data) < 12 or data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError("not a RIFF/WAVE file")
    pos, fmt, pcm = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            tag, ch, sr, _, _, bits = struct.unpack("<HHIIHH", body[:16])
            if tag == 0xFFFE and len(body) >= 26:            # WAVE_FORMAT_EXTENSIBLE
                tag = struct.unpack("<H", body[24:26])[0]
            fmt = (tag, ch, sr, bits)
        elif cid == b"data":
            pcm = body
        pos += 8 + size + (size & 1)
    if fmt is None or pcm is None:
        raise ValueError("WAV file lacks fmt/data chunk")
    tag, ch, sr, bits = fmt
    if tag == 3 and bits == 32:
        x = np.frombuffer(pcm, dtype="<f4").astype(np.float32)
    elif tag == 1 and bits == 16:
        x = np.frombuffer(pcm, dtype="<i2").astype(np.float32) / 32768.0
    elif tag == 1 and bits == 32:
        x = np.frombuffer(pcm, dtype="<i4").astype(np.float32) / 2147483648.0
    elif tag == 1 and bits == 8:
        x = (np.frombuffer(pcm, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    elif tag == 1 and bits == 24:
        b = np.frombuffer(pcm[: len(pcm) // 3 * 3], dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        v = np.where(v >= 1 << 23, v - (1 << 24), v)
        x = v.astype(np.float32) / 8388608.0
    else:
        raise ValueError(f"unsupported WAV encoding (format tag {tag}, {bits} bits)")
    n = len(x) // ch
    return x[: n * ch].reshape(n, ch).T.copy(), sr            # [channels, samples]


def load_audio(audio_path: Union[str, Path], target_sr: int = DEFAULT_SAMPLE_RATE
               ) -> Tuple[torch.Tensor, int]:
    """-> (waveform [1, T] float32, sample_rate); mono mix-down, resample to target_sr."""
    audio_path = Path(audio_path)
    if not audio_path.exists():
        raise FileNotFoundError(f"Audio file not found: {audio_path}")
    if not audio_path.is_file():
        raise ValueError(f"Path is not a file: {audio_path}")
    try:
        with open(audio_path, "rb") as f:
            head = f.read(12)
        if head[:4] == b"RIFF" and head[8:12] == b"WAVE":
            wav, sr = _read_wav(audio_path)
            waveform = torch.from_numpy(wav)
        else:
            import torchaudio                                  # optional dependency
            waveform, sr = torchaudio.load(str(audio_path))
    except Exception as e:
        raise RuntimeError(f"Cannot load audio file: {str(e)}")
    if waveform.shape[0] > 1:
        waveform = torch.mean(waveform, dim=0, keepdim=True)
    if sr != target_sr:
        try:
            from .effects import resample_waveform             # torchaudio.transforms.Resample's algorithm on the GPU
            waveform = resample_waveform(waveform.float().cuda(), int(sr), int(target_sr)).cpu()
            sr = target_sr
        except Exception as e:
            raise RuntimeError(f"Cannot resample audio: {str(e)}")
    return waveform.float(), sr


def save_audio(audio: torch.Tensor, path: Union[str, Path], sample_rate: int = DEFAULT_SAMPLE_RATE) -> None:
    """Clamp to [-1, 1] and write (float32 WAV, what torchaudio.save emits for float tensors)."""
    if not isinstance(audio, torch.Tensor):
        raise ValueError(f"Audio must be torch.Tensor, got {type(audio)}")
    if sample_rate <= 0:
        raise ValueError(f"Sample rate must be positive, got {sample_rate}")
    path = Path(path)
    try:
        path.parent.mkdir(parents=True, exist_ok=True)
    except Exception as e:
        raise IOError(f"Cannot create directory: {str(e)}")
    shape = audio.shape
    if audio.dim() == 1:
        audio = audio.unsqueeze(0)
    elif audio.dim() == 3:
        audio = audio.squeeze(0)
    elif audio.dim() != 2:
        raise ValueError(f"Audio must be 1D, 2D, or 3D tensor, got shape {shape}")
    audio = torch.clamp(audio, AUDIO_CLAMP_MIN, AUDIO_CLAMP_MAX)
    try:
        x = audio.detach().cpu().float().numpy()
        ch, n = x.shape
        body = np.ascontiguousarray(x.T).astype("<f4").tobytes()
        fmt = struct.pack("<HHIIHH", 3, ch, sample_rate, sample_rate * ch * 4, ch * 4, 32)
        with open(path, "wb") as f:
            f.write(b"RIFF" + struct.pack("<I", 4 + 8 + len(fmt) + 8 + len(body)) + b"WAVE")
            f.write(b"fmt " + struct.pack("<I", len(fmt)) + fmt)
            f.write(b"data" + struct.pack("<I", len(body)) + body)
    except Exception as e:
        raise RuntimeError(f"Cannot save audio: {str(e)}")


def message_to_tensor(message: Union[str, List[int]], bits: int = DEFAULT_BITS) -> torch.Tensor:
    """'0101...' or [0,1,...] -> float32 tensor [1, bits] (utils.py:290-353)."""
    if bits <= 0:
        raise ValueError(f"Bits must be positive, got {bits}")
    if isinstance(message, str):
        if not all(c in "01" for c in message):
            raise ValueError("Message string must contain only '0' and '1'")
        if len(message) != bits:
            raise ValueError(f"Message must be {bits} bits, got {len(message)}")
        vals = [int(b) for b in message]
    elif isinstance(message, list):
        if not all(isinstance(x, int) and x in [0, 1] for x in message):
            raise ValueError("Message list must contain only 0 and 1")
        if len(message) != bits:
            raise ValueError(f"Message must be {bits} elements, got {len(message)}")
        vals = message
    else:
        raise TypeError(f"Message must be str or list, got {type(message)}")
    return torch.tensor(vals, dtype=torch.float32).unsqueeze(0)


def tensor_to_message(tensor: torch.Tensor, threshold: float = DECISION_THRESHOLD) -> str:
    """probabilities [B,bits,T] | [B,bits] | [bits] -> bit string of batch element 0
    (mean over time, >= threshold; utils.py:356-412)."""
    if not isinstance(tensor, torch.Tensor):
        raise TypeError(f"Expected torch.Tensor, got {type(tensor)}")
    if not 0 <= threshold <= 1:
        raise ValueError(f"Threshold must be between 0 and 1, got {threshold}")
    shape = tensor.shape
    if tensor.dim() == 3:
        tensor = tensor.mean(dim=2)
    if tensor.dim() == 2:
        tensor = tensor[0]
    if tensor.dim() != 1:
        raise ValueError(f"Cannot process tensor with shape {shape}")
    return "".join(str(int(b)) for b in (tensor >= threshold).int().tolist())
