"""The drop-in boundary on the GPU: waveverify_amd.WaveVerify used the way the reference's
examples use waveverify.WaveVerify (examples/basic_usage.py): files in, files out."""
import os

import numpy as np
import pytest
import torch

from waveverify_amd import WatermarkID
from waveverify_amd.config import default_config
from waveverify_amd.init import random_state_dict
from waveverify_amd.utils import load_audio, save_audio

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def wv():
    from waveverify_amd import WaveVerify
    return WaveVerify.random_init(seed=0)


def test_file_roundtrip_matches_reference(golden_dir, wv, tmp_path):
    g = np.load(os.path.join(golden_dir, "speech_T16000.npz"))
    src, out = tmp_path / "clip.wav", tmp_path / "out" / "wm.wav"
    save_audio(torch.from_numpy(g["x"][0]), src, 16000)
    wm, sr, wid = wv.embed(src, 42, out)                         # golden clip 0 carries ID 42
    assert sr == 16000 and isinstance(wid, WatermarkID) and wid.to_int() == 42
    assert wm.shape == (16000,) and wm.dtype == np.float32
    assert np.abs(wm - g["wm"][0, 0]).max() <= 2e-5
    assert out.exists()
    back, _ = load_audio(out)
    assert np.abs(back.numpy()[0] - np.clip(wm, -1, 1)).max() == 0.0
    det, conf = wv.detect(out)
    ref_bits = "".join(str(int(b)) for b in g["det_bits"][0])
    assert det.bits == ref_bits
    assert abs(conf - float(g["det_mean_prob"][0].mean())) < 1e-5
    assert wv.verify(out, det) is True and wv.verify(out, det.to_int() ^ 1) is False
    mask = wv.locate(out)
    assert mask.shape == (16000,) and 0.0 <= mask.min() and mask.max() <= 1.0
    ref = 1.0 / (1.0 + np.exp(-g["loc_logits_sub"][0, 0]))
    assert np.abs(mask[::7] - ref).max() <= 1e-4


def test_batched_api_and_ids(wv):
    x = torch.randn(3, 1, 4000).clamp(-1, 1) * 0.1
    ids = [WatermarkID.for_creator("a"), WatermarkID.custom(0xBEEF), WatermarkID.for_tracking("7")]
    msg = torch.tensor([[int(c) for c in i.bits] for i in ids], dtype=torch.float32)
    wm = wv.embed_batch(x, msg)
    assert wm.shape == (3, 1, 4000) and wm.is_cuda
    bits, mp = wv.detect_batch(wm)
    assert bits.shape == (3, 16) and mp.shape == (3, 16)
    assert torch.equal(bits, (mp >= 0.5).int())
    assert wv.locate_batch(wm).shape == (3, 4000)


def test_error_wrapping(wv, tmp_path):
    with pytest.raises(RuntimeError, match="Failed to embed watermark: Audio file not found"):
        wv.embed(tmp_path / "missing.wav", 1)
    with pytest.raises(RuntimeError, match="Failed to detect watermark"):
        wv.detect(tmp_path / "missing.wav")
    with pytest.raises(RuntimeError, match="Failed to locate watermark"):
        wv.locate(tmp_path / "missing.wav")
    p = tmp_path / "a.wav"
    save_audio(torch.zeros(1, 800), p)
    with pytest.raises(RuntimeError, match="Failed to embed watermark: Invalid watermark_id"):
        wv.embed(p, 70000)
    with pytest.raises(RuntimeError, match="Failed to verify watermark"):
        wv.verify(p, "abc")


def test_checkpoint_roundtrip(tmp_path, wv):
    """An atomic checkpoint in the reference's format loads into the same nets."""
    from waveverify_amd import WaveVerify
    cfgs = {k: default_config(k) for k in ("generator", "detector", "locator")}
    models = {k: {n: torch.from_numpy(v) for n, v in random_state_dict(c, 0, parametrized=(k == "locator")).items()}
              for k, c in cfgs.items()}
    torch.save({"step": 1, "models": models, "config": None}, tmp_path / "best.pth")
    other = WaveVerify(str(tmp_path))
    x = torch.randn(2, 1, 3200) * 0.1
    msg = torch.randint(0, 2, (2, 16)).float()
    assert torch.equal(other.embed_batch(x, msg), wv.embed_batch(x, msg))
    assert torch.equal(other.detect_batch(x)[1], wv.detect_batch(x)[1])
    assert torch.allclose(other.locate_batch(x), wv.locate_batch(x), atol=1e-6)


def test_forward_captures_into_a_hip_graph(wv):
    """The forward passes are plain launches on the caller's stream (no allocation, no sync):
    they capture into a HIP graph and replay bit-identically."""
    x = (torch.randn(2, 1, 3200) * 0.1).clamp(-1, 1).cuda()
    msg = torch.tensor([[1, 0] * 8, [0, 1] * 8], dtype=torch.float32).cuda()
    gen, det = wv.model.generator, wv.model.detector

    def step():
        wm = gen.generator(x, msg, add_input=True)
        return wm, det.detector_mean_prob(wm)

    wm0, p0 = step()                                   # eager (also sizes the workspaces)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):         # captured on `side`, whose workspaces exist already
        wm1, p1 = step()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(wm1, wm0) and torch.equal(p1, p0)
    # A larger eager batch on the capture stream outgrows its workspaces.  The buffers the graph
    # points at must stay alive (retired, not freed) and must not be handed to anything else.
    xb = (torch.randn(16, 1, 6400) * 0.1).clamp(-1, 1).cuda()
    with torch.cuda.stream(side):
        big = gen.generator(xb, msg[:1], add_input=True)
        junk = [torch.full((1 << 20,), float("nan"), device="cuda") for _ in range(8)]   # would reuse freed blocks
    torch.cuda.synchronize()
    assert len(gen._retired) >= 1
    wm1.zero_(); p1.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(wm1, wm0) and torch.equal(p1, p0)
    assert torch.isfinite(big).all() and len(junk) == 8


def test_load_audio_resamples_on_the_gpu(tmp_path, wv):
    """A 44.1 kHz stereo WAV goes through the native reader, the mono mix-down and the GPU polyphase resampler (441 : 160) and comes
    out as the float64 restatement of torchaudio's algorithm predicts (parity with the library itself is unpinned); embed() then runs."""
    from oracle import wv_oracle_fx as OF
    rng = np.random.default_rng(0)
    t = np.arange(44100) / 44100.0
    st = np.stack([0.3 * np.sin(2 * np.pi * 440 * t), 0.2 * np.sin(2 * np.pi * 1000 * t)]).astype(np.float32) + (0.01 * rng.standard_normal((2, 44100))).astype(np.float32)
    src = str(tmp_path / "in44.wav")
    save_audio(torch.from_numpy(st), src, 44100)
    wav, sr = load_audio(src)
    assert sr == 16000 and tuple(wav.shape) == (1, 16000)
    written = np.clip(st, -1, 1).mean(0, keepdims=True)                       # float32 WAV: exact samples
    ref = OF.resample(written, 44100, 16000)
    assert float(np.abs(wav.numpy() - ref).max()) <= 2e-5
    audio, sr2, _ = wv.embed(src, "1010101010101010")
    assert sr2 == 16000 and audio.shape[-1] == 16000
