#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself (build container only).

The reference's pure-torch `modules/` package is imported from /root/reference and
`model/{generator,detector,locator}.py` are loaded by path, with a ~20-line test-side stub of
the un-vendored `audiotools` package.  Weights are the build's own seeded
state dicts (waveverify_amd.init.random_state_dict) pushed into the reference modules in the
*parametrized* layout (`parametrizations.weight.original0/1`), so the weight-norm fold is
covered too.  Only data (inputs, expected outputs, a few strided activation taps) is written;
no reference source travels.

Usage (from repo root, in the build container):  python tests/golden/make_golden.py
"""
from __future__ import annotations

import importlib.util
import logging
import os
import sys
import types
import wave

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"


def _import_reference():
    import torch
    import torch.nn as nn
    sys.dont_write_bytecode = True
    logging.disable(logging.CRITICAL)
    sys.path.insert(0, REF)
    at = types.ModuleType("audiotools")
    ml = types.ModuleType("audiotools.ml")

    class AudioSignal:                                  # test-side stub (not reference code)
        def __init__(self, audio_data, sample_rate=16000):
            self.audio_data = audio_data
            self.sample_rate = sample_rate

        @property
        def device(self):
            return self.audio_data.device

        @property
        def batch_size(self):
            return self.audio_data.shape[0]

        def to(self, d):
            self.audio_data = self.audio_data.to(d)
            return self

    class BaseModel(nn.Module):
        @property
        def device(self):
            return next(self.parameters()).device

    at.AudioSignal = AudioSignal
    ml.BaseModel = BaseModel
    at.ml = ml
    sys.modules["audiotools"] = at
    sys.modules["audiotools.ml"] = ml
    import modules  # noqa: F401  (reference package)

    def load(name):
        spec = importlib.util.spec_from_file_location(f"ref_{name}", f"{REF}/model/{name}.py")
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        return m

    return (torch, AudioSignal, load("generator").Generator, load("detector").Detector,
            load("locator").Locator)


def _ref_kwargs(cfg):
    kw = dict(dimension=cfg.dimension, channels_enc=cfg.channels_enc, n_fft_base=cfg.n_fft_base,
              n_residual_enc=cfg.n_residual_enc, strides=list(cfg.strides),
              kernel_size=cfg.kernel_size, last_kernel_size=cfg.last_kernel_size,
              residual_kernel_size=cfg.residual_kernel_size, dilation_base=cfg.dilation_base,
              zero_init=cfg.zero_init)
    if cfg.kind == "generator":
        kw.update(channels_dec=cfg.channels_dec, n_residual_dec=cfg.n_residual_dec,
                  msg_dimension=cfg.msg_dimension, nbits=cfg.nbits,
                  embedding_dim=cfg.embedding_dim, embedding_layers=cfg.embedding_layers,
                  freq_bands=cfg.freq_bands)
    elif cfg.kind == "detector":
        kw.update(nbits=cfg.nbits, output_dim=cfg.output_dim)
    else:
        kw.update(output_dim=cfg.output_dim)
    return kw


def build_ref(torch, cls, cfg, seed):
    from waveverify_amd.init import random_state_dict
    model = cls(**_ref_kwargs(cfg)).eval()
    sd = random_state_dict(cfg, seed, parametrized=True)
    tsd = {k: torch.from_numpy(v) for k, v in sd.items()}
    missing, unexpected = model.load_state_dict(tsd, strict=False)
    missing = [m for m in missing if not m.endswith("spec.weight")]   # DFT buffers: keep theirs
    assert not missing and not unexpected, (missing[:5], unexpected[:5])
    return model


def tap_hooks(model, names):
    """Capture the outputs of a few named sub-modules."""
    taps = {}
    mods = dict(model.named_modules())
    hs = []
    for label, mname in names.items():
        hs.append(mods[mname].register_forward_hook(
            lambda m, i, o, label=label: taps.__setitem__(label, o.detach().clone())))
    return taps, hs


def read_speech(path, start, n):
    """examples/audios/*.ogg are RIFF/WAV PCM16 mono 16 kHz."""
    with wave.open(path, "rb") as w:
        assert w.getframerate() == 16000 and w.getnchannels() == 1 and w.getsampwidth() == 2
        w.setpos(start)
        pcm = np.frombuffer(w.readframes(n), dtype="<i2")
    return (pcm.astype(np.float32) / 32768.0)


def sub(a, step):
    """Strided sub-sample along time to keep fixtures small."""
    return np.ascontiguousarray(a[..., ::step])


def main():
    from waveverify_amd.config import default_config
    from waveverify_amd.init import synthetic_clips
    torch, AudioSignal, RG, RD, RL = _import_reference()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    out = {}

    # ---------------- case A: full-size nets, B=2, T in {16000, 16001, 4800} --------------
    cg, cd, cl = (default_config(k, zero_init=True) for k in ("generator", "detector", "locator"))
    G, D, L = build_ref(torch, RG, cg, 0), build_ref(torch, RD, cd, 0), build_ref(torch, RL, cl, 0)
    for T in (16000, 16001, 4800):
        x, msg = synthetic_clips(2, T, seed=1234 + T)
        xt, mt = torch.from_numpy(x), torch.from_numpy(msg)
        with torch.no_grad():
            tg, hg = tap_hooks(G, {"conv_pre": "encoder.conv_pre", "enc0": "encoder.downsample.0",
                                   "enc3": "encoder.downsample.3", "latent": "encoder",
                                   "dec_head": "decoder.model.1", "dec_up0": "decoder.model.8"})
            delta = G(AudioSignal(xt.clone()), mt).audio_data
            for h in hg:
                h.remove()
            wm = delta + xt
            dl = D(AudioSignal(wm.clone()))
            ll = L(AudioSignal(wm.clone()))
            dl_clean = D(AudioSignal(xt.clone()))
        mp = torch.sigmoid(dl).mean(dim=2)
        tag = f"full_T{T}"
        np.savez_compressed(
            os.path.join(HERE, f"{tag}.npz"),
            x=x, msg=msg, delta=delta.numpy(), wm=wm.numpy(),
            det_mean_prob=mp.numpy(), det_bits=(mp >= 0.5).int().numpy(),
            det_margin=(mp - 0.5).abs().min().numpy(),
            det_logits_sub=sub(dl.numpy(), 37), det_clean_mean_prob=torch.sigmoid(dl_clean).mean(2).numpy(),
            loc_logits_sub=sub(ll.numpy(), 7),
            conv_pre_sub=sub(tg["conv_pre"].numpy(), 97),
            enc0_sub=sub(tg["enc0"].numpy(), 53), enc3=tg["enc3"].numpy()[:, ::8],
            latent=tg["latent"].numpy(), dec_head_sub=tg["dec_head"].numpy()[:, ::16],
            dec_up0_sub=tg["dec_up0"].numpy()[:, ::8],
            seed=np.int64(0), T=np.int64(T))
        out[tag] = float((mp - 0.5).abs().min())

    # ---------------- case B: real speech, B=1 (config 1 of BASELINE.json) -----------------
    x = np.stack([read_speech(f"{REF}/examples/audios/audio_sample{i}.ogg", 16000, 16000)
                  for i in (1, 2, 3)])[:, None, :]
    msg = np.array([[int(c) for c in format(v, "016b")] for v in (42, 0xBEEF, 0x1234)], np.float32)
    with torch.no_grad():
        xt = torch.from_numpy(x)
        delta = G(AudioSignal(xt.clone()), torch.from_numpy(msg)).audio_data
        wm = delta + xt
        mp = torch.sigmoid(D(AudioSignal(wm.clone()))).mean(dim=2)
        ll = L(AudioSignal(wm.clone()))
    np.savez_compressed(os.path.join(HERE, "speech_T16000.npz"), x=x, msg=msg, wm=wm.numpy(),
                        det_mean_prob=mp.numpy(), det_bits=(mp >= 0.5).int().numpy(),
                        loc_logits_sub=sub(ll.numpy(), 7), seed=np.int64(0))
    out["speech"] = float((mp - 0.5).abs().min())

    # ---------------- case C: shrunk nets, every tensor kept in full -----------------------
    small = dict(channels_enc=8, dimension=16, strides=[2, 2], n_fft_base=16, zero_init=True)
    sg = default_config("generator", channels_dec=8, n_residual_dec=2, **small)
    sdcfg = default_config("detector", output_dim=8, nbits=16, **small)
    slcfg = default_config("locator", **{**small, "channels_enc": 4, "dimension": 8,
                                         "n_residual_enc": 1, "output_dim": 8})
    g, d, l = build_ref(torch, RG, sg, 7), build_ref(torch, RD, sdcfg, 7), build_ref(torch, RL, slcfg, 7)
    for T in (64, 67, 1):
        x, msg = synthetic_clips(3, T, seed=99 + T)
        with torch.no_grad():
            xt = torch.from_numpy(x)
            tg, hg = tap_hooks(g, {"conv_pre": "encoder.conv_pre", "block0": "encoder.blocks.0",
                                   "spec0": "encoder.spec_blocks.0", "down0": "encoder.downsample.0",
                                   "down1": "encoder.downsample.1", "latent": "encoder",
                                   "dec_head": "decoder.model.1", "dec_pw0": "decoder.model.5",
                                   "dec_up0": "decoder.model.7"})
            delta = g(AudioSignal(xt.clone()), torch.from_numpy(msg)).audio_data
            for h in hg:
                h.remove()
            dl = d(AudioSignal(xt.clone()))
            ll = l(AudioSignal(xt.clone()))
        np.savez_compressed(os.path.join(HERE, f"small_T{T}.npz"), x=x, msg=msg,
                            delta=delta.numpy(), det_logits=dl.numpy(), loc_logits=ll.numpy(),
                            **{f"tap_{k}": v.numpy() for k, v in tg.items()}, seed=np.int64(7))

    # ---------------- case D: dilation > 1 on one resblock (API parity, seanet.py:691) -----
    dil = default_config("detector", output_dim=8, dilation_base=2,
                         **{**small, "n_residual_enc": 2})
    dd = build_ref(torch, RD, dil, 11)
    x, _ = synthetic_clips(2, 50, seed=5)
    with torch.no_grad():
        dl = dd(AudioSignal(torch.from_numpy(x)))
    np.savez_compressed(os.path.join(HERE, "small_dilated_T50.npz"), x=x, det_logits=dl.numpy(),
                        seed=np.int64(11))

    # ---------------- DFT bases the reference builds (conv.py:1003-1026) -------------------
    np.savez_compressed(os.path.join(HERE, "dft_basis.npz"),
                        n64=G.state_dict()["encoder.spec_blocks.0.spec.weight"].numpy()[:, 0, :],
                        n1024_rows_every19=G.state_dict()["encoder.spec_post.spec.weight"]
                        .numpy()[::19, 0, :])
    print("golden margins |p-0.5| min:", out)


if __name__ == "__main__":
    main()
