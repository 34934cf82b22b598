"""Design probe (CPU, no GPU): what does f16 STORAGE of the generator's activations / weights cost in wm accuracy, stage by stage?
Runs the torch port of the oracle with roundings inserted where the f16 kernels round (block input window, intermediate u, block
output, 1x1 weights, spectrogram P, downsample / upsample outputs) for a chosen set of stages, and prints max|d wm| against the exact
run.  `split` stages keep 22-bit operands (hi + lo f16 pair), modelled as exact here.

    python tools/sim_f16.py [B] [T]
"""
import sys
import os

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import wv_oracle_torch as OT                      # noqa: E402
from oracle.wv_oracle import decoder_layout                   # noqa: E402
from waveverify_amd.config import default_config              # noqa: E402
from waveverify_amd.init import random_state_dict, synthetic_clips  # noqa: E402

MODE = {}          # stage name -> "f16" | "split" | "exact"


def q(t, stage, bits=None):
    m = MODE.get(stage, "exact")
    if m == "f16":
        return t.half().float()
    if m == "split":                                          # hi + lo * 2^-11 : 22 bits
        hi = t.half().float()
        lo = ((t - hi) * 2048.0).half().float() / 2048.0
        return hi + lo
    return t


def resnet_block(net, pre, x, idx, rs, stage):
    y = q(F.elu(x * (1 + idx * rs ** 2) ** -0.5), stage)
    for n, (pw, dw) in enumerate(((1, 2), (4, 5))):
        y = OT.sconv1d(y, q(net.w(f"{pre}.block.{pw}.conv.conv.weight"), stage), None)
        wd = net.w(f"{pre}.block.{dw}.conv.conv.weight")
        y = OT.sconv1d(y, wd, net.w(f"{pre}.block.{dw}.conv.conv.bias"), groups=wd.shape[0])
        if n == 0:
            y = q(F.elu(y), stage)
    p = net.opt(f"{pre}.res_scale_param")
    scale = rs * (p.reshape(-1)[0] if p is not None else 1.0)
    return q(y * scale + x, stage + ".stream")


def spec_block(net, pre, x, wav, n_fft, hop, mean, std, rs, stage):
    basis = OT._t(OT.dft_basis(n_fft))[:, None, :].to(wav.dtype)
    c = F.conv1d(F.pad(wav, (n_fft - 1, 0)), basis, None, stride=hop)
    Fq = n_fft // 2 + 1
    y = (c[:, :Fq] ** 2 + c[:, Fq:] ** 2).clamp_min(1e-12).sqrt()
    y = q((y.clamp_min(1e-5).log() - mean) / std, stage)
    y = OT.sconv1d(y, q(net.w(f"{pre}.layer.conv.conv.weight"), stage), None)
    p = net.opt(f"{pre}.scale_param")
    return x + y * (rs * (p.reshape(-1)[0] if p is not None else 1.0))


def encoder(net, x, msg):
    cfg = net.cfg
    rs = cfg.res_scale_enc
    wav = x
    h = OT.sconv1d(x * (1.0 / cfg.wav_std), net.w("encoder.conv_pre.1.conv.conv.weight"), net.w("encoder.conv_pre.1.conv.conv.bias"))
    h = q(h, "enc0.stream")
    e = F.linear(msg, net.w("encoder.msg_embedding.0.weight"), net.w("encoder.msg_embedding.0.bias"))
    for i in range(cfg.embedding_layers):
        j = 1 + 2 * i
        e = F.relu(F.linear(e, net.w(f"encoder.msg_embedding.{j}.weight"), net.w(f"encoder.msg_embedding.{j}.bias")))
    film = e
    stride, mult = 1, 1
    for s, r in enumerate(cfg.ratios_enc):
        st = f"enc{s}"
        for j in range(1, cfg.n_residual_enc + 1):
            h = resnet_block(net, f"encoder.blocks.{s}.{j - 1}", h, j, rs, st)
        h = spec_block(net, f"encoder.spec_blocks.{s}", h, wav, mult * cfg.n_fft_base, stride, cfg.spec_means[s], cfg.spec_stds[s], rs, st)
        stride *= r
        h = q(F.elu(h * (1 + cfg.n_residual_enc * rs ** 2) ** -0.5), st)
        wd = net.w(f"encoder.downsample.{s}.3.conv.conv.weight")
        if MODE.get(st, "exact") != "exact":                 # the composed [M][2r][K] weight is what gets rounded
            pw = net.w(f"encoder.downsample.{s}.2.conv.conv.weight")      # [M][K][1]
            wc = q(pw[:, :, 0, None] * wd[:, 0, None, :], st)              # [M][K][2r]
            h = OT.sconv1d(h, wc, net.w(f"encoder.downsample.{s}.3.conv.conv.bias"), stride=r)
        else:
            h = OT.sconv1d(h, net.w(f"encoder.downsample.{s}.2.conv.conv.weight"), None)
            h = OT.sconv1d(h, wd, net.w(f"encoder.downsample.{s}.3.conv.conv.bias"), stride=r, groups=wd.shape[0])
        bw = h.shape[1] // cfg.freq_bands
        bands = []
        for b in range(cfg.freq_bands):
            g = F.linear(film, net.w(f"encoder.film_layers.{s}.{b}.gamma_layer.weight"), net.w(f"encoder.film_layers.{s}.{b}.gamma_layer.bias")).unsqueeze(-1)
            bt = F.linear(film, net.w(f"encoder.film_layers.{s}.{b}.beta_layer.weight"), net.w(f"encoder.film_layers.{s}.{b}.beta_layer.bias")).unsqueeze(-1)
            bands.append(h[:, b * bw:(b + 1) * bw] * g + bt)
        h = q(torch.cat(bands, dim=1), f"enc{s + 1}.stream" if s + 1 < len(cfg.ratios_enc) else "encpost.stream")
        mult *= 2
    h = spec_block(net, "encoder.spec_post", h, wav, mult * cfg.n_fft_base, stride, cfg.spec_means[-1], cfg.spec_stds[-1], rs, "encpost")
    wd = net.w("encoder.conv_post.1.conv.conv.weight")
    h = OT.sconv1d(q(F.elu(h), "encpost"), wd, None, groups=wd.shape[0])
    h = OT.sconv1d(h, q(net.w("encoder.conv_post.2.conv.conv.weight"), "encpost"), net.w("encoder.conv_post.2.conv.conv.bias"))
    return F.normalize(h, p=2.0, dim=1, eps=1e-12) * (h.shape[1] ** 0.5)


def decoder(net, z):
    cfg = net.cfg
    rs = cfg.res_scale_dec
    i0, i1, ups, il = decoder_layout(cfg)
    h = OT.sconv1d(q(z, "dechead"), q(net.w(f"decoder.model.{i0}.conv.conv.weight"), "dechead"), None)
    wd = net.w(f"decoder.model.{i1}.conv.conv.weight")
    h = OT.sconv1d(h, wd, net.w(f"decoder.model.{i1}.conv.conv.bias"), groups=wd.shape[0])
    post = (1 + cfg.n_residual_dec * rs ** 2) ** -0.5
    for i, (ct, pw, res, r, C) in enumerate(ups):
        st = f"dec{i}"
        h = q(F.elu(h * post if i > 0 else h), st + ".up")
        h = OT.sconvtr1d(h, net.w(f"decoder.model.{ct}.convtr.convtr.weight"), r)
        if MODE.get(st + ".up", "exact") == "f16":
            h = q(h, st + ".up")
        h = OT.sconv1d(h, q(net.w(f"decoder.model.{pw}.conv.conv.weight"), st + ".up"), net.w(f"decoder.model.{pw}.conv.conv.bias"))
        h = q(h, st + ".stream")
        for j, ri in enumerate(res):
            h = resnet_block(net, f"decoder.model.{ri}", h, j, rs, st)
    h = OT.sconv1d(q(F.elu(h * post), "tail"), net.w(f"decoder.model.{il}.conv.conv.weight"), net.w(f"decoder.model.{il}.conv.conv.bias"))
    return torch.tanh(h * cfg.wav_std)


@torch.no_grad()
def embed(net, x, msg):
    return decoder(net, encoder(net, x, msg))[..., : x.shape[-1]] + x


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 16000
    cfg = default_config("generator")
    sd = random_state_dict(cfg, 0)
    x, msg = synthetic_clips(B, T, seed=3)
    net = OT.Net(cfg, sd)
    xt, mt = torch.from_numpy(x), torch.from_numpy(msg)
    MODE.clear()
    ref = embed(net, xt, mt)
    print("delta amplitude: max %.4f rms %.4f" % (float((ref - xt).abs().max()), float((ref - xt).pow(2).mean().sqrt())))
    stages = [f"enc{s}" for s in range(4)] + ["encpost", "dechead"] + [f"dec{i}" for i in range(4)] + ["tail"]
    all_keys = []
    for s in stages:
        all_keys += [s, s + ".stream", s + ".up"]

    def run(label, mode):
        MODE.clear()
        MODE.update(mode)
        out = embed(net, xt, mt)
        d = (out - ref).abs()
        print(f"{label:58s} max|d| {float(d.max()):.3e}  rms {float(d.pow(2).mean().sqrt()):.3e}", flush=True)

    run("everything f16", {k: "f16" for k in all_keys})
    run("everything split (22-bit)", {k: "split" for k in all_keys})
    for s in stages:
        run(f"only {s} f16 (incl. its stream / up)", {k: "f16" for k in (s, s + ".stream", s + ".up")})
    for s in stages:
        run(f"only {s} operands f16, stream exact", {k: "f16" for k in (s, s + ".up")})
    run("operands f16 everywhere, streams split", {**{k: "f16" for k in all_keys}, **{k: "split" for k in all_keys if k.endswith(".stream")}})
    run("encoder all f16, decoder exact", {k: "f16" for k in all_keys if k.startswith("enc")})
    run("encoder all f16 + dechead + dec0, rest exact", {k: "f16" for k in all_keys if k.startswith("enc") or k.startswith("dechead") or k.startswith("dec0")})
    run("encoder + dechead + dec0 + dec1 f16", {k: "f16" for k in all_keys if k.startswith("enc") or k.startswith("dechead") or k.startswith("dec0") or k.startswith("dec1")})


if __name__ == "__main__":
    main()
