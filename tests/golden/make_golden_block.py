#!/usr/bin/env python3
"""Generate tests/golden/grads_block_*.npz and tests/golden/bce_losses.npz from the REFERENCE (build container only):

* forward + backward of whole SEANetResnetBlock modules (/root/reference/modules/seanet.py:123-281: identity
  shortcut, pre_scale from idx, res_scale and the trainable res_scale_param of zero_init blocks) through the
  reference's CPU autograd, in float64 (res_scale is always set, as in the shipped configurations: with res_scale=None
  the block's first in-place ELU would overwrite the tensor its identity shortcut aliases);
* LocalizationLoss / DecodingLoss (/root/reference/scripts/loss.py:947-1099) values and input gradients.  loss.py
  imports audiotools at module level (absent, unused by these two classes): loaded by path with an empty stand-in.

Only data is written.  Usage (from repo root):  python tests/golden/make_golden_block.py"""
import importlib.util
import logging
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
SHORT = {"1.conv.conv.parametrizations.weight.original0": "g_pw", "1.conv.conv.parametrizations.weight.original1": "v_pw",
         "2.conv.conv.parametrizations.weight.original0": "g_dw", "2.conv.conv.parametrizations.weight.original1": "v_dw",
         "2.conv.conv.bias": "b_dw"}
SHORT2 = {k.replace("1.", "4.", 1).replace("2.conv", "5.conv"): v for k, v in SHORT.items()}


def main():
    import torch
    sys.dont_write_bytecode = True
    logging.disable(logging.CRITICAL)
    sys.path.insert(0, REF)
    from modules.seanet import SEANetResnetBlock
    torch.set_num_threads(4)
    for tag, (B, C, T, res_scale, idx, zero_init) in {"small": (3, 40, 36, 0.5773503, 1, True), "c64": (2, 64, 200, 0.5, 2, True),
                                                       "c96": (2, 96, 132, 0.7071068, 0, False), "c160": (1, 160, 64, 0.4472136, 3, True)}.items():
        rng = np.random.default_rng(C + T)
        blk = SEANetResnetBlock(C, kernel_size=5, dilations=[1, 1], skip="identity", causal=True, res_scale=res_scale, idx=idx,
                                zero_init=zero_init).double()
        sd = {}
        for k, v in blk.state_dict().items():
            a = rng.standard_normal(tuple(v.shape))
            if k.endswith("original0"):
                a = 0.5 + np.abs(a)
            elif k.endswith("original1"):
                a = a * (1.0 / np.sqrt(np.prod(v.shape[1:])))
            elif k == "res_scale_param":
                a = 0.7 + 0.2 * a
            else:
                a = a * 0.1
            sd[k] = torch.from_numpy(a.astype(np.float32)).double()
        blk.load_state_dict(sd)
        x = torch.from_numpy(rng.standard_normal((B, C, T)).astype(np.float32)).double().requires_grad_(True)
        dy = torch.from_numpy(rng.standard_normal((B, C, T)).astype(np.float32)).double()
        y = blk(x * 1.0)          # (the block's first ELU is in-place: give it a non-leaf tensor)
        y.backward(dy)
        out = dict(x=x.detach().numpy().astype(np.float32), dy=dy.numpy().astype(np.float32), y=y.detach().numpy().astype(np.float32),
                   dx=x.grad.numpy().astype(np.float32), pre_scale=np.float32(blk.pre_scale if blk.pre_scale is not None else 1.0),
                   res_scale=np.float32(1.0 if res_scale is None else res_scale))
        for name, p in blk.named_parameters():
            if name == "res_scale_param":
                out["res_scale_param"], out["d_res_scale_param"] = p.detach().numpy().astype(np.float32), p.grad.numpy().astype(np.float32)
                continue
            key = name[len("block."):]
            half, short = (1, SHORT[key]) if key in SHORT else (2, SHORT2[key])
            out[f"h{half}_{short}"] = p.detach().numpy().astype(np.float32)
            out[f"h{half}_d{short}"] = p.grad.numpy().astype(np.float32)
        path = os.path.join(HERE, f"grads_block_{tag}.npz")
        np.savez_compressed(path, **out)
        print(f"wrote {path}: B={B} C={C} T={T} pre_scale={float(out['pre_scale']):.4f} keys={len(out)}")

    # ---- the encoder's Downsample unit (seanet.py:733-772): Scale -> ELU -> 1x1 (C -> 2C, no bias) -> depth-wise k = 2r, stride r
    from modules.seanet import Scale
    from modules.conv import SConv1d
    for tag, (B, C, T, r) in {"r2": (2, 40, 64, 2), "r4": (2, 64, 200, 4), "r5": (1, 48, 100, 5), "r8": (2, 64, 96, 8)}.items():
        rng = np.random.default_rng(C + T + r)
        sc = 0.7071068
        ds = torch.nn.Sequential(Scale(1, value=sc, learnable=False, inplace=True), torch.nn.ELU(inplace=True),
                                 SConv1d(C, 2 * C, 1, norm="weight_norm", bias=False, nonlinearity="relu"),
                                 SConv1d(2 * C, 2 * C, kernel_size=2 * r, stride=r, groups=2 * C, norm="weight_norm", causal=True,
                                         pad_mode="constant", bias=True)).double()
        sd = {}
        for k, v in ds.state_dict().items():
            a = rng.standard_normal(tuple(v.shape))
            if k.endswith("original0"):
                a = 0.5 + np.abs(a)
            elif k.endswith("original1"):
                a = a * (1.0 / np.sqrt(np.prod(v.shape[1:])))
            elif k.endswith("bias"):
                a = a * 0.1
            else:
                a = v.numpy()                       # the fixed Scale buffer / parameter
            sd[k] = torch.from_numpy(np.asarray(a, dtype=np.float32)).double().reshape(v.shape)
        ds.load_state_dict(sd)
        x = torch.from_numpy(rng.standard_normal((B, C, T)).astype(np.float32)).double().requires_grad_(True)
        y = ds(x * 1.0)
        dy = torch.from_numpy(rng.standard_normal(tuple(y.shape)).astype(np.float32)).double()
        y.backward(dy)
        p = dict(ds.named_parameters())
        out = dict(x=x.detach().numpy().astype(np.float32), dy=dy.numpy().astype(np.float32), y=y.detach().numpy().astype(np.float32),
                   dx=x.grad.numpy().astype(np.float32), pre_scale=np.float32(sc), ratio=np.int32(r))
        for name, short in (("2.conv.conv.parametrizations.weight.original0", "g_pw"), ("2.conv.conv.parametrizations.weight.original1", "v_pw"),
                            ("3.conv.conv.parametrizations.weight.original0", "g_dw"), ("3.conv.conv.parametrizations.weight.original1", "v_dw"),
                            ("3.conv.conv.bias", "b_dw")):
            out[short] = p[name].detach().numpy().astype(np.float32)
            out["d" + short] = p[name].grad.numpy().astype(np.float32)
        np.savez_compressed(os.path.join(HERE, f"grads_down_{tag}.npz"), **out)
        print(f"wrote grads_down_{tag}.npz: B={B} C={C}->{2 * C} T={T}->{y.shape[-1]} r={r}")

    # ---- conv_pre (seanet.py:657-664) and the SpecBlock add (seanet.py:362-511) ------------------------------------------
    from modules.seanet import SpecBlock
    out = {}
    for i, (B, C, T, ks) in enumerate(((2, 64, 400, 5), (3, 40, 37, 7))):
        rng = np.random.default_rng(500 + i)
        wav_std = 0.1122080159
        cp = torch.nn.Sequential(Scale(1, value=1 / wav_std, learnable=False, inplace=False),
                                 SConv1d(1, C, ks, norm="weight_norm", causal=True, pad_mode="constant", bias=True)).double()
        sd = {}
        for k, v in cp.state_dict().items():
            a = rng.standard_normal(tuple(v.shape))
            a = 0.5 + np.abs(a) if k.endswith("original0") else a * (0.45 if k.endswith("original1") else 0.1)
            sd[k] = torch.from_numpy(a.astype(np.float32)).double()
        cp.load_state_dict(sd)
        x = torch.from_numpy((0.1 * rng.standard_normal((B, 1, T))).astype(np.float32)).double().requires_grad_(True)
        y = cp(x)
        dy = torch.from_numpy(rng.standard_normal(tuple(y.shape)).astype(np.float32)).double()
        y.backward(dy)
        p = dict(cp.named_parameters())
        for k, t in dict(x=x, dy=dy, y=y, dx=x.grad, g=p["1.conv.conv.parametrizations.weight.original0"],
                         v=p["1.conv.conv.parametrizations.weight.original1"], b=p["1.conv.conv.bias"],
                         dg=p["1.conv.conv.parametrizations.weight.original0"].grad, dv=p["1.conv.conv.parametrizations.weight.original1"].grad,
                         db=p["1.conv.conv.bias"].grad).items():
            out[f"pre{i}_{k}"] = t.detach().numpy().astype(np.float32)
        out[f"pre{i}_in_scale"] = np.float32(1 / wav_std)
    for i, (B, C, T, n_fft, hop, zero_init) in enumerate(((2, 64, 64, 64, 1, True), (2, 48, 40, 128, 4, True), (1, 40, 24, 64, 2, False))):
        rng = np.random.default_rng(600 + i)
        sb = SpecBlock("stft", "log", n_fft, C, hop, "weight_norm", {}, False, "constant", False, causal=True, mean=-4.3, std=2.8,
                       res_scale=0.5773503, zero_init=zero_init, inout_norm=True).double()
        sd = dict(sb.state_dict())
        for k, v in sd.items():
            if k.endswith("original0"):
                sd[k] = torch.from_numpy((0.5 + np.abs(rng.standard_normal(tuple(v.shape)))).astype(np.float32)).double()
            elif k.endswith("original1"):
                sd[k] = torch.from_numpy((rng.standard_normal(tuple(v.shape)) * v.shape[1] ** -0.5).astype(np.float32)).double()
            elif k == "scale_param":
                sd[k] = torch.tensor([0.8], dtype=torch.float64)
        sb.load_state_dict(sd)
        wav = torch.from_numpy((0.1 * rng.standard_normal((B, 1, T * hop))).astype(np.float32)).double()
        x = torch.from_numpy(rng.standard_normal((B, C, T)).astype(np.float32)).double().requires_grad_(True)
        with torch.no_grad():
            P = sb.spec(wav).clamp_min(1e-5).log_().sub_(sb.mean).div_(sb.std)
        y = sb(x * 1.0, wav)
        dy = torch.from_numpy(rng.standard_normal(tuple(y.shape)).astype(np.float32)).double()
        y.backward(dy)
        p = dict(sb.named_parameters())
        items = dict(x=x, wav=wav, P=P, dy=dy, y=y, dx=x.grad, g=p["layer.conv.conv.parametrizations.weight.original0"],
                     v=p["layer.conv.conv.parametrizations.weight.original1"], dg=p["layer.conv.conv.parametrizations.weight.original0"].grad,
                     dv=p["layer.conv.conv.parametrizations.weight.original1"].grad)
        if zero_init:
            items.update(scale_param=p["scale_param"], d_scale_param=p["scale_param"].grad)
        for k, t in items.items():
            out[f"spec{i}_{k}"] = t.detach().numpy().astype(np.float32)
        out[f"spec{i}_meta"] = np.array([n_fft, hop, 0.5773503, -4.3, 2.8])
    from modules.seanet import L2Norm
    for i, (B, C, D, T, ks) in enumerate(((2, 64, 128, 50, 5), (3, 40, 16, 37, 7))):
        rng = np.random.default_rng(700 + i)
        cpo = torch.nn.Sequential(torch.nn.ELU(inplace=False),
                                  SConv1d(C, C, ks, groups=C, norm="weight_norm", causal=True, pad_mode="constant", bias=False, nonlinearity="relu"),
                                  SConv1d(C, D, 1, norm="weight_norm", bias=True), L2Norm(D, inout_norm=True)).double()
        sd = {}
        for k, v in cpo.state_dict().items():
            a = rng.standard_normal(tuple(v.shape))
            a = 0.5 + np.abs(a) if k.endswith("original0") else a * (np.prod(v.shape[1:]) ** -0.5 if k.endswith("original1") else 1.0)
            sd[k] = torch.from_numpy(a.astype(np.float32)).double()
        cpo.load_state_dict(sd)
        x = torch.from_numpy(rng.standard_normal((B, C, T)).astype(np.float32)).double().requires_grad_(True)
        y = cpo(x)
        dy = torch.from_numpy(rng.standard_normal(tuple(y.shape)).astype(np.float32)).double()
        y.backward(dy)
        p = dict(cpo.named_parameters())
        names = dict(g_dw="1.conv.conv.parametrizations.weight.original0", v_dw="1.conv.conv.parametrizations.weight.original1",
                     g_pw="2.conv.conv.parametrizations.weight.original0", v_pw="2.conv.conv.parametrizations.weight.original1", b="2.conv.conv.bias")
        for k, t in dict(x=x, dy=dy, y=y, dx=x.grad).items():
            out[f"post{i}_{k}"] = t.detach().numpy().astype(np.float32)
        for k, nm in names.items():
            out[f"post{i}_{k}"] = p[nm].detach().numpy().astype(np.float32)
            out[f"post{i}_d{k}"] = p[nm].grad.numpy().astype(np.float32)
    from modules.conv import SConvTranspose1d
    for i, (B, K, T, r, sc) in enumerate(((2, 64, 50, 8, None), (2, 48, 40, 5, 0.7071068), (1, 40, 36, 4, 0.7071068), (2, 64, 32, 2, 0.7071068))):
        rng = np.random.default_rng(800 + i)
        mods = ([] if sc is None else [Scale(1, value=sc, learnable=False, inplace=True)]) + [
            torch.nn.ELU(inplace=True),
            SConvTranspose1d(K, K, kernel_size=2 * r, stride=r, groups=K, norm="weight_norm", causal=True, trim_right_ratio=1.0, bias=False,
                             nonlinearity="relu"),
            SConv1d(K, K // 2, 1, norm="weight_norm", bias=True)]
        up = torch.nn.Sequential(*mods).double()
        sd = {}
        for k, v in up.state_dict().items():
            a = rng.standard_normal(tuple(v.shape))
            if k.endswith("original0"):
                a = 0.5 + np.abs(a)
            elif k.endswith("original1"):
                a = a * np.prod(v.shape[1:]) ** -0.5
            elif k.endswith("bias"):
                a = a * 0.1
            else:
                a = v.numpy()
            sd[k] = torch.from_numpy(np.asarray(a, dtype=np.float32)).double().reshape(v.shape)
        up.load_state_dict(sd)
        x = torch.from_numpy(rng.standard_normal((B, K, T)).astype(np.float32)).double().requires_grad_(True)
        y = up(x * 1.0)
        dy = torch.from_numpy(rng.standard_normal(tuple(y.shape)).astype(np.float32)).double()
        y.backward(dy)
        p = dict(up.named_parameters())
        o = 0 if sc is None else 1
        names = dict(g_ct=f"{o + 1}.convtr.convtr.parametrizations.weight.original0", v_ct=f"{o + 1}.convtr.convtr.parametrizations.weight.original1",
                     g_pw=f"{o + 2}.conv.conv.parametrizations.weight.original0", v_pw=f"{o + 2}.conv.conv.parametrizations.weight.original1",
                     b=f"{o + 2}.conv.conv.bias")
        for k, t in dict(x=x, dy=dy, y=y, dx=x.grad).items():
            out[f"up{i}_{k}"] = t.detach().numpy().astype(np.float32)
        for k, nm in names.items():
            out[f"up{i}_{k}"] = p[nm].detach().numpy().astype(np.float32)
            out[f"up{i}_d{k}"] = p[nm].grad.numpy().astype(np.float32)
        out[f"up{i}_meta"] = np.array([r, 1.0 if sc is None else sc])
    np.savez_compressed(os.path.join(HERE, "grads_pre_spec.npz"), **out)
    print("wrote grads_pre_spec.npz", len(out), "arrays")

    # ---- losses ----------------------------------------------------------------------------------------------------
    sys.modules["audiotools"] = types.ModuleType("audiotools")
    sys.modules["audiotools"].AudioSignal = type("AudioSignal", (), {})
    sys.modules["audiotools"].STFTParams = type("STFTParams", (), {})
    spec = importlib.util.spec_from_file_location("ref_loss", f"{REF}/scripts/loss.py")
    L = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(L)
    rng = np.random.default_rng(7)
    out = {}
    for i, (B, nb, T) in enumerate(((2, 16, 400), (3, 16, 37), (1, 8, 1), (4, 16, 301))):
        z = torch.from_numpy((3.0 * rng.standard_normal((B, nb, T))).astype(np.float32)).requires_grad_(True)
        zl = torch.from_numpy((3.0 * rng.standard_normal((B, 1, T))).astype(np.float32)).requires_grad_(True)
        if i == 0:
            with torch.no_grad():
                z[0, 0, :4] = torch.tensor([60.0, -60.0, 0.0, 1e-4])          # saturated / zero logits
        mask = torch.from_numpy((rng.random((B, 1, T)) < 0.7).astype(np.float32))
        msg = torch.from_numpy(rng.integers(0, 2, (B, nb)).astype(np.float32))
        ld = L.DecodingLoss()(z, mask, msg)
        ld.backward()
        ll = L.LocalizationLoss()(zl, mask)
        ll.backward()
        for k, v in dict(z=z, zl=zl, mask=mask, msg=msg, dec=ld, loc=ll, dz=z.grad, dzl=zl.grad).items():
            out[f"c{i}_{k}"] = v.detach().numpy().astype(np.float32) if v.dim() else np.float64(float(v))
    np.savez_compressed(os.path.join(HERE, "bce_losses.npz"), **out)
    print("wrote bce_losses.npz")


if __name__ == "__main__":
    main()
