#!/usr/bin/env python3
"""Generate tests/golden/netgrads_{locator,detector}.npz: loss and parameter gradients of WHOLE (shrunk) reference Locator /
Detector modules under the reference's own LocalizationLoss / DecodingLoss, through the reference's CPU autograd in
float64 (build container only).  They pin oracle/wv_oracle_train_torch.py, the gradient oracle of the training step.
Only data is written (inputs, seeds of the parameters, loss, logits, gradients keyed by state-dict name).

Usage (from repo root):  python tests/golden/make_golden_netgrads.py"""
import importlib.util
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import make_golden as MG                      # noqa: E402  (the reference loader + its audiotools stand-in)

CASES = {"locator": dict(kind="locator", channels_enc=8, dimension=16, strides=[4, 2], n_fft_base=16, output_dim=8, seed=3, B=2, T=160),
         "detector": dict(kind="detector", channels_enc=8, dimension=16, strides=[2, 2, 2], n_fft_base=16, output_dim=8, nbits=4,
                          n_residual_enc=2, seed=5, B=2, T=96),
         "generator": dict(kind="generator", channels_enc=8, channels_dec=8, dimension=16, strides=[2, 2], n_fft_base=16, n_residual_dec=2,
                           embedding_dim=16, seed=7, B=3, T=64)}


def main():
    from waveverify_amd.config import default_config
    from waveverify_amd.init import random_state_dict
    torch, _, Generator, Detector, Locator = MG._import_reference()
    sys.modules["audiotools"].STFTParams = type("STFTParams", (), {})
    spec = importlib.util.spec_from_file_location("ref_loss", f"{MG.REF}/scripts/loss.py")
    L = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(L)
    for name, c in CASES.items():
        c = dict(c)
        seed, B, T, kind = c.pop("seed"), c.pop("B"), c.pop("T"), c.pop("kind")
        cfg = default_config(kind, **c)
        model = MG.build_ref(torch, {"detector": Detector, "locator": Locator, "generator": Generator}[kind], cfg, seed).train()
        # float64 where the reference allows it; its encoder casts the message to float32 (seanet.py:909), so the generator runs in float32
        model = model.double() if kind != "generator" else model.float()
        rng = np.random.default_rng(seed)
        x = (0.1 * rng.standard_normal((B, 1, T))).astype(np.float32)
        if kind == "generator":
            # wm = G(x, msg)[..., :T] + x (generator.py:396-413, watermarking.py:423-441) under a plain squared error: it only seeds dL/d(wm)
            msg = rng.integers(0, 2, (B, cfg.nbits)).astype(np.float32)
            target = (x + 0.01 * rng.standard_normal((B, 1, T))).astype(np.float32)
            xt = torch.from_numpy(x).requires_grad_(True)
            wm = model.decode(model.encode(xt, torch.from_numpy(msg)))[..., :T] + xt
            loss = ((wm - torch.from_numpy(target)) ** 2).mean()
            loss.backward()
            out = dict(x=x, msg=msg, target=target, loss=np.float64(float(loss.detach())), wm=wm.detach().numpy().astype(np.float32),
                       dx=xt.grad.numpy().astype(np.float32), cfg=np.array([repr(dict(kind=kind, seed=seed, **c))]))
            n = 0
            for k, p in model.named_parameters():
                if p.grad is not None:
                    out["g:" + k] = p.grad.numpy().astype(np.float64)
                    n += 1
            np.savez_compressed(os.path.join(HERE, f"netgrads_{name}.npz"), **out)
            print(f"wrote netgrads_{name}.npz: loss {float(loss):.6e}, {n} gradient tensors, wm {tuple(wm.shape)}")
            continue
        mask = (rng.random((B, 1, T)) < 0.7).astype(np.float32)
        xt = torch.from_numpy(x).double().requires_grad_(True)
        logits = model.decode(xt, T)
        if kind == "detector":
            msg = rng.integers(0, 2, (B, cfg.nbits)).astype(np.float32)
            loss = L.DecodingLoss()(logits, torch.from_numpy(mask).double(), torch.from_numpy(msg).double())
        else:
            msg = None
            loss = L.LocalizationLoss()(logits, torch.from_numpy(mask).double())
        loss.backward()
        out = dict(x=x, mask=mask, loss=np.float64(float(loss)), logits=logits.detach().numpy().astype(np.float32), dx=xt.grad.numpy().astype(np.float32),
                   cfg=np.array([repr(dict(kind=kind, seed=seed, **c))]))
        if msg is not None:
            out["msg"] = msg
        n = 0
        for k, p in model.named_parameters():
            if p.grad is not None:
                out["g:" + k] = p.grad.numpy().astype(np.float64)
                n += 1
        np.savez_compressed(os.path.join(HERE, f"netgrads_{name}.npz"), **out)
        print(f"wrote netgrads_{name}.npz: loss {float(loss):.6f}, {n} gradient tensors, logits {tuple(logits.shape)}")


if __name__ == "__main__":
    main()
