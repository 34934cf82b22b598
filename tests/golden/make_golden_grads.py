#!/usr/bin/env python3
"""Generate tests/golden/grads_half_*.npz: forward AND backward of one SEANetResnetBlock half
(`Scale -> ELU -> 1x1 (weight norm, no bias) -> depth-wise k5 causal conv (weight norm, bias)`,
/root/reference/modules/seanet.py:39-116 built by dws_conv_block, conv.py:47-88 weight norm) from the
REFERENCE modules' CPU autograd (build container only).  Only data is written: parameters in the live
parametrized layout (g = original0, v = original1), input, upstream gradient, and the gradients torch
computes for x, g, v (both convs) and the bias.

Usage (from repo root):  python tests/golden/make_golden_grads.py
"""
import logging
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def main():
    import torch
    import torch.nn as nn
    sys.dont_write_bytecode = True
    logging.disable(logging.CRITICAL)
    sys.path.insert(0, REF)
    from modules.seanet import dws_conv_block      # reference code (pure torch)
    torch.manual_seed(0)
    torch.set_num_threads(4)
    for tag, (B, C, T, s) in {"small": (3, 8, 37, 0.8660254), "c64": (2, 64, 200, 0.7071068),
                              "c96": (2, 96, 132, 1.0), "c160": (1, 160, 64, 0.5773503)}.items():
        rng = np.random.default_rng(zlib.crc32(tag.encode()) % 1000 + C)   # not hash(): Python salts it per process
        half = nn.Sequential(*dws_conv_block(nn.ELU, {"alpha": 1.0}, C, C, kernel_size=5, causal=True,
                                             norm="weight_norm", bias=True)).double()
        # random parameters in the live weight-norm layout (g far from ||v|| so the fold matters)
        sd = {}
        for k, v in half.state_dict().items():
            a = rng.standard_normal(tuple(v.shape))
            if k.endswith("original0"):
                a = 0.5 + np.abs(a)
            elif k.endswith("original1"):
                a = a * (1.0 / np.sqrt(np.prod(v.shape[1:])))
            else:
                a = a * 0.1
            sd[k] = torch.from_numpy(a.astype(np.float32)).double()
        half.load_state_dict(sd)
        x = torch.from_numpy(rng.standard_normal((B, C, T)).astype(np.float32)).double().requires_grad_(True)
        dy = torch.from_numpy(rng.standard_normal((B, C, T)).astype(np.float32)).double()
        y = half(x * s)
        y.backward(dy)
        p = dict(half.named_parameters())
        out = dict(x=x.detach().numpy().astype(np.float32), dy=dy.numpy().astype(np.float32), pre_scale=np.float32(s),
                   y=y.detach().numpy().astype(np.float32), dx=x.grad.numpy().astype(np.float32))
        for name, short in (("1.conv.conv.parametrizations.weight.original0", "g_pw"),
                            ("1.conv.conv.parametrizations.weight.original1", "v_pw"),
                            ("2.conv.conv.parametrizations.weight.original0", "g_dw"),
                            ("2.conv.conv.parametrizations.weight.original1", "v_dw"),
                            ("2.conv.conv.bias", "b_dw")):
            out[short] = p[name].detach().numpy().astype(np.float32)
            out["d" + short] = p[name].grad.numpy().astype(np.float32)
        path = os.path.join(HERE, f"grads_half_{tag}.npz")
        np.savez_compressed(path, **out)
        print(f"wrote {path}: B={B} C={C} T={T}, |dx|max {np.abs(out['dx']).max():.3f}, keys {sorted(out)[:4]}...")


if __name__ == "__main__":
    main()
