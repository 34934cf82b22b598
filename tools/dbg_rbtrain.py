#!/usr/bin/env python3
"""The training forward of a ResnetBlock through the one-launch kernel: its four saved tensors (u, v, H0, H1) against torch float64.
python tools/dbg_rbtrain.py [C T B]"""
import os, sys
import numpy as np, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from waveverify_amd.train import TrainBlock
C_, T, B = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (64, 200, 2)
g = torch.Generator(device="cpu").manual_seed(0)
x = torch.randn(B, C_, T, generator=g).cuda()
ps = []
for i in range(2):
    ps.append(dict(g_pw=torch.rand(C_, generator=g).cuda() + 0.5, v_pw=(torch.randn(C_, C_, generator=g) * C_ ** -0.5).cuda(),
                   g_dw=torch.rand(C_, generator=g).cuda() + 0.5, v_dw=(torch.randn(C_, 5, generator=g) * 0.4).cuda(), b_dw=(torch.randn(C_, generator=g) * 0.1).cuda()))
rsp = torch.tensor([0.8]).cuda()
pre, rs = 0.866, 0.577
blk = TrainBlock(C_)
y, saved = blk.forward(x, ps, rsp, pre, rs)
torch.cuda.synchronize()
act = (B * C_ * T * 4 + 255) // 256 * 256
tens = [saved[i * act:i * act + B * C_ * T * 4].view(torch.float32).view(B, C_, T) for i in range(4)]
def wn(gw, v): return (v.double() * (gw.double() / v.double().reshape(v.shape[0], -1).norm(dim=1)).reshape(-1, *([1] * (v.dim() - 1))))
xd = x.double()
W1, D1, W2, D2 = wn(ps[0]["g_pw"], ps[0]["v_pw"]), wn(ps[0]["g_dw"], ps[0]["v_dw"]), wn(ps[1]["g_pw"], ps[1]["v_pw"]), wn(ps[1]["g_dw"], ps[1]["v_dw"])
a = F.elu(xd * pre)
H0 = torch.einsum("mk,bkt->bmt", W1, a)
u = F.conv1d(F.pad(H0, (4, 0)), D1[:, None, :], ps[0]["b_dw"].double(), groups=C_)     # saved BEFORE the second half's ELU
H1 = torch.einsum("mk,bkt->bmt", W2, F.elu(u))
v = F.conv1d(F.pad(H1, (4, 0)), D2[:, None, :], ps[1]["b_dw"].double(), groups=C_)
yr = xd + rs * 0.8 * v
for name, got, ref in (("u", tens[0], u), ("v", tens[1], v), ("H0", tens[2], H0), ("H1", tens[3], H1), ("y", y, yr)):
    d = (got.double() - ref).abs()
    bad = (d > 1e-4 * ref.abs().max()).nonzero()
    print(f"{name}: max|d| {float(d.max()):.3e} of {float(ref.abs().max()):.2f}; bad elements {bad.shape[0]}", "first:", bad[:3].tolist(), "times with errors:", sorted(set(bad[:, 2].tolist()))[:12] if bad.shape[0] else "")
# ---- backward on those saved tensors against torch autograd (float64)
dy = torch.randn(B, C_, T, generator=g).cuda()
xr = x.double().clone().requires_grad_(True)
a = F.elu(xr * pre)
H0 = torch.einsum("mk,bkt->bmt", W1, a)
u = F.elu(F.conv1d(F.pad(H0, (4, 0)), D1[:, None, :], ps[0]["b_dw"].double(), groups=C_))
H1 = torch.einsum("mk,bkt->bmt", W2, u)
v = F.conv1d(F.pad(H1, (4, 0)), D2[:, None, :], ps[1]["b_dw"].double(), groups=C_)
(xr + rs * 0.8 * v).backward(dy.double())
gr = blk.backward(x, ps, rsp, pre, rs, dy, saved)
d = (gr["dx"].double() - xr.grad).abs()
print(f"dx: max|d| {float(d.max()):.3e} of {float(xr.grad.abs().max()):.2f}; bad times:", sorted(set((d > 1e-3).nonzero()[:, 2].tolist()))[:20], "bad rows:", sorted(set((d > 1e-3).nonzero()[:, 1].tolist()))[:20])
