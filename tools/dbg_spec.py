"""spec16 (csrc/wv_h16.hip) with an identity 1x1: the log-magnitude feature P per bin and frame against the oracle -- how the two-term
split's leakage into weak DC / Nyquist bins and the n_fft - hop padding slip were found.  python tools/dbg_spec.py <n_fft> <hop>"""
import os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from oracle import wv_oracle as O
from waveverify_amd import ops
n_fft, hop, T = int(sys.argv[1]), int(sys.argv[2]), 16000
rng = np.random.default_rng(n_fft + hop + T)
C, F, Tf = n_fft, n_fft // 2 + 1, -(-T // hop)
wav = np.clip((0.1 * rng.standard_normal((3, 1, T))).astype(np.float32), -1, 1)
wav[1, 0, : T // 3] = 0.0
wav[2] *= 8.0
w = np.zeros((C, F, 1), np.float32)
for f in range(F): w[f, f, 0] = 1.0
x = np.zeros((3, C, Tf), np.float32)
mag = O.causal_stft_mag(wav, n_fft, hop)
P = ((np.log(np.maximum(mag, np.float32(1e-5))) - np.float32(-4.3)) / np.float32(2.8)).astype(np.float32)
got = ops.h16_spec_block(torch.from_numpy(wav).cuda(), w, ops.h16_from_f32(torch.from_numpy(x).cuda()), n_fft, hop, mean=-4.3, std=2.8, out_scale=1.0)
g = ops.h16_to_f32(got, C).cpu().numpy()[:, :F]
d = np.abs(g - P)
print("max dP", d.max(), "at", np.unravel_index(d.argmax(), d.shape))
for b in range(3):
    print("clip", b, "max", d[b].max(), "per-bin max", np.round(d[b].max(axis=1), 2).tolist())
    print("   per-frame max (first 140 frames)", np.round(d[b].max(axis=0), 1)[:140].tolist())
b, f, t = np.unravel_index(d.argmax(), d.shape)
print("P ref", P[b, f, t], "got", g[b, f, t], "mag", mag[b, f, t], "frame mags min/max", mag[b, :, t].min(), mag[b, :, t].max())
big = np.argwhere(d > 0.02)
print(len(big), "elements above 0.02; first", big[:10].tolist())
print("rows 33..: ", np.abs(ops.h16_to_f32(got, C).cpu().numpy()[:, F:]).max())
