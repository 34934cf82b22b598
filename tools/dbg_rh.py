"""resblock16 (csrc/wv_h16.hip): its three output variants against each other, run to run -- how the DPP wait-state hazard in the
stencils was found (a 128-register build differed in one lane pair per row group, differently each run).  python tools/dbg_rh.py <C> <T>"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from waveverify_amd import ops
C, T, B = int(sys.argv[1]), int(sys.argv[2]), 3
rng = np.random.default_rng(C * 3 + T)
rnd = lambda *s, scale=1.0: (scale * rng.standard_normal(s)).astype(np.float32)
X = rnd(B, C, T)
w1, w2 = rnd(C, C, 1, scale=C ** -0.5), rnd(C, C, 1, scale=C ** -0.5)
d1, d2 = rnd(C, 1, 5, scale=0.45), rnd(C, 1, 5, scale=0.45)
b1, b2 = rnd(C, scale=0.1), rnd(C, scale=0.1)
X16 = ops.h16_from_f32(torch.from_numpy(X).cuda())
kw = dict(pre_scale=0.866, out_scale=0.41)
for rep in range(3):
    got, gact = ops.h16_resblock(X16, w1, d1, b1, w2, d2, b2, act_scale=0.7071, **kw)
    raw = ops.h16_resblock(X16, w1, d1, b1, w2, d2, b2, **kw)
    act = ops.h16_resblock(X16, w1, d1, b1, w2, d2, b2, act_scale=0.7071, want_raw=False, **kw)
    got2, gact2 = ops.h16_resblock(X16, w1, d1, b1, w2, d2, b2, act_scale=0.7071, **kw)
    for name, a, b in (("raw vs both", raw, got), ("act vs both", act, gact), ("both vs both", got2, got), ("both act vs both act", gact2, gact)):
        ne = (a != b)
        if ne.any():
            idx = ne.nonzero()
            print(rep, name, int(ne.sum()), "differ; first", idx[:5].tolist(), "t range", int(idx[:, 2].min()), int(idx[:, 2].max()), "groups", sorted(set(idx[:, 1].tolist()))[:20],
                  "max", float((a.float() - b.float()).abs().max()))
        else:
            print(rep, name, "equal")
