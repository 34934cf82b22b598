import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from waveverify_amd.core import WaveVerify
from waveverify_amd.init import synthetic_clips
wv = WaveVerify.random_init(seed=0, device="cuda:0")
gen, det = wv.model.generator, wv.model.detector
for B in (1, 8):
    x_np, msg_np = synthetic_clips(B, 16000, seed=1)
    x, msg = torch.from_numpy(x_np).cuda(), torch.from_numpy(msg_np).cuda()
    def step():
        wm = gen.generator(x, msg, add_input=True)
        return wm, det.detector_mean_prob(wm)
    for _ in range(3): wm0, p0 = step()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(50): step()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t) / 50 * 1e3
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): step()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        wm1, p1 = step()
    g.replay(); torch.cuda.synchronize()
    ok = torch.equal(wm1, wm0) and torch.equal(p1, p0)
    t = time.perf_counter()
    for _ in range(50): g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t) / 50 * 1e3
    print(f"B={B}: eager {eager:.3f} ms, graph {graph:.3f} ms, identical={ok}", flush=True)
