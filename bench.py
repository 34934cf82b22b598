#!/usr/bin/env python3
"""Headline benchmark: clips/s of embed + detect on 1 s / 16 kHz clips, batch 256 per GPU
(BASELINE.json `metric`, configs[1]).  One "step" = one pass of the hot path over one resident
batch: wm = G(x, msg) + x, then the detector's time-averaged bit probabilities on wm and the
>= 0.5 decision.  Inputs are already in HBM when the timed region starts.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Clips are independent units, so N GPUs = N data-parallel shards with NO data-path collective
(SURVEY.md section 8e); weak scaling (256 clips per GPU).  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E spec peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="clips per GPU")
    ap.add_argument("--seconds", type=float, default=1.0)
    ap.add_argument("--cpu-clips", type=int, default=16, help="clips of the CPU-baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", default="embed_detect", choices=["embed_detect", "longform", "detector_stress"],
                    help="embed_detect = BASELINE configs[1] (the headline); longform = configs[3] "
                         "(32 x 30 s, embed+locate+detect); detector_stress = configs[4] (1024 clips, detector only)")
    ap.add_argument("--precision", default="f32", choices=["f32", "f16x3"],
                    help="GEMM core: exact f32 MFMA, or split-f16 (hi+lo, 3 f16 MFMAs, f32 accumulate)")
    return ap.parse_args()


def cpu_baseline(cfgG, cfgD, sdG, sdD, x, msg, n):
    """The reference's CPU path, restated on torch.nn.functional (oracle/wv_oracle_torch.py, pinned to
    the reference's outputs), timed on this box's host cores on a bounded sample of the same workload.
    Reported beside the GPU number, never as it."""
    from oracle import wv_oracle_torch as OT
    threads = min(os.cpu_count() or 1, 16)            # a 1-GPU box's CPU share
    torch.set_num_threads(threads)
    G, D = OT.Net(cfgG, sdG), OT.Net(cfgD, sdD)
    xs, ms = x[:n], msg[:n]
    OT.embed(G, xs[:1], ms[:1])                       # warm-up (oneDNN primitive creation)
    t0 = time.perf_counter()
    wm = OT.embed(G, xs, ms)
    mp = OT.mean_probabilities(OT.detector_logits(D, wm))
    dt = time.perf_counter() - t0
    return dict(value=n / dt, unit="clips/s", cores=threads, kind="port",
                sample=f"{n} clips x 1 s @ 16 kHz, one embed+detect pass of the torch-CPU port of the "
                       f"reference path (fp32, {threads} threads), {dt:.1f} s"), wm.numpy(), mp.numpy()


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1 or os.environ.get("WV_BENCH_FORCE_DIST") == "1":   # the latter: 1-rank rehearsal of the N>1 path
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)          # RCCL; used for barrier + MAX only

    from waveverify_amd import profile
    from waveverify_amd.config import default_config
    from waveverify_amd.init import random_state_dict, synthetic_clips
    from waveverify_amd.nets import HipNet

    if a.workload == "longform":
        a.batch, a.seconds = (32 if a.batch == 256 else a.batch), (30.0 if a.seconds == 1.0 else a.seconds)
    elif a.workload == "detector_stress":
        a.batch = 1024 if a.batch == 256 else a.batch
    T = int(round(a.seconds * 16000))
    B = a.batch
    cfgG, cfgD = default_config("generator"), default_config("detector")
    sdG, sdD = random_state_dict(cfgG, 0), random_state_dict(cfgD, 0)
    G, D = HipNet(cfgG, sdG, dev, precision=a.precision), HipNet(cfgD, sdD, dev, precision=a.precision)
    Lnet = None
    if a.workload == "longform":
        cfgL = default_config("locator")
        Lnet = HipNet(cfgL, random_state_dict(cfgL, 0), dev, precision=a.precision)
    x_np, msg_np = synthetic_clips(B, T, seed=1234 + rank)         # each rank owns its shard
    x, msg = torch.from_numpy(x_np).to(dev), torch.from_numpy(msg_np).to(dev)

    def step():
        if a.workload == "detector_stress":
            mp = D.detector_mean_prob(x)
            return x, mp, mp >= 0.5
        wm = G.generator(x, msg, add_input=True)
        if Lnet is not None:
            Lnet.locator(wm)
        mp = D.detector_mean_prob(wm)
        return wm, mp, mp >= 0.5

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    profile.reset()
    profile.enable(True)                      # HIP event pair around every launch, on the launch stream
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        wm, mp, bits = step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    profile.enable(False)
    if dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    prof = profile.collect()

    if rank != 0:
        if dist:
            dist.destroy_process_group()
        return

    # ---- per-kernel figures from the live event timings of the timed region -------------------
    by_kernel = {}
    for e in prof:
        k = by_kernel.setdefault(e["kernel"], dict(ms=0.0, launches=0, flops=0.0, bytes=0.0))
        for f in ("ms", "launches", "flops", "bytes"):
            k[f] += e[f]
    total_ms = sum(k["ms"] for k in by_kernel.values())
    dom_name, dom = max(by_kernel.items(), key=lambda kv: kv[1]["ms"])
    dom_avg_s = dom["ms"] / dom["launches"] * 1e-3
    ach = dom["flops"] / dom["launches"] / dom_avg_s / 1e12
    traffic, traffic_src, pmc = None, None, None
    try:        # HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/), same workload
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
        if a.workload == "embed_detect" and B == 256 and dom_name in pmc["kernels"]:
            traffic = round(pmc["kernels"][dom_name]["traffic_bytes_per_launch"] / 1e9, 3)
            traffic_src = "profiles/r01_pmc_traffic.json: GB per launch = (2*FETCH_SIZE + WRITE_SIZE) KiB, separate --pmc passes"
    except Exception:
        pass
    roofline = dict(bound="mfma", kernel=dom_name, achieved=round(ach, 2), peak=PEAK_F32_MFMA_TFLOPS,
                    unit="TFLOP/s", frac=round(ach / PEAK_F32_MFMA_TFLOPS, 4), traffic=traffic, traffic_source=traffic_src,
                    algorithmic_gb_per_launch=round(dom["bytes"] / dom["launches"] / 1e9, 3),
                    avg_launch_us=round(dom_avg_s * 1e6, 1), launches_per_step=dom["launches"] // a.steps,
                    share_of_kernel_time=round(dom["ms"] / total_ms, 3),
                    algorithmic_gflop_per_launch=round(dom["flops"] / dom["launches"] / 1e9, 2))
    film = [e for e in prof if e["role"] == "enc.down_film"]
    roofline_film = None
    if film:
        ms = sum(e["ms"] for e in film); by = sum(e["bytes"] for e in film); fl = sum(e["flops"] for e in film)
        n = sum(e["launches"] for e in film)
        # the north_star's "fused Conv1d+FiLM" unit.  With the 1x1 expansion fused in it has
        # AI = 95 FLOP/B, above the f32-matrix ridge (157.3 TF / 8 TB/s = 19.7), so its roofline is
        # the matrix one; the HBM figures are given beside it.
        tf = fl / (ms * 1e-3) / 1e12
        gbs = by / (ms * 1e-3) / 1e9
        roofline_film = dict(kernel="pw_dw (Scale->ELU->1x1->strided DW conv->FiLM)", bound="mfma",
                             achieved=round(tf, 2), peak=PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s",
                             frac=round(tf / PEAK_F32_MFMA_TFLOPS, 4),
                             traffic=(round(pmc["kernels"][film[0]["kernel"]]["traffic_bytes_per_launch"] / 1e9, 3)
                                      if pmc and a.workload == "embed_detect" and B == 256 and a.precision == "f32"
                                      and film[0]["kernel"] in pmc["kernels"] else None),
                             algorithmic_gb_per_launch=round(by / n / 1e9, 3),
                             hbm_gbs=round(gbs, 1), hbm_frac=round(gbs / PEAK_HBM_GBS, 4),
                             arithmetic_intensity_flop_per_byte=round(fl / by, 1),
                             launches_per_step=n // a.steps,
                             algorithmic_mb_per_clip=round(by / n * (n // a.steps) / B / 1e6, 2))
    kernels = sorted(({"kernel": k, "ms_per_step": round(v["ms"] / a.steps, 3),
                       "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2) if v["ms"] else 0.0,
                       "gbs": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) if v["ms"] else 0.0}
                      for k, v in by_kernel.items()), key=lambda d: -d["ms_per_step"])

    # ---- parity beside the number: BER / waveform error vs the oracle on a sample --------------
    metric = {"embed_detect": "clips/sec embed+detect, 1s@16kHz bs=256",
              "longform": "clips/sec embed+locate+detect, 30s@16kHz bs=32",
              "detector_stress": "clips/sec detect, 1s@16kHz bs=1024"}[a.workload]
    out = dict(metric=metric, value=round(world * B * a.steps / elapsed, 2),
               unit="clips/s", n_gpus=world, steps=a.steps, warmup=a.warmup,
               ms_per_step=round(elapsed / a.steps * 1e3, 3), higher_is_better=True, scaling="weak",
               vs_baseline=None, dtype=a.precision, data="synthetic",
               config=dict(workload=f"{a.workload}: {B} clips x {a.seconds:g} s @ 16 kHz per GPU "
                                    f"(BASELINE.json configs[{dict(embed_detect=1, longform=3, detector_stress=4)[a.workload]}]), "
                                    "seeded random weights",
                           batch_per_gpu=B, global_batch=B * world, clip_samples=T,
                           parallelism=f"dp{world} (independent clip shards, no data-path collective)"),
               roofline=roofline, roofline_film=roofline_film, kernels=kernels[:8],
               kernel_time_ms_per_step=round(total_ms / a.steps, 3))
    if not a.no_cpu_baseline and a.workload == "embed_detect":
        cb, wm_ref, mp_ref = cpu_baseline(cfgG, cfgD, sdG, sdD, x_np, msg_np, min(a.cpu_clips, B))
        n = wm_ref.shape[0]
        out["cpu_baseline"] = cb
        out["parity"] = dict(
            clips_checked=n,
            wm_max_abs_err=float(np.abs(wm[:n].cpu().numpy() - wm_ref).max()),
            mean_prob_max_abs_err=float(np.abs(mp[:n].cpu().numpy() - mp_ref).max()),
            ber_vs_oracle=float(((mp_ref >= 0.5) != bits[:n].cpu().numpy()).mean()))
    print(json.dumps(out), flush=True)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
