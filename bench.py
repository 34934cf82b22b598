#!/usr/bin/env python3
"""Headline benchmark: clips/s of embed + detect on 1 s / 16 kHz clips, batch 256 per GPU
(BASELINE.json `metric`, configs[1]).  One "step" = one pass of the hot path over one resident
batch: wm = G(x, msg) + x, then the detector's time-averaged bit probabilities on wm and the
>= 0.5 decision.  Inputs are already in HBM when the timed region starts.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Started as plain `python bench.py --gpus N` (N > 1, no WORLD_SIZE in the environment) the script launches that
torch.distributed.run command itself as a CHILD process -- before anything touches the GPU, never by exec -- relays rank 0's JSON
line and exits with the child's code.

Clips are independent units, so N GPUs = N data-parallel shards with NO data-path collective
(SURVEY.md section 8e); weak scaling (256 clips per GPU).  Rank 0 prints ONE JSON line.

Two passes: `value` / `ms_per_step` come from a CLEAN pass (W warm-up + exactly K steps, no event
recording, barrier + synchronize on both sides, MAX over ranks); the per-kernel table and the roofline
blocks come from a second, PROFILED pass of the same K steps (a hipEvent pair around every launch on
the launch stream, ~190 pairs per step), whose own wall time is reported as `profiled_ms_per_step`.

Other workloads (--workload): longform (configs[3]), detector_stress (configs[4]) and grad_allreduce
(configs[2]: the training step's only exchange, a bucketed RCCL all-reduce of 56.1 + 170.1 MB of fp32
gradients; reports ms per all-reduce and bus GB/s).  `--precision f16` (embed_detect, longform, detector_stress) runs the
f16-operand / f32-accumulate mode of the three nets (csrc/wv_h16.hip; configs[1] "bf16" / configs[4] "fp16" as BASELINE words them) and
prints ITS OWN line: dtype "f16", the exact-f32 mode timed beside it, both modes' outputs against each other and against the oracle.
The default (exact f32) line carries a `reduced_precision` block -- that mode's ms per step, wm max|d| and BER on the same batch -- so
the driver's record holds it; `value` is always the exact path's.
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
PEAK_F16_MFMA_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense f16 / bf16 (v_mfma_f32_32x32x16_f16, 32 cycles per SIMD)
PMC_FILE = os.path.join(ROOT, "profiles", "r04_pmc_traffic.json")
PMC_FILE_F16 = os.path.join(ROOT, "profiles", "r04_pmc_traffic_f16.json")   # the same passes over the f16 mode's command (tools/profile_f16.sh)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="clips per GPU")
    ap.add_argument("--seconds", type=float, default=1.0)
    ap.add_argument("--cpu-clips", type=int, default=16, help="clips of the CPU-baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", default="embed_detect",
                    choices=["embed_detect", "longform", "detector_stress", "grad_allreduce", "train_step", "rendezvous"],
                    help="embed_detect = BASELINE configs[1] (the headline); longform = configs[3] "
                         "(32 x 30 s, embed+locate+detect); detector_stress = configs[4] (1024 clips, detector "
                         "only); grad_allreduce = configs[2]'s gradient exchange (no model compute); train_step = the "
                         "part of configs[2]'s training step that runs on the HIP training units (G+D+L, 64 clips per GPU)")
    ap.add_argument("--bucket-mb", type=float, default=25.0, help="grad_allreduce: bucket size")
    ap.add_argument("--precision", default="f32", choices=["f32", "f16"],
                    help="f16 = the f16-operand / f32-accumulate mode of the nets (csrc/wv_h16.hip) for embed_detect / longform / detector_stress: "
                         "its own line (dtype f16) with its own parity block and roofs; the headline `value` is always exact f32")
    return ap.parse_args()


def cpu_baseline(cfgG, cfgD, sdG, sdD, x, msg, n):
    """The reference's CPU path, restated on torch.nn.functional (oracle/wv_oracle_torch.py, pinned to
    the reference's outputs), timed on this box's host cores on a bounded sample of the same workload:
    1 warm-up + 3 timed repetitions, generator / detector split.  Reported
    beside the GPU number, never as it."""
    from oracle import wv_oracle_torch as OT
    threads = min(os.cpu_count() or 1, 16)            # a 1-GPU box's CPU share
    torch.set_num_threads(threads)
    G, D = OT.Net(cfgG, sdG), OT.Net(cfgD, sdD)
    xs, ms = x[:n], msg[:n]
    wm = OT.embed(G, xs, ms)                          # warm-up (oneDNN primitive creation), full sample
    OT.mean_probabilities(OT.detector_logits(D, wm))
    tg, td = [], []
    for _ in range(3):
        t0 = time.perf_counter()
        wm = OT.embed(G, xs, ms)
        t1 = time.perf_counter()
        mp = OT.mean_probabilities(OT.detector_logits(D, wm))
        t2 = time.perf_counter()
        tg.append(t1 - t0); td.append(t2 - t1)
    dt = float(np.median([a + b for a, b in zip(tg, td)]))
    return dict(value=n / dt, unit="clips/s", cores=threads, host_cpus=os.cpu_count(), kind="port",
                generator_s=round(float(np.median(tg)), 3), detector_s=round(float(np.median(td)), 3),
                sample=f"{n} clips x 1 s @ 16 kHz, embed+detect by the torch-CPU port of the reference path "
                       f"(fp32, {threads} threads of the host's {os.cpu_count()} CPUs), 1 warm-up + 3 repetitions, median {dt:.2f} s per pass "
                       f"(generator {np.median(tg):.2f} s, detector {np.median(td):.2f} s)"), wm.numpy(), mp.numpy()


def grad_allreduce(a, dev, dist, world, rank):
    """configs[2]: the per-step gradient all-reduce of the reference's data-parallel training
    (scripts/train.py:875-876,1277,1347), on synthetic fp32 gradients of the real sizes."""
    from waveverify_amd import parallel
    bucket = int(a.bucket_mb * 1024 * 1024)
    payloads = parallel.GRAD_PAYLOAD_BYTES
    # parameter granularity does not matter to the wire: model each payload as 1 MB tensors
    grads = [torch.randn(250_000, device=dev) for nbytes in payloads.values() for _ in range(nbytes // 1_000_000)]
    total = sum(g.numel() * 4 for g in grads)
    buckets = parallel.plan_buckets([g.numel() for g in grads], bucket_bytes=bucket)

    def step():
        parallel.allreduce_mean_(grads, buckets)

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        ms = elapsed / a.steps * 1e3
        bus = 2.0 * (world - 1) / world * total / (ms * 1e-3) / 1e9 if world > 1 else 0.0
        print(json.dumps(dict(
            metric="grad all-reduce ms per training step (56.1 + 170.1 MB fp32, bucketed)", value=round(ms, 3), unit="ms",
            n_gpus=world, rccl_ranks=dist.get_world_size() if dist else 0, steps=a.steps, warmup=a.warmup, ms_per_step=round(ms, 3),
            higher_is_better=False, scaling="strong", vs_baseline=None, dtype="f32", data="synthetic",
            config=dict(workload="grad_allreduce: BASELINE.json configs[2] gradient exchange only (no model compute)",
                        payload_bytes=total, payloads=payloads, bucket_bytes=bucket, buckets=len(buckets),
                        backend="nccl (RCCL)" if dist else "none (single rank: flatten/unflatten only)",
                        world_size=dist.get_world_size() if dist else 1),
            bus_gb_per_s=round(bus, 2),
            note="bus GB/s = 2(N-1)/N * bytes / time (ring); xGMI gives one ~153 GB/s link per peer")), flush=True)


def train_step(a, dev, dist, world, rank):
    """BASELINE configs[2] (64 clips x 1 s per GPU): the generator-update step of the reference's loop on the HIP training units
    (waveverify_amd.train.WatermarkTrainer): G forward, one-launch augmentation, D and L forward, DecodingLoss / LocalizationLoss /
    waveform loss, backward through D, L, the augmentation and G with live weight norm, mean all-reduce of the three flat gradient
    arenas (RCCL when N > 1), generator-only clipping, AdamW.  The reference's step also has audio effects, mel / STFT losses and a
    discriminator, which stay on PyTorch and are NOT in this number, so it is not the headline metric."""
    from waveverify_amd import profile
    from waveverify_amd.config import default_config
    from waveverify_amd.init import random_state_dict, synthetic_clips
    from waveverify_amd.train import WatermarkTrainer
    B = 64 if a.batch == 256 else a.batch
    T = int(round(a.seconds * 16000))
    x_np, msg_np = synthetic_clips(B, T, seed=1234 + rank)
    x, msg = torch.from_numpy(x_np).to(dev), torch.from_numpy(msg_np.astype(np.float32)).to(dev)
    cfgs = [default_config(k) for k in ("generator", "detector", "locator")]
    sds = [random_state_dict(c, 0, parametrized=True) for c in cfgs]
    tr = WatermarkTrainer(cfgs[0], sds[0], cfgs[1], sds[1], cfgs[2], sds[2], device=dev)
    np.random.seed(1234 + rank)
    torch.manual_seed(1234 + rank)
    outs = []

    def step():
        outs.append(tr.step(x, msg))

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # anatomy: a second pass of the same K steps with a hipEvent pair around every instrumented launch (the K1 GEMMs of forward and
    # backward, the dW GEMM); the stencil / reduction / optimizer kernels of the training library carry no events and make up the rest
    profile.reset()
    profile.enable(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    elapsed_prof = time.perf_counter() - t0
    profile.enable(False)
    by, stencil = {}, {}
    for e in profile.collect():
        k = (stencil if e["kernel"].startswith("dw_bwd") else by).setdefault(e["kernel"], dict(ms=0.0, launches=0, flops=0.0, bytes=0.0))
        for f in ("ms", "launches", "flops", "bytes"):
            k[f] += e[f]
    if rank == 0:
        ms = elapsed / a.steps * 1e3
        kernels = sorted(({"kernel": k, "ms_per_step": round(v["ms"] / a.steps, 3), "launches_per_step": v["launches"] // a.steps,
                           "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2) if v["ms"] else 0.0,
                           "frac_of_f32_mfma_peak": round(v["flops"] / (v["ms"] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 3) if v["ms"] else 0.0}
                          for k, v in by.items()), key=lambda d: -d["ms_per_step"])
        inst_ms = sum(v["ms"] for v in by.values()) / a.steps
        inst_fl = sum(v["flops"] for v in by.values()) / a.steps
        pick = lambda o: {k: round(float(o[k].item()), 5) for k in ("dec/loss", "loc/loss", "waveform/loss")}       # noqa: E731
        print(json.dumps(dict(
            metric="clips/sec training step (generator update: G+D+L forward/backward, BCE + waveform losses), 1s@16kHz bs=64 per GPU",
            value=round(world * B * a.steps / elapsed, 2), unit="clips/s", n_gpus=world, rccl_ranks=dist.get_world_size() if dist else 0,
            steps=a.steps, warmup=a.warmup,
            ms_per_step=round(ms, 3), higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f32", data="synthetic",
            config=dict(workload=f"train_step: BASELINE.json configs[2] per-GPU batch ({B} clips x {a.seconds:g} s); the part of the reference's "
                                 "generator update that runs on the HIP training units (no audio effects, mel/STFT losses or discriminator)",
                        batch_per_gpu=B, global_batch=B * world, clip_samples=T,
                        parallelism=f"dp{world} (three flat gradient arenas, bucketed mean all-reduce)",
                        parameters=dict(generator=int(tr.G.arena.numel()), detector=int(tr.D.arena.numel()), locator=int(tr.L.arena.numel()))),
            losses_first=pick(outs[0]), losses_last=pick(outs[-1]),
            roofline=dict(bound="mfma", peak=PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s",
                          instrumented_ms_per_step=round(inst_ms, 3), instrumented_tflop_per_step=round(inst_fl / 1e12, 3),
                          achieved=round(inst_fl / (inst_ms * 1e-3) / 1e12, 2) if inst_ms else None,
                          frac=round(inst_fl / (inst_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4) if inst_ms else None,
                          step_frac=round(inst_fl / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                          note="matrix kernels only (K1 forward / backward GEMMs, dW GEMM): their flops over their own event time; step_frac = the "
                               "same flops over the whole step, i.e. with the bandwidth-bound stencil / reduction / optimizer kernels in the denominator"),
            roofline_stencil_backward=dict(
                bound="hbm", peak=PEAK_HBM_GBS, unit="GB/s",
                ms_per_step=round(sum(v["ms"] for v in stencil.values()) / a.steps, 3),
                achieved=round(sum(v["bytes"] for v in stencil.values()) / max(sum(v["ms"] for v in stencil.values()), 1e-9) / 1e6, 1),
                frac=round(sum(v["bytes"] for v in stencil.values()) / max(sum(v["ms"] for v in stencil.values()), 1e-9) / 1e6 / PEAK_HBM_GBS, 4),
                kernels=sorted(({"kernel": k, "ms_per_step": round(v["ms"] / a.steps, 3), "launches_per_step": v["launches"] // a.steps,
                                 "gbs": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) if v["ms"] else 0.0} for k, v in stencil.items()),
                               key=lambda d: -d["ms_per_step"]),
                note="the depth-wise stencils' transposes (dh and the tap / bias sums; the second half's also the res_scale dot): algorithmic bytes "
                     "= read dy, read h (and v), write dh, over their own event time") if stencil else None,
            kernels=kernels[:12], profiled_ms_per_step=round(elapsed_prof / a.steps * 1e3, 3),
            note="backward on saved activations (the blocks' 1x1 outputs kept from forward), activation derivative / residual / scale "
                 "epilogues fused into the K1 GEMMs, parameter gradients written in place into one flat arena per net")), flush=True)


def self_launch(a) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as a child `torch.distributed.run` (the reference's
    counterpart is the Accelerator's launcher-driven DDP, scripts/train.py:180,875-876).  The parent never initialises the GPU and
    never replaces itself; it relays the one JSON line rank 0 prints and returns the child's exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:                                   # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")             # dmabuf IPC: RCCL across processes needs it on this driver
    child = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in child.stdout.splitlines():                          # rank 0's single JSON line; anything else goes to stderr
        try:
            if ln.lstrip().startswith("{"):
                json.loads(ln)
                line = ln
                continue
        except ValueError:
            pass
        print(ln, file=sys.stderr)
    if line is not None:
        print(line, flush=True)
    elif child.returncode == 0:
        print("bench.py: the ranks printed no JSON line", file=sys.stderr)
        return 1
    return child.returncode


def rendezvous(a, dist, world, rank, backend):
    """Launcher check (no model, no GPU needed with WV_BENCH_BACKEND=gloo): every rank joins, a barrier and the MAX all-reduce of the
    elapsed time run exactly as in the timed workloads, rank 0 prints the line."""
    t0 = time.perf_counter()
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        print(json.dumps(dict(metric="launcher rendezvous (no compute)", value=round(elapsed * 1e3, 3), unit="ms", n_gpus=world,
                              rccl_ranks=world if backend == "nccl" else 0, backend=backend, steps=0, warmup=0,
                              higher_is_better=False, config=dict(workload="rendezvous"))), flush=True)


def main():
    a = parse()
    if a.precision != "f32" and a.workload not in ("embed_detect", "longform", "detector_stress"):   # before any GPU call
        raise SystemExit("--precision f16 exists for the inference workloads (embed_detect, longform, detector_stress); training is exact f32")
    env_world = os.environ.get("WORLD_SIZE")
    if a.gpus > 1 and env_world is None:
        raise SystemExit(self_launch(a))                          # before any GPU call
    world = int(env_world or "1")
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and (world > 1 or a.gpus > 1):
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    backend = os.environ.get("WV_BENCH_BACKEND", "nccl")          # test-only switch: "gloo" runs the launcher path without GPUs
    if backend not in ("nccl", "gloo") or (backend == "gloo" and a.workload != "rendezvous"):
        raise SystemExit("WV_BENCH_BACKEND=gloo is for --workload rendezvous only")
    dist = None
    if backend == "gloo":
        if world > 1:
            import torch.distributed as dist
            dist.init_process_group("gloo")
        rendezvous(a, dist, world, rank, backend)
        if dist:
            dist.destroy_process_group()
        return
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1 or os.environ.get("WV_BENCH_FORCE_DIST") == "1":   # the latter: 1-rank rehearsal of the N>1 path
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)          # RCCL; used for barrier + MAX only (embed/detect)
        world = dist.get_world_size()                            # n_gpus is what RCCL reports
    if a.workload == "rendezvous":
        rendezvous(a, dist, world, rank, backend)
        if dist:
            dist.destroy_process_group()
        return
    if a.workload in ("grad_allreduce", "train_step"):
        (grad_allreduce if a.workload == "grad_allreduce" else train_step)(a, dev, dist, world, rank)
        if dist:
            dist.destroy_process_group()
        return

    from waveverify_amd import profile
    from waveverify_amd.config import default_config
    from waveverify_amd.init import random_state_dict, synthetic_clips
    from waveverify_amd.nets import HipNet

    if a.workload == "longform":
        a.batch, a.seconds = (32 if a.batch == 256 else a.batch), (30.0 if a.seconds == 1.0 else a.seconds)
    elif a.workload == "detector_stress":
        a.batch = 1024 if a.batch == 256 else a.batch
    f16 = a.precision == "f16"
    T = int(round(a.seconds * 16000))
    B = a.batch
    cfgG, cfgD = default_config("generator"), default_config("detector")
    sdG, sdD = random_state_dict(cfgG, 0), random_state_dict(cfgD, 0)
    G, D = HipNet(cfgG, sdG, dev), HipNet(cfgD, sdD, dev)
    Lnet = None
    if a.workload == "longform":
        cfgL = default_config("locator")
        Lnet = HipNet(cfgL, random_state_dict(cfgL, 0), dev)
    x_np, msg_np = synthetic_clips(B, T, seed=1234 + rank)         # each rank owns its shard
    x, msg = torch.from_numpy(x_np).to(dev), torch.from_numpy(msg_np).to(dev)

    def step(precision=a.precision):
        if a.workload == "detector_stress":
            mp = D.detector_mean_prob(x, precision=precision)
            return x, mp, mp >= 0.5
        wm = G.generator(x, msg, add_input=True, precision=precision)
        if Lnet is not None:
            Lnet.locator(wm, precision=precision)
        mp = D.detector_mean_prob(wm, precision=precision)
        return wm, mp, mp >= 0.5

    def timed(profiled: bool):
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        if profiled:
            profile.reset()
            profile.enable(True)                  # HIP event pair around every launch, on the launch stream
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            out = step()
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        dt = time.perf_counter() - t0
        profile.enable(False)
        if dist:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, out

    for _ in range(a.warmup):
        step()
    elapsed, (wm, mp, bits) = timed(False)        # the number: clean pass, events off
    elapsed_prof, _ = timed(True)                 # the anatomy: same K steps with per-launch events
    prof = profile.collect()
    with_locator = None
    if a.workload == "embed_detect":              # BASELINE configs[1] words the step "generator+locator+detector": the same K steps
        cfgL = default_config("locator")          # with the locator's forward on wm as well, reported beside the headline
        Lnet = HipNet(cfgL, random_state_dict(cfgL, 0), dev)
        step()
        el, _ = timed(False)
        with_locator = dict(metric="clips/sec embed+locate+detect, 1s@16kHz bs=256", value=round(world * B * a.steps / el, 2),
                            ms_per_step=round(el / a.steps * 1e3, 3))
        Lnet = None

    if rank != 0:
        if dist:
            dist.destroy_process_group()
        return

    # ---- per-kernel figures from the live event timings of the profiled pass --------------------
    by_kernel = {}
    for e in prof:
        k = by_kernel.setdefault(e["kernel"], dict(ms=0.0, launches=0, flops=0.0, bytes=0.0))
        for f in ("ms", "launches", "flops", "bytes"):
            k[f] += e[f]
    total_ms = sum(k["ms"] for k in by_kernel.values())
    dom_name, dom = max(by_kernel.items(), key=lambda kv: kv[1]["ms"])
    dom_avg_s = dom["ms"] / dom["launches"] * 1e-3
    ach = dom["flops"] / dom["launches"] / dom_avg_s / 1e12
    pmc = None
    try:        # HBM bytes per launch from separate rocprofv3 --pmc passes over this same command (profiles/), valid for the
        pmc = json.load(open(PMC_FILE_F16 if f16 else PMC_FILE))   # library build they were taken on only: dropped when the kernel sources changed since
        from waveverify_amd import _lib
        if pmc.get("library") != _lib.load().wv_version().decode():
            pmc = None
    except Exception:
        pmc = None
    headline = a.workload == "embed_detect" and B == 256 and T == 16000   # the shape the counter passes ran on (both modes: tools/profile_bench.sh, profile_f16.sh)

    def traffic_of(kernel):
        kernel = kernel.replace(",flat", "")       # flat tiling is a launch-time property of the same kernel symbol
        if pmc and headline and kernel in pmc.get("kernels", {}):
            return round(pmc["kernels"][kernel]["traffic_bytes_per_launch"] / 1e9, 3)
        return None

    # which roof bounds the dominant kernel: its algorithmic arithmetic intensity against the f32-matrix ridge
    dom_ai = dom["flops"] / max(dom["bytes"], 1.0)
    # the matrix roof of the pipe the dominant kernel runs on: the f16 mode's own kernels (conv_pre16, resblock16, spec16, conv16) on the
    # f16 pipe, everything else (incl. the f32 tail of the f16 mode) on the f32 pipe
    on_f16 = f16 and ("16" in dom_name.split("<")[0])
    peak_mfma = PEAK_F16_MFMA_TFLOPS if on_f16 else PEAK_F32_MFMA_TFLOPS
    ridge = peak_mfma * 1e12 / (PEAK_HBM_GBS * 1e9)                    # 19.7 FLOP/B (f32), 312 FLOP/B (f16)
    gbs_dom = dom["bytes"] / dom["launches"] / dom_avg_s / 1e9
    if dom_ai >= ridge:
        roof = dict(bound="mfma", achieved=round(ach, 2), peak=peak_mfma, unit="TFLOP/s",
                    frac=round(ach / peak_mfma, 4))
    else:
        roof = dict(bound="hbm", achieved=round(gbs_dom, 1), peak=PEAK_HBM_GBS, unit="GB/s",
                    frac=round(gbs_dom / PEAK_HBM_GBS, 4))
    roofline = dict(kernel=dom_name, **roof, traffic=traffic_of(dom_name),
                    traffic_source=(f"profiles/{os.path.basename(PMC_FILE_F16 if f16 else PMC_FILE)}: GB per launch = (2*FETCH_SIZE + WRITE_SIZE) KiB, "
                                    "separate rocprofv3 --pmc passes over this command") if traffic_of(dom_name) else None,
                    arithmetic_intensity_flop_per_byte=round(dom_ai, 1), ridge_flop_per_byte=round(ridge, 1),
                    tflops=round(ach, 2), hbm_gbs=round(gbs_dom, 1),
                    algorithmic_gb_per_launch=round(dom["bytes"] / dom["launches"] / 1e9, 3),
                    avg_launch_us=round(dom_avg_s * 1e6, 1), launches_per_step=dom["launches"] // a.steps,
                    share_of_kernel_time=round(dom["ms"] / total_ms, 3),
                    algorithmic_gflop_per_launch=round(dom["flops"] / dom["launches"] / 1e9, 2))
    film = [e for e in prof if e["role"] == "enc.down_film"]     # (the f16 mode's FiLM convs run under "enc16.down_film": no block for them)
    roofline_film = None
    if film:
        ms = sum(e["ms"] for e in film); by = sum(e["bytes"] for e in film); fl = sum(e["flops"] for e in film)
        n = sum(e["launches"] for e in film)
        # the north_star's "fused Conv1d+FiLM" unit.  With the 1x1 expansion fused in it has
        # AI ~ 70-95 FLOP/B, above the f32-matrix ridge (157.3 TF / 8 TB/s = 19.7), so its roofline is
        # the matrix one; the HBM figures are given beside it.
        tf = fl / (ms * 1e-3) / 1e12
        gbs = by / (ms * 1e-3) / 1e9
        roofline_film = dict(kernel="pw_dw (ELU(s*x) -> 1x1 -> strided DW conv -> FiLM [-> ELU copy])", bound="mfma",
                             achieved=round(tf, 2), peak=PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s",
                             frac=round(tf / PEAK_F32_MFMA_TFLOPS, 4),
                             algorithmic_gb_per_launch=round(by / n / 1e9, 3),
                             hbm_gbs=round(gbs, 1), hbm_frac=round(gbs / PEAK_HBM_GBS, 4),
                             arithmetic_intensity_flop_per_byte=round(fl / by, 1),
                             launches_per_step=n // a.steps,
                             per_launch=[dict(kernel=e["kernel"], us=round(e["ms"] / e["launches"] * 1e3, 1),
                                              tflops=round(e["flops"] / (e["ms"] * 1e-3) / 1e12, 1),
                                              launches_per_step=e["launches"] // a.steps) for e in film],
                             algorithmic_mb_per_clip=round(by / a.steps / B / 1e6, 2),
                             note="north_star prices this unit against the HBM roof; with the 1x1 expansion fused in (SURVEY 8d allows it) its "
                                  "arithmetic intensity lies above the f32-matrix ridge (19.7 FLOP/B), so the MATRIX roof is the binding one: "
                                  "`frac` is of that roof, `hbm_frac` is what its algorithmic bytes reach of 8 TB/s")
    kernels = sorted(({"kernel": k, "ms_per_step": round(v["ms"] / a.steps, 3),
                       "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2) if v["ms"] else 0.0,
                       "gbs": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) if v["ms"] else 0.0}
                      for k, v in by_kernel.items()), key=lambda d: -d["ms_per_step"])

    metric = {"embed_detect": "clips/sec embed+detect, 1s@16kHz bs=256",
              "longform": "clips/sec embed+locate+detect, 30s@16kHz bs=32",
              "detector_stress": "clips/sec detect, 1s@16kHz bs=1024"}[a.workload]
    if f16:
        metric = metric.replace(",", " (f16-operand / f32-accumulate mode),", 1)
    step_flops = sum(v["flops"] for v in by_kernel.values()) / a.steps
    out = dict(metric=metric, value=round(world * B * a.steps / elapsed, 2),
               unit="clips/s", n_gpus=world, rccl_ranks=dist.get_world_size() if dist else 0, steps=a.steps, warmup=a.warmup,
               ms_per_step=round(elapsed / a.steps * 1e3, 3), higher_is_better=True, scaling="weak",
               vs_baseline=None, dtype="f16" if f16 else "f32", data="synthetic",
               config=dict(workload=f"{a.workload}: {B} clips x {a.seconds:g} s @ 16 kHz per GPU "
                                    f"(BASELINE.json configs[{dict(embed_detect=1, longform=3, detector_stress=4)[a.workload]}]), "
                                    "seeded random weights",
                           batch_per_gpu=B, global_batch=B * world, clip_samples=T,
                           parallelism=f"dp{world} (independent clip shards, no data-path collective)"),
               with_locator=with_locator, roofline=roofline, roofline_film=roofline_film, kernels=kernels[:10],
               profiled_pass=dict(profiler_on=True, ms_per_step=round(elapsed_prof / a.steps * 1e3, 3),
                                  kernel_time_ms_per_step=round(total_ms / a.steps, 3),
                                  step_tflops=round(step_flops / (total_ms / a.steps * 1e-3) / 1e12, 2),
                                  note="hipEvent pair around every launch; `value` is from the clean pass before it"),
               scaling_measured=("N>1 not measured in this run" if world == 1 else "this line"))
    if not a.no_cpu_baseline and a.workload == "embed_detect" and world == 1 and not f16:      # the CPU leg: rank 0 at N = 1 only
        cb, wm_ref, mp_ref = cpu_baseline(cfgG, cfgD, sdG, sdD, x_np, msg_np, min(a.cpu_clips, B))
        n = wm_ref.shape[0]
        out["cpu_baseline"] = cb
        out["parity"] = dict(
            clips_checked=n,
            wm_max_abs_err=float(np.abs(wm[:n].cpu().numpy() - wm_ref).max()),
            mean_prob_max_abs_err=float(np.abs(mp[:n].cpu().numpy() - mp_ref).max()),
            ber_vs_oracle=float(((mp_ref >= 0.5) != bits[:n].cpu().numpy()).mean()))
    if not f16 and a.workload == "embed_detect":
        # The f16-operand / f32-accumulate mode of the same step (BASELINE configs[1] words the step "bf16"), so that the driver's record of
        # the DEFAULT run holds it: the same K steps, wm and bits against this run's exact outputs on the whole batch, and -- when the CPU
        # leg ran -- against the oracle on its sample.  A throughput mode beside the exact path: `value` above is exact f32.
        step("f16")
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(a.steps):
            wm16, mp16, bits16 = step("f16")
        torch.cuda.synchronize(); e16 = time.perf_counter() - t0
        rp = dict(dtype="f16 operands / f32 accumulate (csrc/wv_h16.hip): activations and weights cross HBM as f16, sums, stencils, ELU, FiLM, tanh in f32",
                  metric=metric.replace(",", " (f16-operand mode),", 1), value=round(world * B * a.steps / e16, 2), unit="clips/s",
                  ms_per_step=round(e16 / a.steps * 1e3, 3), speedup_vs_exact=round(elapsed / e16, 3),
                  wm_max_abs_diff_vs_exact=float((wm16 - wm).abs().max()), wm_bar=1e-4,
                  mean_prob_max_abs_diff_vs_exact=float((mp16 - mp).abs().max()),
                  bits_compared=int(bits.numel()), bits_differ_vs_exact=int((bits16 != bits).sum()),
                  note="rank 0's shard; same inputs and weights as the exact pass above; `python bench.py --precision f16` prints this mode's own line with its roofline")
        if "parity" in out:
            n = out["parity"]["clips_checked"]
            rp.update(clips_checked_vs_oracle=n, wm_max_abs_err_vs_oracle=float(np.abs(wm16[:n].cpu().numpy() - wm_ref).max()),
                      mean_prob_max_abs_err_vs_oracle=float(np.abs(mp16[:n].cpu().numpy() - mp_ref).max()),
                      ber_vs_oracle=float(((mp_ref >= 0.5) != bits16[:n].cpu().numpy()).mean()))
        out["reduced_precision"] = rp
    if f16:
        # The mode's own evidence: the exact-f32 path of this library on the same K steps (same input, same weights), the two modes'
        # outputs against each other on the whole batch, and against the torch-CPU port of the reference on a sample.
        step("f32")
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(a.steps):
            wm32, mp32, _ = step("f32")
        torch.cuda.synchronize(); e32 = time.perf_counter() - t0
        d = (mp - mp32).abs()
        margin = (mp32 - 0.5).abs()
        out["vs_f32_mode"] = dict(f32_clips_per_s=round(B * a.steps / e32, 2), f32_ms_per_step=round(e32 / a.steps * 1e3, 3),
                                  speedup=round(e32 / elapsed, 3), mean_prob_max_abs_diff=float(d.max()),
                                  bits_compared=int(mp.numel()), bits_differ=int(((mp >= 0.5) != (mp32 >= 0.5)).sum()),
                                  min_margin_f32=float(margin.min()),
                                  bits_with_margin_above_4x_diff=int((margin > 4 * float(d.max())).sum()))
        if a.workload != "detector_stress":
            out["vs_f32_mode"].update(wm_max_abs_diff=float((wm - wm32).abs().max()), wm_bar=1e-4)
            if not a.no_cpu_baseline and world == 1 and a.workload == "embed_detect":
                cb, wm_ref, mp_ref = cpu_baseline(cfgG, cfgD, sdG, sdD, x_np, msg_np, min(a.cpu_clips, B))
                n = wm_ref.shape[0]
                out["cpu_baseline"] = cb
                out["parity"] = dict(clips_checked=n, wm_max_abs_err=float(np.abs(wm[:n].cpu().numpy() - wm_ref).max()), wm_bar=1e-4,
                                     mean_prob_max_abs_err=float(np.abs(mp[:n].cpu().numpy() - mp_ref).max()),
                                     ber_vs_oracle=float(((mp_ref >= 0.5) != bits[:n].cpu().numpy()).mean()),
                                     note="torch-CPU port of the reference on the first clips of the batch; the reference-golden checks are tests/test_gpu_h16.py")
        elif not a.no_cpu_baseline and world == 1:
            from oracle import wv_oracle_torch as OT
            n = min(a.cpu_clips, B)
            torch.set_num_threads(min(os.cpu_count() or 1, 16))
            mp_ref = OT.mean_probabilities(OT.detector_logits(OT.Net(cfgD, sdD), x_np[:n])).numpy()
            got = mp[:n].cpu().numpy()
            err = float(np.abs(got - mp_ref).max())
            mg = np.abs(mp_ref - 0.5)
            out["parity"] = dict(clips_checked=n, mean_prob_max_abs_err=err, ber_vs_oracle=float(((mp_ref >= 0.5) != (got >= 0.5)).mean()),
                                 bits_decidable_at_4x_err=int((mg > 4 * err).sum()), bits=int(mg.size),
                                 note="torch-CPU port of the reference detector on the first clips of the batch; the golden-fixture checks "
                                      "(reference outputs, incl. the narrow-margin detector) are tests/test_gpu_h16.py")
    print(json.dumps(out), flush=True)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
