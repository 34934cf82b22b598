"""The N>1 path on CPU: two gloo ranks shard a clip batch, run embed -> detect on their shards
(the numpy oracle on shrunk nets stands in for the GPU nets: the code under test is the
sharding + gather of waveverify_amd/parallel.py, which has no data-path collective), and the
gathered result must equal the unsharded run."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import wv_oracle as O
from waveverify_amd import parallel
from waveverify_amd.config import default_config
from waveverify_amd.init import random_state_dict, synthetic_clips

SMALL = dict(channels_enc=8, dimension=16, strides=[2, 2], n_fft_base=16)


def _nets():
    cg = default_config("generator", channels_dec=8, n_residual_dec=2, **SMALL)
    cd = default_config("detector", output_dim=8, nbits=16, **SMALL)
    G, D = O._Net(cg, random_state_dict(cg, 7)), O._Net(cd, random_state_dict(cd, 7))
    embed = lambda x, m: torch.from_numpy(O.embed(cg, G, x.numpy(), m.numpy()))
    detect = lambda x: torch.from_numpy(O.mean_probabilities(O.detector_forward(cd, D, x.numpy())))
    return embed, detect


def _worker(rank, world, port, B, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    try:
        x, msg = synthetic_clips(B, 400, seed=11)
        embed, detect = _nets()
        wm, mp_local, mp_all = parallel.embed_detect_sharded(
            embed, detect, torch.from_numpy(x), torch.from_numpy(msg), rank, world)
        lo, hi = parallel.shard_bounds(B, rank, world)
        assert wm.shape[0] == hi - lo == mp_local.shape[0]
        np.save(os.path.join(out_dir, f"mp_all_{rank}.npy"), mp_all.numpy())
        np.save(os.path.join(out_dir, f"wm_{rank}.npy"), wm.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("B", [5, 4])
def test_two_rank_sharding_matches_single_process(tmp_path, B):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, B, str(tmp_path)), nprocs=2, join=True)
    x, msg = synthetic_clips(B, 400, seed=11)
    embed, detect = _nets()
    wm_ref = embed(torch.from_numpy(x), torch.from_numpy(msg))
    mp_ref = detect(wm_ref).numpy()
    a, b = (np.load(tmp_path / f"mp_all_{r}.npy") for r in (0, 1))
    assert a.shape == (B, 16) and np.array_equal(a, b)            # every rank holds the full result
    # the numpy stand-in is only reproducible to BLAS blocking (which depends on the batch shape);
    # the GPU nets are bit-identical per clip (tests/test_gpu_nets.py::test_batch_independence...)
    assert np.abs(a - mp_ref).max() <= 1e-6
    assert np.array_equal(a >= 0.5, mp_ref >= 0.5)
    wm = np.concatenate([np.load(tmp_path / f"wm_{r}.npy") for r in (0, 1)])
    assert wm.shape == wm_ref.shape and np.abs(wm - wm_ref.numpy()).max() <= 1e-6


# ---- gradient-bucket all-reduce (BASELINE configs[2]: its microbench is bench.py --workload grad_allreduce)
def test_bucket_plan_covers_every_parameter_once():
    from waveverify_amd import params
    from waveverify_amd.config import default_config
    numels = [int(np.prod(shape)) for _, shape, _ in params.param_specs(default_config("generator"))]
    buckets = parallel.plan_buckets(numels, bucket_bytes=8 << 20)
    flat = [i for b in buckets for i in b]
    assert sorted(flat) == list(range(len(numels)))                 # a partition
    assert flat == list(reversed(range(len(numels))))               # in gradient-ready order
    sizes = [sum(numels[i] for i in b) * 4 for b in buckets]
    assert all(s >= 8 << 20 for s in sizes[:-1]) and len(buckets) >= 3
    assert parallel.plan_buckets([10, 10 ** 8, 10], bucket_bytes=1 << 20) == [[2, 1], [0]]


def _ar_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(100 + rank)
        shapes = [(7, 3), (1000,), (33, 5, 2), (1,), (257, 64)]
        grads = [torch.randn(*s, generator=g) for s in shapes]
        buckets = parallel.plan_buckets([t.numel() for t in grads], bucket_bytes=4096)
        n = parallel.allreduce_mean_(grads, buckets)
        assert n == len(buckets) >= 2
        torch.save(grads, os.path.join(out_dir, f"g{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_two_rank_bucketed_allreduce_is_the_mean(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_ar_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    shapes = [(7, 3), (1000,), (33, 5, 2), (1,), (257, 64)]
    want = None
    for r in (0, 1):
        g = torch.Generator().manual_seed(100 + r)
        gr = [torch.randn(*s, generator=g) for s in shapes]
        want = gr if want is None else [a + b for a, b in zip(want, gr)]
    want = [w / 2 for w in want]
    for r in (0, 1):
        got = torch.load(tmp_path / f"g{r}.pt", weights_only=True)
        for a, b in zip(got, want):
            assert torch.allclose(a, b, atol=1e-6)


def _flat_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from waveverify_amd.parallel import allreduce_mean_flat_
    arena = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    n = allreduce_mean_flat_(arena, bucket_bytes=4 * 300)          # 300-float buckets -> 4 collectives, last one ragged
    q.put((rank, n, arena.clone()))
    dist.destroy_process_group()


def test_flat_arena_allreduce_two_ranks():
    """The training slices keep gradients in one flat arena: buckets are slices, reduced in place from the end."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + 7
    ps = [ctx.Process(target=_flat_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(60)
    want = torch.arange(1000, dtype=torch.float32) * 1.5
    for rank, n, arena in res:
        assert n == 4 and torch.equal(arena, want)


def _bench(args, **env):
    import subprocess
    import sys
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=e, capture_output=True, text=True, timeout=300)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher: the parent starts two ranks as a child torch.distributed.run (gloo here: the
    test-only backend switch, no GPU), relays rank 0's single JSON line and returns the child's exit code."""
    import json
    r = _bench(["--gpus", "2", "--workload", "rendezvous"], WV_BENCH_BACKEND="gloo")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["backend"] == "gloo" and out["config"]["workload"] == "rendezvous"


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    r = _bench(["--gpus", "2", "--workload", "rendezvous"], WV_BENCH_BACKEND="gloo", WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    assert r.returncode != 0 and "--gpus 2 but WORLD_SIZE=3" in r.stderr
    r = _bench(["--gpus", "1", "--workload", "rendezvous"], WV_BENCH_BACKEND="gloo", WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    assert r.returncode != 0 and "--gpus 1 but WORLD_SIZE=2" in r.stderr


def test_bench_keeps_the_f16_mode_off_the_training_workloads():
    """`--precision f16` selects the f16-operand mode's OWN line for the inference workloads (embed_detect, longform, detector_stress; the
    default line's `value` stays exact f32 and carries the mode in a `reduced_precision` block); the training workloads refuse it, before
    any GPU call."""
    for wl in ("train_step", "grad_allreduce"):
        r = _bench(["--workload", wl, "--precision", "f16"])
        assert r.returncode != 0 and "training is exact f32" in r.stderr, (wl, r.stderr[-300:])
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")).read()
    assert 'out["reduced_precision"] = rp' in src and 'dtype="f16" if f16 else "f32"' in src


def _overlap_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(100 + rank)
        sizes = [7, 300, 1, 64, 1200, 5, 900, 33]                       # parameters of an arena, in forward order
        ranges, off = {}, 0
        for i, n in enumerate(sizes):
            ranges[f"p{i}"] = (off, off + n)
            off += n
        grads = torch.randn(off, generator=g)
        ref = grads.clone()
        n_ref = parallel.allreduce_mean_flat_(ref, bucket_bytes=4 * 500)         # post-backward version: 500-element buckets
        live = grads.clone()
        red = parallel.OverlappedFlatReducer(live, ranges, bucket_bytes=4 * 500)
        launched = []
        for keys in (["p7", "p6"], ["p5"], ["p3", "p4"], ["p2"], ["p0"]):       # backward order; p1 is never marked
            launched.append(red.mark(keys))
        n_live = red.wait()                                                      # flushes the buckets p1 holds back
        assert n_live == n_ref and sum(launched) >= 2 and sum(launched) < n_live
        assert torch.equal(live, ref)
        torch.save(live, os.path.join(out_dir, f"r{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_overlapped_reducer_equals_the_post_backward_all_reduce(tmp_path):
    """Buckets launched while 'backward' is still marking parameters give exactly the arena the post-backward all-reduce gives, on
    both ranks; buckets holding an unmarked parameter go out at wait()."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_overlap_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    assert torch.equal(a, b)


def test_overlapped_reducer_without_a_process_group_is_inert():
    g = torch.arange(10, dtype=torch.float32)
    red = parallel.OverlappedFlatReducer(g, {"a": (0, 4), "b": (4, 10)}, bucket_bytes=16)
    assert red.mark(["b"]) >= 1 and red.wait() == 0 and torch.equal(g, torch.arange(10, dtype=torch.float32))
