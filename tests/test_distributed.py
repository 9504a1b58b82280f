"""CPU, world_size 2 over gloo: the N>1 path of bench.py -- contiguous utterance shards, one weight-blob broadcast,
no data-path collective, concatenated shard results == unsharded result."""
import importlib
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    D = importlib.import_module("speech-backbones_amd.dist")
    r, w, _ = D.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    # weight blob: rank 0 "packs", the others receive
    blob = torch.arange(1000, dtype=torch.uint8) if rank == 0 else torch.zeros(1000, dtype=torch.uint8)
    D.broadcast_packed(blob, src=0)
    ok_blob = bool(torch.equal(blob, torch.arange(1000, dtype=torch.uint8)))
    # shard a batch of 5 utterances; per-sample "work" stands in for the sampler (it never mixes batch entries)
    g = torch.Generator().manual_seed(0)
    batch = torch.randn(5, 80, 8, generator=g)
    lo, hi = D.shard_bounds(5, world, rank)
    local = batch[lo:hi] * 2.0 + 1.0
    outs = D.gather_outputs(local, dst=0)
    t = D.max_over_ranks(1.0 + rank, torch.device("cpu"))
    D.barrier()
    if rank == 0:
        q.put((ok_blob, bool(torch.equal(torch.cat(outs, 0), batch * 2.0 + 1.0)), t, (lo, hi)))
    else:
        q.put((ok_blob, True, t, (lo, hi)))
    dist.destroy_process_group()


def test_two_rank_shard_and_broadcast():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[0] and r[1] for r in res)
    assert all(abs(r[2] - 2.0) < 1e-9 for r in res)           # max over ranks
    assert sorted(r[3] for r in res) == [(0, 3), (3, 5)]


def _gpu_worker(rank, world, port, q):
    """The real N>1 path of bench.py on one visible GPU: rank 0 packs, the blob is broadcast, every rank samples its
    contiguous utterance shard with the HIP kernels; rank 0 checks concatenation == unsharded run, bit for bit."""
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from oracle import gradtts_oracle as O
    S = importlib.import_module("speech-backbones_amd")
    D = importlib.import_module("speech-backbones_amd.dist")
    D.init_from_env(backend="gloo")
    dev = torch.device("cuda:0")
    plan = S.Plan()
    if rank == 0:
        blob = plan.pack(O.make_estimator_state(seed=0), dev)
    else:
        blob = torch.zeros(plan.packed_bytes(), dtype=torch.uint8, device=dev)
    D.broadcast_packed(blob, src=0)
    inp = O.make_inputs(5, 64, seed=3)
    lo, hi = D.shard_bounds(5, world, rank)
    z, m, mu = (inp[k][lo:hi].contiguous().to(dev) for k in ("z", "mask", "mu"))
    local = plan.reverse_diffusion(blob, z, m, mu, 4).cpu()
    outs = D.gather_outputs(local, dst=0)
    t = D.max_over_ranks(1.0 + rank, torch.device("cpu"))
    ok = True
    if rank == 0:
        whole = plan.reverse_diffusion(blob, inp["z"].to(dev), inp["mask"].to(dev), inp["mu"].to(dev), 4).cpu()
        ref = O.reverse_diffusion(O.make_estimator_state(seed=0), inp["z"], inp["mask"], inp["mu"], 4)
        ok = bool(torch.equal(torch.cat(outs, 0), whole)) and float((whole - ref).abs().max() / ref.abs().max()) < 1e-4
    D.barrier()
    q.put((ok, t, (lo, hi)))
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_ranks_run_the_real_sampler_on_their_shards():
    assert torch.cuda.is_available()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert all(r[0] for r in res)
    assert all(abs(r[1] - 2.0) < 1e-9 for r in res)
    assert sorted(r[2] for r in res) == [(0, 3), (3, 5)]


def test_shard_helpers():
    D = importlib.import_module("speech-backbones_amd.dist")
    for n, w in [(16, 8), (5, 2), (3, 4), (128, 8)]:
        b = [D.shard_bounds(n, w, r) for r in range(w)]
        assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
        assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1
    fb = D.shard_by_frames([100, 10, 10, 10, 100, 10], 2)
    assert fb[0][0] == 0 and fb[-1][1] == 6 and fb[0][1] == fb[1][0]


def _bcast_worker(rank, world, uid_q, q):
    """gtts_bcast_weights (the C ABI's RCCL entry point for non-torch hosts) carrying real bytes through an ncclComm_t created
    with ctypes on the RCCL the process already has loaded.  world == 2 (two visible GPUs): rank 1 starts from zeros and must end
    up with rank 0's bytes -- a broadcast that moved nothing fails.  world == 1 (the one-GPU test box): the call path only."""
    import ctypes
    import sys
    sys.path.insert(0, ROOT)
    S = importlib.import_module("speech-backbones_amd")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    torch.zeros(1, device=dev)
    path = None
    for line in open("/proc/self/maps"):
        if "librccl" in line:
            path = line.split()[-1]
            break
    rccl = ctypes.CDLL(path or "librccl.so", mode=ctypes.RTLD_GLOBAL)

    class UniqueId(ctypes.Structure):
        _fields_ = [("internal", ctypes.c_char * 128)]

    uid = UniqueId()
    rccl.ncclGetUniqueId.argtypes = [ctypes.POINTER(UniqueId)]
    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
    rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
    if rank == 0:
        assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
        for _ in range(world - 1):
            uid_q.put(bytes(ctypes.string_at(ctypes.byref(uid), 128)))
    else:
        ctypes.memmove(ctypes.byref(uid), uid_q.get(timeout=120), 128)
    comm = ctypes.c_void_p()
    assert rccl.ncclCommInitRank(ctypes.byref(comm), world, uid, rank) == 0
    g = torch.Generator().manual_seed(0)
    want = torch.randint(0, 256, (3 * 1024 * 1024 + 17,), dtype=torch.uint8, generator=g).to(dev)
    blob = want.clone() if rank == 0 else torch.zeros_like(want)
    L = S._lib.lib()
    rc = L.gtts_bcast_weights(ctypes.c_void_p(blob.data_ptr()), blob.numel(), 0, comm, S._lib._stream())
    torch.cuda.synchronize()
    ok = rc == 0 and bool(torch.equal(blob, want))
    err = L.gtts_last_error().decode() if rc else ""
    rccl.ncclCommDestroy(comm)
    q.put((ok, rc, err, path, rank))


@pytest.mark.gpu
def test_bcast_weights_carries_bytes_through_rccl():
    world = 2 if torch.cuda.device_count() >= 2 else 1
    ctx = mp.get_context("spawn")
    q, uid_q = ctx.Queue(), ctx.Queue()
    procs = [ctx.Process(target=_bcast_worker, args=(r, world, uid_q, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    print("RCCL library:", res[0][3], " ranks:", world)
    for ok, rc, err, path, rank in res:
        assert ok, (rank, rc, err)


def _run_bench(*flags, timeout=300):
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(flags), env=env, capture_output=True, text=True,
                          timeout=timeout)


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment is a one-command path (round-5 review item 8): it re-executes
    itself under torch.distributed.run with N ranks on 127.0.0.1.  Rehearsed here on CPU over gloo (--dry-run: launcher, rendezvous,
    blob broadcast, shards, barrier-bracketed timing, max over ranks) -- the line reports n_gpus = N and one time per rank."""
    import json
    r = _run_bench("--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                       # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["valid"] is False and d["data"] == "dry-run"
    assert len(d["config"]["per_rank_ms_per_step"]) == 2 and d["config"]["weight_bcast_ms"] >= 0.0
    assert d["ms_per_step"] >= max(d["config"]["per_rank_ms_per_step"]) - 1e-3      # the line's time is the max over ranks
    r = _run_bench("--gpus", "2", "--dry-run", "--total-batch", "6", "--steps", "1", "--warmup", "0")
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["scaling"] == "strong" and d["config"]["per_gpu_batch"] == 3


@pytest.mark.skipif(torch.cuda.is_available() and torch.cuda.device_count() >= 2, reason="needs a box with fewer than 2 GPUs")
def test_bench_fails_loudly_without_enough_devices():
    r = _run_bench("--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-extras", timeout=600)
    assert r.returncode != 0
    assert "--gpus 2 but only" in (r.stderr + r.stdout)
