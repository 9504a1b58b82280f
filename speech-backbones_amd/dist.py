"""Multi-GPU plumbing: one process per GPU, utterances sharded across ranks, ONE collective (the packed-weight
broadcast over RCCL/xGMI at load time) and none per diffusion step -- no operation of the sampling path mixes batch
entries.  `torch.distributed` is used as the RCCL binding (backend "nccl" == RCCL on ROCm);
on CPU the same code runs over gloo for the tests.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_bounds(n_items, world, rank):
    """Contiguous split of n_items utterances over `world` ranks (first ranks get the remainder)."""
    base, rem = divmod(int(n_items), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_by_frames(lengths, world):
    """Contiguous split balancing the number of valid frames per rank (ragged batches).  Returns [(lo, hi)]."""
    lengths = [int(v) for v in lengths]
    total = sum(lengths)
    bounds, lo, acc = [], 0, 0
    for r in range(world):
        target = total * (r + 1) / world
        hi = lo
        while hi < len(lengths) - (world - 1 - r) and (acc < target or hi == lo):
            acc += lengths[hi]
            hi += 1
        if r == world - 1:
            hi = len(lengths)
        bounds.append((lo, hi))
        lo = hi
    return bounds


def broadcast_packed(blob, src=0):
    """Rank `src` packed the weights; everyone else receives the blob (uint8 tensor, same size everywhere)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        if blob.is_cuda and dist.get_backend() == "gloo":      # CPU test rigs: stage through the host
            host = blob.cpu()
            dist.broadcast(host, src=src)
            blob.copy_(host)
        else:
            dist.broadcast(blob, src=src)                      # RCCL (ncclBroadcast) over xGMI
    return blob


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(value, device):
    """Max of a python float over all ranks (timing aggregation for bench.py)."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_floats(value, device):
    """[value of rank 0, value of rank 1, ...] on every rank (per-rank timings in bench.py's line)."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return [float(value)]
    mine = torch.tensor([float(value)], dtype=torch.float64, device=device)
    outs = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(outs, mine)
    return [float(t.item()) for t in outs]


def gather_outputs(local_out, dst=0):
    """Optional: collect per-rank output mels on rank `dst` (not part of any timed region)."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return [local_out]
    world = dist.get_world_size()
    sizes = [torch.zeros(1, dtype=torch.long, device=local_out.device) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([local_out.shape[0]], dtype=torch.long, device=local_out.device))
    outs = []
    for r in range(world):
        buf = local_out if r == dist.get_rank() else torch.empty((int(sizes[r]),) + tuple(local_out.shape[1:]),
                                                                  dtype=local_out.dtype, device=local_out.device)
        dist.broadcast(buf, src=r)
        outs.append(buf)
    return outs if dist.get_rank() == dst else None
