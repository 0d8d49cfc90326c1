"""Batch-wise slice sharding over the GPUs of one node (one process per GPU, torch.distributed;
backend "nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests).

The path shards naturally: under per-slice semantics a slice never interacts with
another one, so ranks work on disjoint contiguous slice ranges with NO data-path collective; the only
exchange is ONE all-gather of the finished [B/N,1,512,512] blocks at the very end.  Noise is keyed by
the GLOBAL slice id (diffusion.NoiseSource), so results do not depend on the number of ranks.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun contract). Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # IPDM_DIST_BACKEND=gloo: plumbing tests of the N>1 path on a box with fewer GPUs than ranks
            backend = os.environ.get("IPDM_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(n_slices, rank, world):
    """Contiguous partition [lo, hi) of n_slices over `world` ranks (remainder to the first ranks)."""
    base, rem = divmod(n_slices, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_slices(local_out, n_slices, rank, world, force_collective=False):
    """The single collective of the path: gathers every rank's [b_r, ...] block into [n_slices, ...]
    (identical on all ranks).  Equal shards use one all_gather_into_tensor; ragged ones pad to the
    largest shard.  A single rank returns its block untouched unless `force_collective` asks for the
    collective anyway (the world-size-1 RCCL test: the same code path the N-rank job takes)."""
    if world == 1 and not (force_collective and dist.is_initialized()):
        return local_out
    sizes = [shard_range(n_slices, r, world) for r in range(world)]
    counts = [hi - lo for lo, hi in sizes]
    mx = max(counts)
    x = local_out.contiguous()
    if x.shape[0] < mx:
        pad = torch.zeros((mx - x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        x = torch.cat([x, pad], 0)
    if x.is_cuda and dist.get_backend() == "gloo":
        # plumbing runs of the N>1 path on a box with fewer GPUs than ranks (IPDM_DIST_BACKEND=gloo): gloo gathers
        # through host memory; the production backend ("nccl" = RCCL over xGMI) gathers device buffers directly
        host = torch.empty((world * mx,) + tuple(x.shape[1:]), dtype=x.dtype)
        dist.all_gather_into_tensor(host, x.cpu())
        out = host.to(x.device)
    else:
        out = torch.empty((world * mx,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(out, x)
    if all(c == mx for c in counts):
        return out
    return torch.cat([out[r * mx: r * mx + counts[r]] for r in range(world)], 0)


def barrier():
    if dist.is_initialized():
        dist.barrier()


def describe():
    """What the bench line records about the job's ranks: {"world": N, "backend": "nccl"|"gloo"|None}."""
    if not dist.is_initialized():
        return {"world": 1, "backend": None}
    return {"world": dist.get_world_size(), "backend": dist.get_backend()}


def max_over_ranks(value, device):
    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_over_ranks(value, device):
    """Every rank's scalar, in rank order (identical on all ranks): bench.py reports the per-rank step times beside their
    maximum, so that a scaling record shows load imbalance and not just its effect."""
    if not dist.is_initialized():
        return [float(value)]
    world = dist.get_world_size()
    dev = "cpu" if dist.get_backend() == "gloo" else device
    mine = torch.tensor([value], dtype=torch.float64, device=dev)
    out = torch.empty(world, dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(out, mine)
    return [float(v) for v in out.cpu()]
