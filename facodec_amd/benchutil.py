"""Timing scaffold shared by bench.py and the gloo tests: one process per GPU, clips sharded across
ranks with no data-path collective (clips are independent units); only the
timing itself uses collectives (barrier + MAX all-reduce of the elapsed time)."""
import os
import time

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment (torch.distributed.run)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # a single rank gets a process group too when FAC_FORCE_ALLREDUCE=1 asks for the collective path on a one-GPU box
    if (world > 1 or (os.environ.get("FAC_FORCE_ALLREDUCE") == "1" and "RANK" in os.environ)) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # "nccl" is RCCL on ROCm; FAC_DIST_BACKEND=gloo lets two ranks share one GPU in a smoke test
            backend = os.environ.get("FAC_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_clips(n_total, world, rank):
    """Contiguous block partition of clip indices [0, n_total) -> this rank's range."""
    per = (n_total + world - 1) // world
    lo = min(n_total, rank * per)
    return lo, min(n_total, lo + per)


def timed_steps(step_fn, steps, warmup, sync_fn, device=None):
    """W untimed warm-up steps, then exactly K steps bracketed by barrier + device sync on both
    sides; returns the MAX elapsed seconds over ranks."""
    for _ in range(warmup):
        step_fn()
    sync_fn()
    if dist.is_initialized():
        dist.barrier()
    sync_fn()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    sync_fn()
    if dist.is_initialized():
        dist.barrier()
    sync_fn()
    dt = time.perf_counter() - t0
    if dist.is_initialized():
        t = torch.tensor([dt], dtype=torch.float64, device=device if device is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def aggregate_units(units_this_rank, device=None):
    """Whole-job units = sum over ranks."""
    if not dist.is_initialized():
        return float(units_this_rank)
    t = torch.tensor([float(units_this_rank)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
