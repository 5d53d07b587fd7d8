"""Multi-GPU layout of the benchmark: one process per GPU, launched by torch.distributed.run.

The reference is single-GPU and this round shards nothing inside a superstep ("replicas only",
DESIGN.md section 7): every rank advances an independent replica of the workload that differs only in its KMC
random stream.  The only collective is the max-over-ranks of the timed region (RCCL on GPUs, gloo in the CPU tests).
"""
import os

import torch
import torch.distributed as dist


def init(backend=None):
    """Returns (rank, world, local_rank).  Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* from the environment.
    On GPUs the rank's device is selected BEFORE the process group is created (RCCL binds to the current device)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
    if backend == "nccl" and torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local_rank


def replica_kmc_seed(base_seed: int, rank: int) -> int:
    """KMC stream of a replica (rank 0 keeps the reference's rnd_seed_kmc)."""
    return base_seed + rank


def barrier():
    if dist.is_initialized():
        if dist.get_backend() == "nccl":
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()


def max_over_ranks(seconds: float, device="cpu") -> float:
    if not dist.is_initialized():
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def aggregate_rate(steps_per_rank: int, world: int, elapsed_max: float) -> float:
    """Whole-job throughput: every replica completed steps_per_rank steps within the slowest rank's time."""
    return world * steps_per_rank / elapsed_max


def finalize():
    if dist.is_initialized():
        dist.destroy_process_group()
