"""Multi-GPU layout of the benchmark: one process per GPU, launched by torch.distributed.run.

The reference is single-GPU.  Two ways to use N GPUs:

* replicas (weak scaling, the default of bench.py): every rank advances an independent replica of the workload that
  differs only in its KMC random stream; the only collective is the max-over-ranks of the timed region.
* sharded solve (strong scaling): all ranks advance the SAME simulation in lockstep and the segment stage of the
  current solve's A*p is dealt to the ranks, completed by one all-gather per CG iteration inside the library
  (csrc/comm.hip).  `attach_solver_comm` creates that communicator: RCCL over xGMI when the process group is nccl,
  the host-callback transport over gloo otherwise (rehearsal with several ranks on one GPU, CPU tests).
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


def from_rank0(value: float, device="cpu") -> float:
    """Rank 0's value on every rank.  Ranks that advance ONE simulation in lockstep must take every control decision that depends
    on a local clock (how many steps fit a time budget) from one rank: a rank that stops a step early leaves the others in a collective."""
    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.broadcast(t, src=0)
    return float(t.item())


def aggregate_rate(steps_per_rank: int, world: int, elapsed_max: float) -> float:
    """Whole-job throughput: every replica completed steps_per_rank steps within the slowest rank's time."""
    return world * steps_per_rank / elapsed_max


_solver_cb = None      # keeps the ctypes callback object alive while the library holds its address


def _gloo_allgather_cb(group):
    """dkmc_allgather_fn over a torch.distributed group with CPU tensors: all-gathers the pinned host buffer in place."""
    import ctypes
    import numpy as np
    from . import lib

    def cb(host_buf, bytes_per_rank, rank, nranks, _user):
        try:
            n = bytes_per_rank // 8
            arr = np.ctypeslib.as_array(ctypes.cast(host_buf, ctypes.POINTER(ctypes.c_double)), shape=(nranks * n,))
            full = torch.from_numpy(arr)
            chunks = [full[r * n:(r + 1) * n] for r in range(nranks)]
            dist.all_gather(chunks, chunks[rank].clone(), group=group)
            return 0
        except Exception as exc:              # an exception must not unwind through the C caller
            print("devicekmc_amd.parallel: all-gather callback failed: %r" % (exc,), flush=True)
            return 1
    return lib.ALLGATHER_FN(cb)


def attach_solver_comm(transport=None, group=None):
    """Attach the current-solve communicator of this rank (call after the device has been selected, on every rank).
    transport: "rccl", "host", or None = rccl if the default process group is nccl, host otherwise.
    A world of one attaches nothing unless a transport is named explicitly (tests of the 1-rank RCCL path)."""
    global _solver_cb
    from . import lib
    L = lib.load()
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if transport is None:
        if world == 1:
            return "none"
        transport = "rccl" if dist.get_backend(group) == "nccl" else "host"
    if transport == "rccl":
        import ctypes
        idbuf = ctypes.create_string_buffer(128)
        if rank == 0:
            lib.check(L.dkmc_comm_unique_id(idbuf))
        if world > 1:
            dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
            t = torch.frombuffer(bytearray(idbuf.raw), dtype=torch.uint8).to(dev)
            dist.broadcast(t, src=0, group=group)
            idbuf = ctypes.create_string_buffer(bytes(t.cpu().numpy().tobytes()), 128)
        lib.check(L.dkmc_comm_init_rccl(world, rank, idbuf))
    elif transport == "host":
        _solver_cb = _gloo_allgather_cb(group) if world > 1 else lib.ALLGATHER_FN(lambda *a: 0)
        lib.check(L.dkmc_comm_init_host(world, rank, _solver_cb, None))
    else:
        raise ValueError("unknown transport %r" % (transport,))
    return transport


def attach_peer_exchange(slot_doubles, group=None):
    """One-shot peer-write exchange of the sharded block-CG beside the attached communicator (csrc/comm.hip): every rank exports its
    exchange buffer (hipIpc), the handles travel through the process group, every rank maps every peer's buffer.  slot_doubles: capacity of
    a slot, >= 16 |S| + 2 of the solves to come (a solve whose slot does not fit falls back to the communicator's all-gather on every rank
    alike).  All ranks must sit on one node.  Returns True on every rank, or False on every rank if any rank could not map a peer -- or if
    the ranks sit on different devices, which the library refuses (validated on one device only; DKMC_PEER_CROSS_DEVICE=1 overrides)."""
    import ctypes
    from . import lib
    L = lib.load()
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    mine = ctypes.create_string_buffer(192)          # two hipIpc handles + the rank's device identity (csrc/comm.hip)
    ok = L.dkmc_comm_peer_prepare(int(slot_doubles), mine) == 0
    on_gpu = dist.is_initialized() and dist.get_backend(group) == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")
    t = torch.frombuffer(bytearray(mine.raw), dtype=torch.uint8).to(dev)
    parts = [torch.empty_like(t) for _ in range(world)]
    if world > 1:
        dist.all_gather(parts, t, group=group)
    else:
        parts = [t]
    allh = b"".join(bytes(x.cpu().numpy().tobytes()) for x in parts)
    ok = ok and L.dkmc_comm_peer_attach(ctypes.create_string_buffer(allh, len(allh))) == 0
    flag = torch.tensor([0 if ok else 1], dtype=torch.int32, device=dev)
    if world > 1:
        dist.all_reduce(flag, group=group)
    if int(flag.item()) != 0:
        L.dkmc_clear_error()
        L.dkmc_comm_peer_detach()
        return False
    return True


def peer_exchange_info():
    import ctypes
    from . import lib
    r, s, n, us = ctypes.c_int(0), ctypes.c_longlong(0), ctypes.c_longlong(0), ctypes.c_double(0.0)
    lib.load().dkmc_comm_peer_info(ctypes.byref(r), ctypes.byref(s), ctypes.byref(n), ctypes.byref(us))
    return {"ready": bool(r.value), "slot_doubles": s.value, "exchanges": n.value, "mean_us": us.value}


def detach_solver_comm():
    global _solver_cb
    from . import lib
    lib.check(lib.load().dkmc_comm_destroy())
    _solver_cb = None


def finalize():
    if dist.is_initialized():
        dist.destroy_process_group()
