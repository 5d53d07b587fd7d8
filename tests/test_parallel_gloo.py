"""N > 1 paths on CPU with two gloo ranks: the replica mode of bench.py (seeds, barrier + max-over-ranks timing, aggregate
rate) and the host half of the sharded solve's exchange step (C ABI -> ctypes callback -> gloo all-gather)."""
import os
import socket

import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import numpy as np
    from devicekmc_amd import parallel, rng
    r, w, lr = parallel.init("gloo")
    assert (r, w) == (rank, world)
    seed = parallel.replica_kmc_seed(1, r)
    u = rng.StdMT19937(seed).uniform_batch(4)
    parallel.barrier()
    elapsed = parallel.max_over_ranks(1.0 + rank)            # rank 1 is the slow one
    rate = parallel.aggregate_rate(10, w, elapsed)
    assert parallel.from_rank0(7.0 + rank) == 7.0            # lockstep ranks take clock-dependent decisions from rank 0
    q.put((rank, seed, float(u[0]), elapsed, rate))
    parallel.finalize()


def test_two_rank_replicas():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs: p.start()
    out = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs: p.join(60); assert p.exitcode == 0
    (r0, s0, u0, e0, v0), (r1, s1, u1, e1, v1) = out
    assert (s0, s1) == (1, 2) and u0 != u1              # replicas follow different KMC streams; rank 0 = reference seed
    assert e0 == e1 == 2.0                              # max over ranks
    assert v0 == v1 == 2 * 10 / 2.0                     # aggregate steps/s over both replicas


def _worker_comm(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import ctypes as C
    import numpy as np
    from devicekmc_amd import lib, parallel
    parallel.init("gloo")
    L = lib.load()
    assert parallel.attach_solver_comm() == "host"           # gloo group -> host-callback transport
    n, r, t = C.c_int(), C.c_int(), C.c_int()
    L.dkmc_comm_info(C.byref(n), C.byref(r), C.byref(t))
    attached = (n.value, r.value, t.value)
    count = 1000
    buf = np.full(world * count, -1.0)
    buf[rank * count:(rank + 1) * count] = np.arange(count) + 1000.0 * rank       # own chunk in place
    lib.check(L.dkmc_comm_allgather_host(buf.ctypes.data_as(C.c_void_p), count))
    parallel.detach_solver_comm()
    L.dkmc_comm_info(C.byref(n), None, C.byref(t))
    q.put((rank, attached, buf, t.value))
    parallel.finalize()


def test_solver_comm_host_transport():
    import numpy as np
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_comm, args=(r, world, port, q)) for r in range(world)]
    for p in procs: p.start()
    out = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
    for p in procs: p.join(60); assert p.exitcode == 0
    want = np.concatenate([np.arange(1000) + 1000.0 * r for r in range(world)])
    for rank, (attached, buf, transport_after) in ((o[0], o[1:]) for o in out):
        assert attached == (world, rank, 2)                   # nranks, rank, DKMC_COMM_HOST while attached
        assert np.array_equal(buf, want)                      # every rank holds every chunk
        assert transport_after == 0                           # detached
