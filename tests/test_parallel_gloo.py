"""N > 1 path of bench.py on CPU: two gloo ranks, replica seeds, barrier + max-over-ranks timing, aggregate rate."""
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
