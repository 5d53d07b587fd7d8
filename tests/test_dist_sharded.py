"""The sharded current solve (csrc/comm.hip): N ranks advance one simulation in lockstep, the segment stage of A*p is dealt
to the ranks and completed by one all-gather per CG iteration.  The contract is bit-identity with the single-GPU path.

* two ranks sharing cuda:0 over the host-callback transport (gloo) -- RCCL refuses two ranks on one device, and the test
  box has one GPU; this covers the partitioning, the lockstep launch plan and the exchange placement;
* the RCCL transport itself with a communicator of one rank (dlopen, ncclCommInitRank, in-place ncclAllGather on the
  engine's stream).
(File name: sorts before the other GPU tests so that the ranks are spawned from a parent that has not touched the GPU.)
"""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

Vd = 5.0
NSTEPS = 3


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _supersteps(nsteps, seed=1, tiles=0):
    """nsteps supersteps of the 2.5 nm device from a fresh state; returns everything a caller of the path can observe.
    tiles=0: the single-GPU solve reads every stored entry (the arithmetic the sharded solve reproduces bit for bit)."""
    import torch
    from devicekmc_amd import host, lib, params, structure
    lib.load().dkmc_set_symmetric_tiles(tiles)
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    s = structure.load_structure(os.path.join(g, "device_2.5nm.npz"))
    p = params.KMCParameters(); p.solve_heating_global = True; p.rnd_seed_kmc = seed
    dev = host.Device(s, p)
    sim = host.KMCProcess(dev, p.freq)
    gb = dev.make_gpubuf("cuda:0")
    dev.setLaplacePotential(gb, p, Vd); gb.sync_HostToGPU(dev)
    trace, iters = [], []
    for k in range(nsteps):
        dev.updateCharge(gb); dev.updatePotential(gb, p, Vd, k)
        _, dt = sim.executeKMCStep(gb, dev)
        dev.updatePower(gb, p, Vd); dev.updateTemperature(gb, p, dt)
        torch.cuda.synchronize()
        trace.append((dt, dev.imacro, dev.T_bg)); iters.append(host.get_stats()["cg_iters_X"])
    fields = {n: gb.t[n].cpu().numpy().copy() for n in ("site_power", "site_potential_boundary", "site_potential_charge",
                                                        "site_charge", "site_element", "atom_virtual_potentials")}
    return trace, iters, fields, dict(host.get_stats())


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    from devicekmc_amd import parallel
    parallel.init("gloo")
    torch.cuda.set_device(0)
    ref = _supersteps(NSTEPS) if rank == 0 else None           # single-GPU path, no communicator
    ref_tiles = _supersteps(NSTEPS, tiles=1) if rank == 0 else None    # default single-GPU arithmetic (symmetric tiles)
    parallel.barrier()
    assert parallel.attach_solver_comm() == "host"
    got = _supersteps(NSTEPS)
    parallel.detach_solver_comm()
    parallel.barrier()
    q.put((rank, ref, got, ref_tiles))
    parallel.finalize()


def test_two_ranks_lockstep_bit_identical():
    import __graft_entry__ as g
    g.build()
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs: p.start()
    out = sorted((q.get(timeout=600) for _ in range(world)), key=lambda t: t[0])
    for p in procs: p.join(120); assert p.exitcode == 0
    (_, ref, got0, ref_tiles), (_, _, got1, _) = out
    rtrace, riters, rfields, _ = ref
    for rank, (trace, iters, fields, st) in enumerate((got0, got1)):
        assert trace == rtrace, (rank, trace, rtrace)                 # dt, I_macro, T_bg of every step: exact
        assert iters == riters                                        # same CG iteration counts
        for n in rfields:
            assert np.array_equal(fields[n], rfields[n]), (rank, n)   # every field a caller can read back: bit-identical
        assert st["comm_ranks"] == 2 and st["comm_count_per_rank"] % 2 == 0
    # against the default single-GPU arithmetic (symmetric tiles): same events, fields equal to rounding
    for (dt, im, tb), (dt2, im2, tb2) in zip(rtrace, ref_tiles[0]):
        assert abs(dt - dt2) <= 1e-9 * dt and abs(im - im2) <= 1e-9 * abs(im) and abs(tb - tb2) <= 1e-9 * tb
    assert np.array_equal(rfields["site_element"], ref_tiles[2]["site_element"])
    nseg = got0[3]["spmv_segments"]
    assert got0[3]["comm_local_segments"] + got1[3]["comm_local_segments"] == nseg > 0      # the ranks split the segments
    assert abs(got0[3]["comm_local_segments"] - got1[3]["comm_local_segments"]) <= 64      # balanced up to one row


def test_rccl_transport_one_rank():
    import __graft_entry__ as g
    g.build()
    from devicekmc_amd import lib, parallel
    ref = _supersteps(2)
    assert parallel.attach_solver_comm("rccl") == "rccl"
    try:
        import ctypes as C
        n, r, t = C.c_int(), C.c_int(), C.c_int()
        lib.load().dkmc_comm_info(C.byref(n), C.byref(r), C.byref(t))
        assert (n.value, r.value, t.value) == (1, 0, 1)
        got = _supersteps(2)
    finally:
        parallel.detach_solver_comm()
        lib.load().dkmc_set_symmetric_tiles(1)
    assert got[0] == ref[0] and got[1] == ref[1]
    for name in ref[2]:
        assert np.array_equal(got[2][name], ref[2][name]), name
    assert got[3]["comm_ranks"] == 1 and got[3]["comm_local_segments"] == got[3]["spmv_segments"]
