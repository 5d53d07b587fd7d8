"""The sharded current solve (csrc/comm.hip + xt.hip / cg.hip): N ranks advance one simulation in lockstep.

* Tiled X (default, dkmc_set_x_format(1)): the work items (runs of tiles) are dealt to the ranks in contiguous, byte-balanced
  shares; a rank generates, stores and streams only its tiles; one all-reduce of |S| doubles per matrix-vector product.  All ranks
  hold the same bits; the result equals the single-GPU one to rounding.
* CSR X (dkmc_set_x_format(0)): the long rows are dealt to the ranks, one all-gather of row sums per iteration, values and
  summation orders of the single-GPU kernels: bit-identical to the single-GPU run.

Covered here: two ranks sharing cuda:0 over the host-callback transport (gloo) -- RCCL refuses two ranks on one device and the
test box has one GPU; this covers the partitioning, the per-rank storage, the lockstep launch plan and the exchange placement --
and the RCCL transport itself with a communicator of one rank (dlopen, ncclCommInitRank, in-place collectives on the engine's
stream).  (File name: sorts before the other GPU tests so that the ranks are spawned from a parent that has not touched the GPU.)
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


def _supersteps(nsteps, seed=1, fmt=0, big=False):
    """nsteps supersteps from a fresh state of the 2.5 nm device (big: the 85 071-site 7.5 nm device); returns everything a
    caller of the path can observe.  fmt: dkmc_set_x_format (0 = CSR X, the arithmetic the sharded solve reproduces bit for bit)."""
    import torch
    from devicekmc_amd import host, lib, params, structure
    lib.load().dkmc_set_x_format(fmt)
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    if big:
        s = structure.load_structure(os.path.join(g, "device_7.5nm.npz"))
        p = params.KMCParameters(rnd_seed=5, lattice=(108.984050, 76.725000, 76.725000), num_atoms_first_layer=1296,
                                 num_atoms_contact=12960, A=76.725e-10 * 76.725e-10)
        p.cg_tol = 1e-10        # converged solves: the comparison is then about rounding, not about which iterate the stop test picks
    else:
        s = structure.load_structure(os.path.join(g, "device_2.5nm.npz"))
        p = params.KMCParameters()
    p.solve_heating_global = True; p.rnd_seed_kmc = seed
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
    lib.load().dkmc_set_x_format(1)
    return trace, iters, fields, dict(host.get_stats())


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    from devicekmc_amd import parallel
    parallel.init("gloo")
    torch.cuda.set_device(0)
    ref = _supersteps(NSTEPS) if rank == 0 else None           # single-GPU path, no communicator
    ref_tiles = _supersteps(2, fmt=1, big=True) if rank == 0 else None      # default single-GPU arithmetic (tiled X), 85 k sites
    parallel.barrier()
    assert parallel.attach_solver_comm() == "host"
    got = _supersteps(NSTEPS)
    got_tiles = _supersteps(2, fmt=1, big=True)                 # block-CG distributed by row slabs (default): three exchanges per sweep
    from devicekmc_amd import lib as _lib
    _lib.load().dkmc_set_x_slab(0)
    got_tiles_ag = _supersteps(2, fmt=1, big=True)              # all-gather variant: tile stream sharded only, one all-gather per sweep
    _lib.load().dkmc_set_x_slab(1)
    _lib.load().dkmc_set_k_blocked(0)                           # K on the CSR positions: with a communicator the CG on K is distributed by row slabs too
    got_kslab = _supersteps(2, fmt=1, big=True)
    _lib.load().dkmc_set_k_blocked(1)
    parallel.detach_solver_comm()
    parallel.barrier()
    q.put((rank, ref, got, ref_tiles, got_tiles, got_tiles_ag, got_kslab))
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
    (_, ref, got0, ref_tiles, gt0, ga0, gk0), (_, _, got1, _, gt1, ga1, gk1) = out
    rtrace, riters, rfields, _ = ref
    for rank, (trace, iters, fields, st) in enumerate((got0, got1)):
        assert trace == rtrace, (rank, trace, rtrace)                 # dt, I_macro, T_bg of every step: exact
        assert iters == riters                                        # same CG iteration counts
        for n in rfields:
            assert np.array_equal(fields[n], rfields[n]), (rank, n)   # every field a caller can read back: bit-identical
        assert st["comm_ranks"] == 2 and st["comm_count_per_rank"] % 2 == 0
    # sharded solve on the tiled X: the ranks agree bit for bit and match the single-GPU solve to rounding
    assert gt0[0] == gt1[0] and gt0[1] == gt1[1]
    for n in gt0[2]:
        assert np.array_equal(gt0[2][n], gt1[2][n]), n
    assert gt0[3]["spmv_tiles"] > 0 and gt0[3]["comm_ranks"] == 2
    for (dt, im, tb), (dt2, im2, tb2) in zip(gt0[0], ref_tiles[0]):
        assert abs(dt - dt2) <= 1e-8 * dt and abs(im - im2) <= 1e-8 * abs(im) and abs(tb - tb2) <= 1e-8 * tb
    assert np.array_equal(gt0[2]["site_element"], ref_tiles[2]["site_element"])
    assert np.abs(gt0[2]["site_power"] - ref_tiles[2]["site_power"]).max() <= 1e-8 * np.abs(ref_tiles[2]["site_power"]).max()
    # every rank generated, stored and streamed only its share of the tiles
    tot = gt0[3]["xt_subblocks"]
    assert tot == gt1[3]["xt_subblocks"] > 0 and gt0[3]["xt_local_subblocks"] + gt1[3]["xt_local_subblocks"] == tot
    assert abs(gt0[3]["xt_local_subblocks"] - gt1[3]["xt_local_subblocks"]) <= 8 * 16          # balanced up to one work item
    # the all-gather variant (dkmc_set_x_slab(0)): ranks bit-identical, equal to the single-GPU run to rounding, like the slab-distributed default
    assert ga0[0] == ga1[0] and ga0[1] == ga1[1]
    for n in ga0[2]:
        assert np.array_equal(ga0[2][n], ga1[2][n]), n
    for (dt, im, tb), (dt2, im2, tb2) in zip(ga0[0], ref_tiles[0]):
        assert abs(dt - dt2) <= 1e-8 * dt and abs(im - im2) <= 1e-8 * abs(im) and abs(tb - tb2) <= 1e-8 * tb
    # K-CG distributed by row slabs as well (dkmc_set_k_blocked(0): CSR positions): ranks bit-identical, the single-GPU trajectory to rounding
    assert gk0[0] == gk1[0] and gk0[1] == gk1[1]
    for n in gk0[2]:
        assert np.array_equal(gk0[2][n], gk1[2][n]), n
    for (dt, im, tb), (dt2, im2, tb2) in zip(gk0[0], ref_tiles[0]):
        assert abs(dt - dt2) <= 1e-7 * dt and abs(im - im2) <= 1e-7 * abs(im) and abs(tb - tb2) <= 1e-7 * tb
    assert np.abs(gk0[2]["site_potential_boundary"] - ref_tiles[2]["site_potential_boundary"]).max() <= 1e-7 * Vd
    assert gk0[3]["kcg_blocked"] == 0
    # per rank and all-gather of that variant (width 16): |S| x 16 tile sums + the stop decision + the abort word
    # (at the 1e-10 of this test the s x s systems can lose definiteness in the last sweeps -- the solve then finishes in the
    # single-vector loop, |S| + 2 doubles per exchange, on every rank alike: they hold the same Gram matrices)
    assert ga0[3]["xb_width"] == 16 and ga0[3]["xb_fallback"] == ga1[3]["xb_fallback"]
    assert ga0[3]["comm_count_per_rank"] == (16 if not ga0[3]["xb_fallback"] else 1) * ga0[3]["xt_ns"] + 2, (ga0[3]["xb_fallback"], ga0[3]["comm_count_per_rank"], ga0[1])
    # slab-distributed default: a rank RECEIVES about |S| x 16 / 2 partial sums + |S| x 16 / 2 rows of QS + its halo + one Gram block per sweep
    assert gt0[3]["xb_width"] == 16 and gt0[3]["xb_fallback"] == gt1[3]["xb_fallback"]
    if not gt0[3]["xb_fallback"]:
        assert 0 < gt0[3]["comm_count_per_rank"] < 1.6 * 16 * gt0[3]["xt_ns"], (gt0[3]["comm_count_per_rank"], gt0[3]["xt_ns"])
    # the tunnelling-coefficient cache is sharded with the tiles: a rank holds the left-contact columns of its own vacancies and the
    # right-contact columns of its own windows for all vacancies -- together about what one GPU holds, each well under it
    one, b0, b1 = ref_tiles[3]["tcache_bytes"], gt0[3]["tcache_bytes"], gt1[3]["tcache_bytes"]
    assert one > 0 and 0 < b0 < 0.8 * one and 0 < b1 < 0.8 * one and b0 + b1 < 1.5 * one, (one, b0, b1)
    nseg = got0[3]["spmv_segments"]
    assert got0[3]["comm_local_segments"] + got1[3]["comm_local_segments"] == nseg > 0      # the ranks split the segments
    assert abs(got0[3]["comm_local_segments"] - got1[3]["comm_local_segments"]) <= 64      # balanced up to one row


def _fault_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    from devicekmc_amd import host, lib, parallel, params, structure
    from devicekmc_amd.lib import DeviceKMCError
    parallel.init("gloo")
    torch.cuda.set_device(0)
    assert parallel.attach_solver_comm() == "host"
    L = lib.load()
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    s = structure.load_structure(os.path.join(g, "device_2.5nm.npz"))
    p = params.KMCParameters(); p.solve_heating_global = True
    dev = host.Device(s, p); sim = host.KMCProcess(dev, p.freq)
    gb = dev.make_gpubuf("cuda:0")
    dev.setLaplacePotential(gb, p, Vd); gb.sync_HostToGPU(dev)
    seen = []
    # fault in the assembly; on the host side of block-CG iteration 5 (the default loop); on the host side of iteration 5 of the
    # single-vector loop (dkmc_set_x_block(1)); clean step
    for phase, it in ((1, 0), (3, 5), (2, 5), (0, 0)):
        dev.updateCharge(gb); dev.updatePotential(gb, p, Vd, 0)
        sim.executeKMCStep(gb, dev)
        L.dkmc_set_x_block(1 if phase == 2 else 16)
        if rank == 1 and phase:
            L.dkmc_debug_inject_fault(phase, it)
        try:
            dev.updatePower(gb, p, Vd)
            seen.append((0, dev.imacro))
        except DeviceKMCError as exc:
            seen.append((1, str(exc)))
            L.dkmc_clear_error()
        parallel.barrier()
    parallel.detach_solver_comm()
    q.put((rank, seen))
    parallel.finalize()


def test_sharded_error_path_returns_on_every_rank():
    """A rank-local failure inside a sharded current solve must not leave the peers blocked in a collective.  Rank 1 fails once in the
    assembly of X (before the first collective: the ranks agree on the outcome of the set-up, comm_agree) and once on the host side of an
    iteration of each CG loop, the block-CG (default) and the single-vector one (the abort word travels with the next all-reduce): every
    time BOTH ranks return an error from update_power -- the
    failing rank its own, the peer "a peer rank ..." -- and the next, clean superstep runs on both and gives the same current."""
    import __graft_entry__ as g
    g.build()
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_fault_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs: p.start()
    out = sorted((q.get(timeout=300) for _ in range(world)), key=lambda t: t[0])
    for p in procs: p.join(60); assert p.exitcode == 0
    (_, s0), (_, s1) = out
    assert [k for k, _ in s0] == [1, 1, 1, 0] and [k for k, _ in s1] == [1, 1, 1, 0], (s0, s1)
    assert "peer rank" in s0[0][1] and "injected fault (assembly" in s1[0][1]
    assert "peer rank" in s0[1][1] and "injected fault (block-CG iteration" in s1[1][1]
    assert "peer rank" in s0[2][1] and "injected fault (CG iteration" in s1[2][1]
    assert s0[3][1] == s1[3][1] and s0[3][1] != 0.0


def _peer_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    from devicekmc_amd import host, lib, parallel, params, structure
    from devicekmc_amd.lib import DeviceKMCError
    parallel.init("gloo")
    torch.cuda.set_device(0)
    L = lib.load()
    assert parallel.attach_solver_comm() == "host"
    L.dkmc_set_x_slab(0)                                        # the peer-write exchange carries the slots of the ALL-GATHER variant of the sharded block loop
    got_ag = _supersteps(2, fmt=1, big=True)                    # exchange of the block loop = the communicator's all-gather
    ok = parallel.attach_peer_exchange(16 * 9000 + 2)           # |S| = 8 352 at 85 k sites
    L.dkmc_set_profiling(1)
    got_peer = _supersteps(2, fmt=1, big=True) if ok else None  # exchange = push + signal + wait over hipIpc-mapped buffers
    L.dkmc_set_profiling(0)
    info = parallel.peer_exchange_info()
    # a rank-local failure on the host side of a block iteration, with the peer exchange carrying the abort word; then a clean step
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    s = structure.load_structure(os.path.join(g, "device_2.5nm.npz"))
    p = params.KMCParameters(); p.solve_heating_global = True
    dev = host.Device(s, p); sim = host.KMCProcess(dev, p.freq)
    gb = dev.make_gpubuf("cuda:0")
    dev.setLaplacePotential(gb, p, Vd); gb.sync_HostToGPU(dev)
    seen = []
    for phase, it in ((3, 5), (0, 0)):
        dev.updateCharge(gb); dev.updatePotential(gb, p, Vd, 0)
        sim.executeKMCStep(gb, dev)
        if rank == 1 and phase:
            L.dkmc_debug_inject_fault(phase, it)
        try:
            dev.updatePower(gb, p, Vd)
            seen.append((0, dev.imacro))
        except DeviceKMCError as exc:
            seen.append((1, str(exc)))
            L.dkmc_clear_error()
        parallel.barrier()
    info2 = parallel.peer_exchange_info()
    # the failed solve dropped the attachment on every rank (their sequence counters may have drifted apart): the clean step above ran over the
    # all-gather.  Attaching again resets counters and flags on every rank; one more clean step over the peer exchange
    ok2 = parallel.attach_peer_exchange(16 * 9000 + 2)
    dev.updateCharge(gb); dev.updatePotential(gb, p, Vd, 1)
    sim.executeKMCStep(gb, dev)
    dev.updatePower(gb, p, Vd)
    seen.append((0, dev.imacro))
    info3 = parallel.peer_exchange_info()
    info3["reattached"] = bool(ok2)
    parallel.detach_solver_comm()
    q.put((rank, ok, got_ag, got_peer, info, (info2, info3), seen))
    parallel.finalize()


def test_peer_write_exchange_two_ranks_one_gpu():
    """The one-shot peer-write exchange of the sharded block-CG (csrc/comm.hip): two processes sharing cuda:0 map each
    other's exchange buffers over hipIpc; a sweep's exchange is push + signal + bounded wait on the stream, and every rank adds the slots in
    rank order.  Same slots, same order as with the communicator's all-gather: the two supersteps at 85 k sites give the SAME BITS as over
    the all-gather, on both ranks; the abort word of a failing rank still ends the loop on both ranks; the next clean step agrees."""
    import __graft_entry__ as g
    g.build()
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_peer_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs: p.start()
    out = sorted((q.get(timeout=600) for _ in range(world)), key=lambda t: t[0])
    for p in procs: p.join(120); assert p.exitcode == 0
    (_, ok0, ag0, pe0, info0, info0b, seen0), (_, ok1, ag1, pe1, info1, info1b, seen1) = out
    assert ok0 and ok1
    print("peer exchange (two processes, one GPU):", info0, info1, "sweeps", pe0[1])
    for ag, pe in ((ag0, pe0), (ag1, pe1), (ag0, pe1)):
        assert pe[0] == ag[0] and pe[1] == ag[1]                      # dt, I_macro, T_bg and sweep counts of every step: exact
        for n in ag[2]:
            assert np.array_equal(pe[2][n], ag[2][n]), n
    assert pe0[3]["xb_width"] == 16 and pe0[3]["comm_ranks"] == 2
    for info in (info0, info1):
        assert info["ready"] and info["slot_doubles"] == 16 * 9000 + 2
        assert info["exchanges"] >= sum(pe0[1]) > 0                   # one exchange per sweep (+ the first product of each solve)
        assert 0.0 < info["mean_us"] < 5e4, info
    # the fault path: both ranks return an error, the clean step after it runs on both and agrees; the small device used the exchange too
    assert [k for k, _ in seen0] == [1, 0, 0] and [k for k, _ in seen1] == [1, 0, 0], (seen0, seen1)
    assert "peer rank" in seen0[0][1] and "injected fault (block-CG iteration" in seen1[0][1]
    assert seen0[1][1] == seen1[1][1] != 0.0 and seen0[2][1] == seen1[2][1] != 0.0
    # a failed solve drops the attachment on BOTH ranks (no rank keeps waiting for a sequence number its peer never reaches); a new attach works
    for after_fault, after_reattach in (info0b, info1b):
        assert not after_fault["ready"] and after_fault["exchanges"] == 0
        assert after_reattach["reattached"] and after_reattach["ready"] and after_reattach["exchanges"] > 0


def test_emulated_shares_cover_the_tunnelling_block():
    """Shares of 2 ... 64 ranks built on one GPU as a sharded assembly builds them (dkmc_xt_check_shares): every stored sub-block and
    every work item in exactly one share; the ranks' partial row sums add up to the one-GPU tile pass."""
    import ctypes as C
    import __graft_entry__ as g
    g.build()
    from devicekmc_amd import lib
    ref = _supersteps(1, fmt=1, big=True)
    L = lib.load()
    for nr in (1, 2, 3, 8, 64):
        md, ma, sb, its, itot = C.c_double(), C.c_double(), C.c_longlong(), C.c_longlong(), C.c_int()
        lib.check(L.dkmc_xt_check_shares(nr, C.byref(md), C.byref(ma), C.byref(sb), C.byref(its), C.byref(itot)))
        assert sb.value == ref[3]["xt_subblocks"] and its.value == itot.value > 0, (nr, sb.value, its.value, itot.value)
        assert ma.value > 0 and md.value <= 1e-12 * ma.value, (nr, md.value, ma.value)
    assert L.dkmc_xt_check_shares(65, None, None, None, None, None) != 0        # more ranks than the split table holds: refused
    L.dkmc_clear_error()
    # the 2.5 nm device has 32 tiles: with 64 ranks most shares are empty
    small = _supersteps(1, fmt=1)
    assert 0 < small[3]["spmv_tiles"] < 64
    md, ma, sb, its, itot = C.c_double(), C.c_double(), C.c_longlong(), C.c_longlong(), C.c_int()
    lib.check(L.dkmc_xt_check_shares(64, C.byref(md), C.byref(ma), C.byref(sb), C.byref(its), C.byref(itot)))
    # work items = the 32 single-tile runs + the empty runs that complete every group of four at strip and share ends
    assert sb.value == small[3]["xt_subblocks"] and its.value == itot.value >= small[3]["spmv_tiles"] and itot.value % 4 == 0
    assert md.value <= 1e-12 * ma.value


def test_slab_distributed_block_cg_virtual_ranks():
    """the block-CG with its STATE distributed by spatial row slabs (csrc/xtb_slab.inc), run with N = 1, 2, 5, 8
    VIRTUAL ranks inside one process on the X of the 85 071-site device (dkmc_xtb_emulate_slabs: every virtual rank with its own panels, lists
    and exchange buffers, shares of the tiles as a sharded assembly builds them, the three exchanges of a sweep as device copies).  The
    emulation itself fails unless all virtual ranks leave the loop at the same sweep and end with the same bits; checked here: the distributed
    solution agrees with the one-GPU block-CG to 1e-8 of the largest entry at a converged tolerance (two roundings of the same Krylov process),
    the sweep counts agree to a few sweeps, the slabs are balanced, and a rank receives ~1/N of what the all-gather variant moved."""
    import ctypes as C
    import __graft_entry__ as g
    g.build()
    from devicekmc_amd import lib
    L = lib.load()
    ref = _supersteps(1, fmt=1, big=True)
    ns = ref[3]["xt_ns"]
    for nr in (1, 2, 5, 8):
        rd, it_s, it_r = C.c_double(-1), C.c_int(0), C.c_int(0)
        us, xd, mm = (C.c_double * 8)(), (C.c_longlong * 3)(), (C.c_int * 2)()
        lib.check(L.dkmc_xtb_emulate_slabs(nr, 16, 1e-10, nr // 2 if nr > 1 else -1, 0, C.byref(rd), C.byref(it_s), C.byref(it_r), us, xd, mm))
        print("slabs N=%d: rel diff %.2e, sweeps %d (one GPU %d), rows per slab %d..%d, doubles received per sweep %s, kernel us %s"
              % (nr, rd.value, it_s.value, it_r.value, mm[0], mm[1], list(xd), [round(x, 1) for x in us]))
        assert 0 <= rd.value <= 1e-8, (nr, rd.value)
        assert abs(it_s.value - it_r.value) <= max(3, it_r.value // 20), (nr, it_s.value, it_r.value)
        assert mm[0] > 0 and mm[1] <= 1.25 * mm[0] + 64, (nr, mm[0], mm[1])            # balanced by row count (cuts fall on bins of the lateral coordinate)
        if nr > 1:
            # exchange 1: the other ranks' partial sums of MY S rows; exchange 3: everybody else's rows of QS + my halo: together ~ 2 |S| 16 (N - 1) / N + halo
            assert xd[0] <= 1.3 * (nr - 1) * (ns // nr + 64) * 16 + 64 * nr, (nr, list(xd))
            assert xd[1] == (nr - 1) * (6 * 256 + 2)
            assert 16 * (ns - ns // nr - 64 * nr) <= xd[2] <= 16 * ns + 16 * 0.6 * 60000, (nr, list(xd))
            assert us[0] > 0 and us[3] > 0 and us[6] > 0


def test_slab_distributed_K_cg_virtual_ranks():
    """SURVEY 8(e) row "K-CG" / configs[4]'s domain-decomposed potential: the CG on K distributed by lateral row slabs (csrc/kcg.hip), run with
    N = 1, 2, 5, 8 VIRTUAL ranks in one process on the background-potential system of the 85 071-site device (dkmc_kcg_emulate_slabs: product,
    update, direction on each virtual rank's rows, the two dot products completed by all-gathers of block partials, the halo of the scaled
    direction by an all-to-all-v -- all as device copies).  The emulation fails unless all virtual ranks stop at the same iteration with the
    same r.r and end with the same bits; checked here against the one-GPU reference-order loop: potentials within 2e-7 V at a converged
    tolerance, iteration counts within 2 %, slabs balanced, the halo a small fraction of a slab."""
    import ctypes as C
    import __graft_entry__ as g
    g.build()
    import torch
    from devicekmc_amd import host, lib, params, structure
    L = lib.load()
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    s = structure.load_structure(os.path.join(gdir, "device_7.5nm.npz"))
    p = params.KMCParameters(rnd_seed=5, lattice=(108.984050, 76.725000, 76.725000), num_atoms_first_layer=1296, num_atoms_contact=12960, A=76.725e-10 * 76.725e-10)
    p.cg_tol = 1e-10
    dev = host.Device(s, p); gb = dev.make_gpubuf("cuda:0")
    dev.setLaplacePotential(gb, p, Vd); gb.sync_HostToGPU(dev)
    dev.updateCharge(gb)
    L.dkmc_set_cg_tolerance(p.cg_tol)
    n1 = p.num_atoms_first_layer
    before = gb.site_potential_boundary.cpu().numpy().copy()
    for nr in (1, 2, 5, 8):
        md, it_s, it_r = C.c_double(-1), C.c_int(0), C.c_int(0)
        us, hr = (C.c_double * 4)(), (C.c_longlong * 2)()
        lib.check(L.dkmc_kcg_emulate_slabs(C.byref(gb.c), dev.N, n1, n1, Vd, p.high_G, p.low_G, len(p.metals), nr, nr // 2 if nr > 1 else -1, 0,
                                           C.byref(md), C.byref(it_s), C.byref(it_r), us, hr))
        print("K slabs N=%d: max |dphi| %.2e V, iterations %d (one GPU %d), halo doubles per iteration %d, rows of the largest slab %d, kernel us %s"
              % (nr, md.value, it_s.value, it_r.value, hr[0], hr[1], [round(x, 1) for x in us]))
        # (two converged solves that round differently: cond(K) ~ 1e8 x the 1e-10 of the scaled residual leaves ~1e-8 V; 2.9e-8 V measured at N = 1,
        # where only the grouping of the block partials differs from the one-GPU loop)
        assert 0 <= md.value <= 2e-7, (nr, md.value)
        assert it_r.value > 100 and abs(it_s.value - it_r.value) <= max(3, it_r.value // 50), (nr, it_s.value, it_r.value)
        m = dev.N - 2 * n1
        assert hr[1] <= 1.25 * (m // nr) + 64
        if nr > 1:
            assert 0 < hr[0] < 0.8 * hr[1]
    torch.cuda.synchronize()
    assert np.array_equal(gb.site_potential_boundary.cpu().numpy(), before)          # the buffer is not changed


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
        got_t = _supersteps(1, fmt=1, big=True)               # all-reduce variant (ncclAllReduce in place), 85 k sites, block-CG (default)
        lib.load().dkmc_set_x_block(1)
        got_t1 = _supersteps(1, fmt=1, big=True)              # ... and the single-vector loop
    finally:
        lib.load().dkmc_set_x_block(16)
        parallel.detach_solver_comm()
    # (the sharded loops carry no split polynomial preconditioner: the one-GPU run they are compared with bit for bit runs without it too)
    lib.load().dkmc_set_x_poly(0)
    try:
        ref_t = _supersteps(1, fmt=1, big=True)
        lib.load().dkmc_set_x_block(1)
        ref_t1 = _supersteps(1, fmt=1, big=True)
    finally:
        lib.load().dkmc_set_x_block(16); lib.load().dkmc_set_x_poly(8)
    # one rank: the all-reduce is the identity and the exchange buffer holds the sums the one-GPU kernels form: same bits as without it, in both loops
    assert got_t[0] == ref_t[0] and got_t[1] == ref_t[1] and got_t[3]["spmv_tiles"] > 0 and got_t[3]["xb_width"] == 16
    assert got_t1[0] == ref_t1[0] and got_t1[1] == ref_t1[1] and got_t1[3]["xb_width"] == 1
    assert got_t[3]["xt_local_subblocks"] == got_t[3]["xt_subblocks"]
    assert got[0] == ref[0] and got[1] == ref[1]
    for name in ref[2]:
        assert np.array_equal(got[2][name], ref[2][name]), name
    assert got[3]["comm_ranks"] == 1 and got[3]["comm_local_segments"] == got[3]["spmv_segments"]
