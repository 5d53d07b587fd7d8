"""Start vector of the current solve (dkmc_set_current_warm_start).  Mode 1 -- the library default since round 5 -- starts from the
previous step's solution (a private unscaled copy), which is what the reference's own comment asks for (current_solver_gpu.cu:976-977);
mode 0 reads gpubuf.atom_virtual_potentials as the reference code does, and that buffer was scaled by G0 in place after the previous
solve (:1013-1016).  A start vector changes the iterates, never the contract: the solution within the reference's stop test.  Checked
here with the criteria of the block loop (tests/test_gpu_block_cg.py): true scaled residual of the ORACLE's X, events with the oracle fed
the GPU's potentials, I_macro against the reference-order solve of the same state, the reference's own log, restart."""
import numpy as np
import pytest

from conftest import params_7p5
from test_gpu_parity import Vd, _scaled_residual, get, hip, make_pair, put  # noqa: F401

pytestmark = pytest.mark.gpu


def test_warm_start_coupled_supersteps_default_tolerance_7p5(dev_7p5, hip):
    """85 071 sites, the library's defaults (tolerance 1e-6, block-CG of width 16, warm start 1), ten coupled supersteps.  Every step: the
    oracle FED the GPU's potentials selects the same events and dt; the GPU's X solution meets the stop test in the true scaled residual
    of the oracle's X assembled from the same state; I_macro agrees to 1e-5 with the reference-order solve of the same state
    (single-vector loop, reference start vector).  From the second step on the warm start needs fewer sweeps than the reference start."""
    host, L = hip
    p = params_7p5(); p.solve_heating_global = False
    assert L.dkmc_get_current_warm_start() == 1 and L.dkmc_get_x_block() == 16 and p.cg_tol == 1e-6
    dev, sim, gb, o = make_pair(dev_7p5, p, hip)
    sweeps_warm, sweeps_ref_start = [], []
    for k in range(10):
        dev.updateCharge(gb); dev.updatePotential(gb, p, Vd, k)
        o.update_charge()
        assert np.array_equal(get(gb, "site_charge"), o.charge)
        o.pot_boundary[:] = get(gb, "site_potential_boundary"); o.pot_charge[:] = get(gb, "site_potential_charge")
        _, dt = sim.executeKMCStep(gb, dev, want_log=True)
        odt = o.execute_kmc_step()
        assert np.array_equal(sim.last_event_log, o.last_events["log"]), k
        assert abs(dt - odt) <= 1e-12 * odt
        dev.updatePower(gb, p, Vd)
        st = host.get_stats()
        assert st["xb_width"] == 16 and st["xb_fallback"] == 0
        sweeps_warm.append(st["cg_iters_X"])
        im_warm = dev.imacro
        m = get(gb, "atom_virtual_potentials").copy()
        X = o.assemble_X()
        assert st["X_nnz"] == len(X["col"])
        assert _scaled_residual(X["row_ptr"], X["col"], X["data"], m, p.G0, p.X_loop_G) <= 10 * p.cg_tol, (k, st["cg_iters_X"])
        # the same state solved in reference order: single-vector loop from the reference's start vector (the private copy is left alone in mode 0)
        try:
            L.dkmc_set_current_warm_start(0); L.dkmc_set_x_block(1)
            dev.updatePower(gb, p, Vd)
            assert host.get_stats()["xb_width"] == 1
            assert abs(im_warm / dev.imacro - 1) <= 1e-5, (k, im_warm, dev.imacro)
            # ... and the block loop from the reference's start vector, for the sweep count
            L.dkmc_set_x_block(16)
            put(gb, "atom_virtual_potentials", m)
            dev.updatePower(gb, p, Vd)
            sweeps_ref_start.append(host.get_stats()["cg_iters_X"])
        finally:
            L.dkmc_set_current_warm_start(1); L.dkmc_set_x_block(16)
        put(gb, "atom_virtual_potentials", m)
    print("sweeps per step, warm start:", sweeps_warm, " reference start:", sweeps_ref_start)
    assert abs(sweeps_warm[0] - sweeps_ref_start[0]) <= 5      # first solve: nothing to start from
    assert sum(sweeps_warm[1:]) < 0.8 * sum(sweeps_ref_start[1:]), (sweeps_warm, sweeps_ref_start)


@pytest.mark.parametrize("block", [16, 1])
def test_warm_start_reference_log_7p5(dev_7p5, hip, ref_logs, block):
    """The reference's own CUDA-path log of configs[1] under log_revision() with the warm start on (block loop and single-vector loop):
    KMC time and Current [uA] of all 19 logged supersteps to the six printed digits -- the start vector does not move a converged solve."""
    host, L = hip
    gold = ref_logs["timing_7.5nm/output_noguess.txt"]["steps"]
    p = params_7p5().log_revision()
    L.dkmc_set_x_block(block)
    assert L.dkmc_get_current_warm_start() == 1
    dev = host.Device(dev_7p5, p); sim = host.KMCProcess(dev, p.freq)
    gb = dev.make_gpubuf("cuda:0")
    dev.setLaplacePotential(gb, p, Vd)
    t = 0.0
    sweeps = []
    for k in range(len(gold)):
        dev.updateCharge(gb); dev.updatePotential(gb, p, Vd, k)
        _, dt = sim.executeKMCStep(gb, dev); t += dt
        dev.updatePower(gb, p, Vd)
        sweeps.append(host.get_stats()["cg_iters_X"])
        assert abs(t / gold[k]["KMC time"] - 1) < 5e-6, (k, t, gold[k])
        assert abs(dev.imacro * 1e6 - gold[k]["Current [uA]"]) <= 1e-4, (k, dev.imacro * 1e6, gold[k])     # 6th printed digit +- 1
    print("block", block, "sweeps per step:", sweeps)
    assert max(sweeps[1:]) < sweeps[0]


def test_warm_vector_export_import(cell_2p5, hip):
    """dkmc_get / set_current_warm_vector: the private copy is the unscaled solution (buffer / G0 with heating off); restoring it into a
    fresh GPUBuffers makes the next solve start where the original would (same sweep count, same bits)."""
    import ctypes as C
    from devicekmc_amd import params as pm
    from devicekmc_amd.lib import check
    host, L = hip
    p = pm.KMCParameters(); p.solve_heating_global = False

    def fresh():
        dev = host.Device(cell_2p5, p); sim = host.KMCProcess(dev, p.freq); gb = dev.make_gpubuf("cuda:0")
        dev.setLaplacePotential(gb, p, Vd); gb.sync_HostToGPU(dev)
        dev.updateCharge(gb); dev.updatePotential(gb, p, Vd, 0)
        return dev, sim, gb

    dev, sim, gb = fresh()
    n = C.c_int(-1)
    check(L.dkmc_get_current_warm_vector(C.byref(gb.c), None, 0, C.byref(n)))
    assert n.value == 0
    dev.updatePower(gb, p, Vd)
    check(L.dkmc_get_current_warm_vector(C.byref(gb.c), None, 0, C.byref(n)))
    assert n.value == dev.N_atom + 1
    w = np.zeros(n.value)
    check(L.dkmc_get_current_warm_vector(C.byref(gb.c), w.ctypes.data_as(C.c_void_p), n.value, C.byref(n)))
    m = get(gb, "atom_virtual_potentials")
    assert np.array_equal(w * p.G0, m[:n.value])
    dev.updatePower(gb, p, Vd)                      # second solve of the same state, from the solution
    second = (dev.imacro, host.get_stats()["cg_iters_X"], get(gb, "atom_virtual_potentials").copy())
    assert second[1] <= 2
    dev2, sim2, gb2 = fresh()
    check(L.dkmc_set_current_warm_vector(C.byref(gb2.c), w.ctypes.data_as(C.c_void_p), n.value))
    dev2.updatePower(gb2, p, Vd)
    assert (dev2.imacro, host.get_stats()["cg_iters_X"]) == second[:2] and np.array_equal(get(gb2, "atom_virtual_potentials"), second[2])
    check(L.dkmc_set_current_warm_vector(C.byref(gb2.c), None, 0))
    check(L.dkmc_get_current_warm_vector(C.byref(gb2.c), None, 0, C.byref(n)))
    assert n.value == 0


def test_auxiliary_columns_warm_start_option(cell_2p5, hip):
    """dkmc_set_x_aux_warm(1) (off by default: profiles/r05_ab_aux_warm.json): the hash auxiliary columns start from the solutions the previous
    solve left, the default set becomes half smooth + half hash.  Whatever the auxiliary columns start from, the physical column's solution meets
    the reference's stop test in the true scaled residual of the CSR X; here over four coupled supersteps, and the auxiliary panel survives an
    export / import (restart)."""
    import ctypes as C
    from devicekmc_amd import params as pm
    from devicekmc_amd.lib import check
    host, L = hip
    p = pm.KMCParameters(); p.solve_heating_global = False
    try:
        L.dkmc_set_x_aux_warm(1)
        dev = host.Device(cell_2p5, p); sim = host.KMCProcess(dev, p.freq); gb = dev.make_gpubuf("cuda:0")
        dev.setLaplacePotential(gb, p, Vd); gb.sync_HostToGPU(dev)
        sweeps = []
        for k in range(4):
            dev.updateCharge(gb); dev.updatePotential(gb, p, Vd, k)
            sim.executeKMCStep(gb, dev)
            dev.updatePower(gb, p, Vd)
            st = host.get_stats()
            assert st["xb_width"] == 16 and st["xb_fallback"] == 0 and st["xb_aux"] == 2       # 2: mixed set
            sweeps.append(st["cg_iters_X"])
            m = get(gb, "atom_virtual_potentials").copy()
            L.dkmc_set_x_format(0); L.dkmc_set_current_warm_start(0)
            try:
                dev.updatePower(gb, p, Vd)                     # the CSR X of the same state (for the true residual); leaves the private copies alone
                rp, ci, data = host.get_last_X()
            finally:
                L.dkmc_set_x_format(1); L.dkmc_set_current_warm_start(1)
            assert _scaled_residual(rp, ci, data, m, p.G0, p.X_loop_G) <= 10 * p.cg_tol, (k, sweeps)
        print("sweeps with the warm auxiliary start (2.5 nm):", sweeps)
        n = C.c_longlong(0)
        check(L.dkmc_get_current_warm_aux(C.byref(gb.c), None, 0, C.byref(n)))
        assert n.value == 16 * (dev.N_atom + 1)
        w = np.zeros(n.value)
        check(L.dkmc_get_current_warm_aux(C.byref(gb.c), w.ctypes.data_as(C.c_void_p), n.value, C.byref(n)))
        assert np.abs(w.reshape(-1, 16)[:, 8:]).max() > 0 and np.abs(w.reshape(-1, 16)[:, 1:8]).max() > 0
        check(L.dkmc_set_current_warm_aux(C.byref(gb.c), w.ctypes.data_as(C.c_void_p), n.value))
        check(L.dkmc_set_current_warm_aux(C.byref(gb.c), None, 0))
        check(L.dkmc_get_current_warm_aux(C.byref(gb.c), None, 0, C.byref(n)))
        assert n.value == 0
    finally:
        L.dkmc_set_x_aux_warm(0)
