"""Block-CG of the current solve on the tiled X (csrc/xtb.hip, dkmc_set_x_block): the MFMA tile x panel product against the single-vector
tile kernel, the block solve against the single-vector solve and against the oracle's X, determinism, and the coupled superstep at the
default tolerance with the oracle fed the GPU's potentials.  The block loop does not follow the reference's iterate sequence: the
contract is the SOLUTION within the reference's stop test (||S (X m - b)||_2 <= tol on the physical column)."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import params_7p5
from test_gpu_parity import Vd, _fresh_device, _scaled_residual, get, hip, make_pair, put  # noqa: F401

pytestmark = pytest.mark.gpu


def _solve(dev, gb, p, hip, width, tol, start=None, power=False):
    """One current solve from a zero (or given) start vector.  The solution is read with heating off (with it on, update_m shifts the
    node potentials, current_solver_gpu.cu:1044-1047); power=True repeats the solve with heating on for the dissipated power."""
    host, L = hip
    L.dkmc_set_x_block(width)
    L.dkmc_set_current_warm_start(0)            # these tests compare solves FROM THE START VECTOR THEY PUT (sweep counts included)
    p.cg_tol = tol
    p.solve_heating_global = False
    put(gb, "atom_virtual_potentials", np.zeros(dev.N_atom + 2) if start is None else start)
    dev.updatePower(gb, p, Vd)
    st = host.get_stats()
    rec = dict(m=get(gb, "atom_virtual_potentials").copy(), im=dev.imacro, iters=st["cg_iters_X"], width=st["xb_width"], fallback=st["xb_fallback"])
    if power:
        p.solve_heating_global = True
        put(gb, "atom_virtual_potentials", np.zeros(dev.N_atom + 2) if start is None else start)
        dev.updatePower(gb, p, Vd)
        rec["power"] = get(gb, "site_power").copy()
        p.solve_heating_global = False
    return rec


@pytest.mark.parametrize("which", ["2.5nm", "7.5nm"])
def test_tile_panel_product_matches_single_vector_kernel(cell_2p5, dev_7p5, hip, which):
    """One sweep of the MFMA tile x panel product (k_xtb_apply + the fold of k_xtb_rows) over 16 test vectors equals 16 passes of the
    single-vector tile kernel over the same tiles: both are sums of the same products, regrouped (1e-12 of the largest sum)."""
    from devicekmc_amd import params as pm
    host, L = hip
    structure, p = (cell_2p5, pm.KMCParameters()) if which == "2.5nm" else (dev_7p5, params_7p5())
    p.solve_heating_global = False
    try:
        L.dkmc_set_x_block(1)
        dev, sim, gb, _ = _fresh_device(structure, p, hip)
        dev.updateCharge(gb); dev.updatePotential(gb, p, Vd, 0); dev.updatePower(gb, p, Vd)
        from devicekmc_amd.lib import check
        # the tiles the product runs over: full and partial ones (the loop takes the two kinds on different paths)
        h = (C.c_longlong * 11)()
        check(L.dkmc_xt_tile_census(h))
        assert sum(h[:9]) > 0 and h[0] == 0 and h[8] > 0 and sum(h[1:8]) > 0, list(h)
        for form in (0, 1):                                  # 0: the product form of the loop, 1: the round-4 form (same sums, same order)
            L.dkmc_set_x_apply_form(form)
            for width in (16, 12, 8, 4):
                d, a = C.c_double(-1), C.c_double(-1)
                check(L.dkmc_xtb_check_product(width, C.byref(d), C.byref(a)))
                assert a.value > 0 and d.value <= 1e-12 * a.value, (form, width, d.value, a.value)
    finally:
        L.dkmc_set_x_block(16); L.dkmc_set_x_apply_form(0)


def test_block_cg_agrees_with_single_vector_cg_7p5(dev_7p5, hip):
    """85 071 sites, converged solves (1e-10): block widths 4, 8, 16 against the single-vector loop.  Every solution meets the stop test in
    the TRUE scaled residual of the CSR X; I_macro, the solution and the dissipated power agree to 1e-8; the block loop needs fewer sweeps
    (the oracle's X on the CPU: 666 -> 208 / 133 / 95 at 1e-6); a second run gives the same bits."""
    host, L = hip
    p = params_7p5(); p.cg_tol = 1e-10; p.solve_heating_global = False
    try:
        L.dkmc_set_x_format(0); L.dkmc_set_x_block(1)
        dev, sim, gb, _ = _fresh_device(dev_7p5, p, hip)
        dev.updateCharge(gb); dev.updatePotential(gb, p, Vd, 0); dev.updatePower(gb, p, Vd)
        rp, ci, data = host.get_last_X()
        L.dkmc_set_x_format(1)
        ref = _solve(dev, gb, p, hip, 1, 1e-10, power=True)
        assert ref["width"] == 1
        assert _scaled_residual(rp, ci, data, ref["m"], p.G0, p.X_loop_G) <= 1e-9
        n = len(rp) - 1
        for width in (4, 8, 16):
            a = _solve(dev, gb, p, hip, width, 1e-10, power=True)
            assert a["width"] == width and a["fallback"] == 0
            # (1e-10 is 6e-15 of ||S b||: the recurrence residual and the true one part at the 1e-10 level; 1.0e-9 measured at width 8)
            assert _scaled_residual(rp, ci, data, a["m"], p.G0, p.X_loop_G) <= 3e-9, (width, a["iters"])
            assert abs(a["im"] / ref["im"] - 1) <= 1e-8, width
            assert np.abs(a["m"][:n] - ref["m"][:n]).max() <= 1e-8 * np.abs(ref["m"][:n]).max(), width
            assert np.abs(a["power"] - ref["power"]).max() <= 1e-8 * np.abs(ref["power"]).max(), width
            assert a["iters"] < 0.5 * ref["iters"], (width, a["iters"], ref["iters"])
            b = _solve(dev, gb, p, hip, width, 1e-10)
            assert np.array_equal(a["m"], b["m"]) and a["iters"] == b["iters"] and a["im"] == b["im"], width
        # the default tolerance: stop test met in the true residual, far fewer sweeps
        r1 = _solve(dev, gb, p, hip, 1, 1e-6)
        for width in (8, 16):
            a = _solve(dev, gb, p, hip, width, 1e-6)
            assert _scaled_residual(rp, ci, data, a["m"], p.G0, p.X_loop_G) <= 1e-5, (width, a["iters"])
            assert a["iters"] < 0.3 * r1["iters"], (width, a["iters"], r1["iters"])
            assert abs(a["im"] / r1["im"] - 1) <= 1e-5
    finally:
        L.dkmc_set_x_format(1); L.dkmc_set_x_block(16); L.dkmc_set_cg_tolerance(1e-6)


@pytest.mark.parametrize("width", [8, 16])
def test_block_cg_superstep_default_tolerance_7p5(dev_7p5, hip, width):
    """Three coupled supersteps at the default tolerance with the block loop on: the oracle FED the GPU's potentials selects the same events
    and dt, and the GPU's X solution meets the stop test in the true scaled residual of the oracle's X, assembled from the same state."""
    host, L = hip
    p = params_7p5(); p.solve_heating_global = False
    try:
        L.dkmc_set_x_block(width)
        dev, sim, gb, o = make_pair(dev_7p5, p, hip)
        for k in range(3):
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
            assert st["xb_width"] == width and st["xb_fallback"] == 0
            X = o.assemble_X()
            m = get(gb, "atom_virtual_potentials")
            assert st["X_nnz"] == len(X["col"])
            assert _scaled_residual(X["row_ptr"], X["col"], X["data"], m, p.G0, p.X_loop_G) <= 10 * p.cg_tol, (k, st["cg_iters_X"])
    finally:
        L.dkmc_set_x_block(16)


def test_block_cg_small_and_degenerate_systems(cell_2p5, hip):
    """2.5 nm device (few tiles, partial tiles, launch-bound) at widths 2, 3, 8, 16, at the nominal bias and at one where the tunnelling
    block is nearly empty (0.03 V: contact-contact pairs fall below the 0.01 eV threshold).  Every solution meets the stop test in the true
    scaled residual of the CSR X; at 5 V the single-vector loop's I_macro is met to 1e-8 and the node potentials to 1e-7 of the largest.  (At 0.03 V two solves that both meet the 1e-10 stop test differ by up to 1.6e-5 in the interior nodes, 12 decades below the driven
    ones: cond(X) x residual, not a property of either loop.)"""
    from devicekmc_amd import params as pm
    host, L = hip
    try:
        for V in (5.0, 0.03):
            p = pm.KMCParameters(); p.solve_heating_global = False; p.cg_tol = 1e-10
            L.dkmc_set_x_block(1)
            dev = host.Device(cell_2p5, p); gb = dev.make_gpubuf("cuda:0")
            dev.setLaplacePotential(gb, p, V); gb.sync_HostToGPU(dev)
            dev.updateCharge(gb); dev.updatePotential(gb, p, V, 0)
            put(gb, "atom_virtual_potentials", np.zeros(dev.N_atom + 2)); dev.updatePower(gb, p, V)
            ref = (get(gb, "atom_virtual_potentials").copy(), dev.imacro, host.get_stats()["cg_iters_X"])
            rp, ci, data = host.get_last_X()
            assert _scaled_residual(rp, ci, data, ref[0], p.G0, p.X_loop_G, Vd=V) <= 1e-9
            for width in (2, 3, 8, 16):
                L.dkmc_set_x_block(width)
                put(gb, "atom_virtual_potentials", np.zeros(dev.N_atom + 2)); dev.updatePower(gb, p, V)
                st = host.get_stats()
                # (at 0.03 V the s x s systems can lose definiteness close to convergence -- width 3 does, after 191 sweeps: the solve then
                # finishes in the single-vector loop from the last good iterate, which is the designed behaviour and is checked like any other)
                assert st["xb_width"] == width and (st["xb_fallback"] == 0 or V < 1), (V, width, st["cg_iters_X"])
                m = get(gb, "atom_virtual_potentials")
                n = dev.N_atom + 1
                assert _scaled_residual(rp, ci, data, m, p.G0, p.X_loop_G, Vd=V) <= 3e-9, (V, width, st["cg_iters_X"])
                rel = np.abs(m[:n] - ref[0][:n]).max() / np.abs(ref[0][:n]).max()
                assert rel <= (1e-7 if V > 1 else 1e-3), (V, width, st["cg_iters_X"], ref[2], rel)
                # I_macro sums x (m_c - m_1) over the source row: at 0.03 V it is 2e-15 A, a 1e-12 cancellation of the node potentials, and two
                # solves that both meet the stop test give it to a few per cent only (measured 1.2e-3 at width 2, 2.7e-2 at width 3); at 5 V to 1e-8
                assert abs(dev.imacro - ref[1]) <= (1e-8 if V > 1 else 1e-1) * abs(ref[1]), (V, width, dev.imacro, ref[1])
    finally:
        L.dkmc_set_x_block(16); L.dkmc_set_cg_tolerance(1e-6)


@pytest.mark.parametrize("which", ["2.5nm", "7.5nm"])
def test_smooth_auxiliary_columns(cell_2p5, dev_7p5, hip, which):
    """The auxiliary right-hand sides of the block-CG are a free choice (dkmc_set_x_aux): the fixed-seed hash set, or the smooth set (the lowest
    Laplacian modes of the device's bounding box over the Jacobi scaling; default at tolerances >= 1e-8).  Whatever the set, the solution of the
    physical column meets the reference's stop test in the TRUE scaled residual of the CSR X and I_macro agrees to the tolerance's level; the
    smooth set needs fewer sweeps on these devices (oracle's X on the CPU: 48 -> 31 and 95 -> 76 with x modes); below 1e-8 the default falls
    back to the hash set (smooth columns converge before the physical one does and the s x s systems lose rank)."""
    from devicekmc_amd import params as pm
    host, L = hip
    structure, p = (cell_2p5, pm.KMCParameters()) if which == "2.5nm" else (dev_7p5, params_7p5())
    p.solve_heating_global = False
    try:
        L.dkmc_set_x_format(0); L.dkmc_set_x_block(1)
        dev, sim, gb, _ = _fresh_device(structure, p, hip)
        dev.updateCharge(gb); dev.updatePotential(gb, p, Vd, 0); dev.updatePower(gb, p, Vd)
        rp, ci, data = host.get_last_X()
        L.dkmc_set_x_format(1)
        res = {}
        for mode in (0, 1, 3, 2):
            L.dkmc_set_x_aux(mode)
            a = _solve(dev, gb, p, hip, 16, 1e-6)
            st = host.get_stats()
            assert st["xb_aux"] == (0 if mode == 0 else 1) and a["fallback"] == 0, mode
            assert _scaled_residual(rp, ci, data, a["m"], p.G0, p.X_loop_G) <= 1e-5, (mode, a["iters"])
            res[mode] = a
        for mode in (1, 3, 2):
            assert abs(res[mode]["im"] / res[0]["im"] - 1) <= 1e-5, mode
            assert res[mode]["iters"] < 0.95 * res[0]["iters"], (mode, res[mode]["iters"], res[0]["iters"])
        assert res[2]["iters"] == res[1]["iters"] and np.array_equal(res[2]["m"], res[1]["m"])      # default = the box modes at this tolerance
        b = _solve(dev, gb, p, hip, 16, 1e-6)                                                         # bitwise reproducible
        assert np.array_equal(b["m"], res[2]["m"]) and b["iters"] == res[2]["iters"]
        L.dkmc_set_x_aux(2)
        c = _solve(dev, gb, p, hip, 16, 1e-10)
        assert host.get_stats()["xb_aux"] == 0                                                        # converged tolerances keep the hash set
        assert _scaled_residual(rp, ci, data, c["m"], p.G0, p.X_loop_G) <= 3e-9
    finally:
        L.dkmc_set_x_format(1); L.dkmc_set_x_block(16); L.dkmc_set_x_aux(2); L.dkmc_set_cg_tolerance(1e-6)


def test_dpp_lane_exchange_matches_shfl_xor(tmp_path):
    """csrc/common.h replaces the ds_bpermute hipcc emits for __shfl_xor by DPP moves (xor_lane: quad_perm, row_shl / row_shr under bank masks, row_ror) in
    every per-row reduction of the sparse kernels: the probe builds those forms stand-alone and compares them with __shfl_xor lane by lane for the
    offsets 1, 2, 4, 8 (the reductions then add the same pairs in the same order: same bits as before the replacement)."""
    import shutil
    import subprocess
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not (os.path.exists(hipcc) or shutil.which(hipcc)):
        pytest.skip("no hipcc on this box")
    exe = str(tmp_path / "probe_dpp_xor")
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "probe_dpp_xor.hip")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", src, "-o", exe], check=True, capture_output=True, timeout=600)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "identical to __shfl_xor" in r.stdout, (r.returncode, r.stdout, r.stderr)


def test_split_polynomial_preconditioner_7p5(dev_7p5, hip):
    """dkmc_set_x_poly(d): the block loop on L A L (L = the degree-d series of (I - N)^(-1/2) on the neighbour part of the Jacobi-scaled X).  85 071
    sites, zero and non-zero start vectors, tolerances 1e-6 and 1e-10: every solution meets the reference's stop test in the TRUE scaled residual of the
    CSR X (the preconditioned loop stops on the residual of L A L and then checks the true one), I_macro agrees with the plain block loop's to the
    tolerance's level, and the loop needs fewer sweeps (CPU experiment, tools/precond_block_proto.py: 95 -> 54 / 44 / 34 at d = 1 / 2 / 4)."""
    host, L = hip
    p = params_7p5(); p.solve_heating_global = False
    try:
        L.dkmc_set_x_format(0); L.dkmc_set_x_block(1); L.dkmc_set_x_poly(0)
        dev, sim, gb, _ = _fresh_device(dev_7p5, p, hip)
        dev.updateCharge(gb); dev.updatePotential(gb, p, Vd, 0); dev.updatePower(gb, p, Vd)
        rp, ci, data = host.get_last_X()
        L.dkmc_set_x_format(1)
        for tol, rbound, ibound in ((1e-6, 1.0001e-6, 1e-5), (1e-10, 3e-9, 1e-8)):
            plain = _solve(dev, gb, p, hip, 16, tol)
            assert plain["width"] == 16 and plain["fallback"] == 0
            sweeps = {0: plain["iters"]}
            for d in (1, 2, 4, 8):
                L.dkmc_set_x_poly(d)
                a = _solve(dev, gb, p, hip, 16, tol)
                # a start vector that is not zero (the preconditioned loop takes it into its right-hand side): the library's warm start from the solve
                # just made -- the same state, so the start vector already meets the stop test or needs a sweep or two
                L.dkmc_set_current_warm_start(1)
                dev.updatePower(gb, p, Vd)
                st = host.get_stats()
                b = dict(m=get(gb, "atom_virtual_potentials").copy(), im=dev.imacro, iters=st["cg_iters_X"], width=st["xb_width"], fallback=st["xb_fallback"])
                L.dkmc_set_current_warm_start(0)
                L.dkmc_set_x_poly(0)
                for rec in (a, b):
                    assert rec["width"] == 16 and rec["fallback"] == 0
                    assert _scaled_residual(rp, ci, data, rec["m"], p.G0, p.X_loop_G) <= rbound, (tol, d, rec["iters"])
                    assert abs(rec["im"] - plain["im"]) <= ibound * abs(plain["im"]), (tol, d, rec["im"], plain["im"])
                sweeps[d] = a["iters"]
                # (at 1e-10 the true residual of a converged solve sits at its rounding floor, ~1e-9 -- see test_block_cg_agrees_with_single_vector_cg_7p5 --
                # above the tolerance: a re-solve then iterates again, with or without the preconditioner)
                if tol >= 1e-6: assert b["iters"] <= 3, (tol, d, b["iters"])
            assert sweeps[8] < sweeps[4] < sweeps[2] < sweeps[1] < sweeps[0] and 3 * sweeps[8] < sweeps[0], sweeps
    finally:
        L.dkmc_set_x_block(16); L.dkmc_set_x_format(1); L.dkmc_set_cg_tolerance(1e-6); L.dkmc_set_x_poly(8); L.dkmc_set_current_warm_start(1)
