"""Parity of the HIP path (through the C ABI) against the CPU oracle and the golden fixtures.
Bit-exact for integer / index work (charges, event types, selected events, sparsity patterns);
fp64 fields within the tolerance written next to each assertion."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import params_7p5

pytestmark = pytest.mark.gpu

Vd = 5.0


def _torch():
    import torch
    return torch


@pytest.fixture(scope="module")
def hip():
    import __graft_entry__ as g
    g.build()
    from devicekmc_amd import host, lib
    assert _torch().cuda.is_available()
    return host, lib.load()


def make_pair(structure, p, hip, tol=None, warm=None):
    """(Device, KMCProcess, GPUBuffers, OracleKMC) in the state right after setLaplacePotential.  warm: start vector of the current solve
    (dkmc_set_current_warm_start); None = the library default (1: the previous solution), 0 = the reference code's G0-scaled buffer."""
    from oracle import oracle as oc
    host, L = hip
    if tol is not None:
        p.cg_tol = tol
    dev = host.Device(structure, p)
    sim = host.KMCProcess(dev, p.freq)
    gb = dev.make_gpubuf("cuda:0")
    if warm is not None:
        L.dkmc_set_current_warm_start(warm)
    dev.setLaplacePotential(gb, p, Vd)
    gb.sync_HostToGPU(dev)
    o = oc.OracleKMC(structure.element, structure.x, structure.y, structure.z, p)
    o.set_laplace_potential(Vd)
    return dev, sim, gb, o


def put(gb, name, arr):
    t = gb.t[name]
    t.copy_(_torch().as_tensor(np.ascontiguousarray(arr)).to(t.dtype))


def get(gb, name):
    return gb.t[name].cpu().numpy()


@pytest.fixture(scope="module")
def pair_2p5(cell_2p5, hip):
    from devicekmc_amd import params as pm
    p = pm.KMCParameters(); p.solve_heating_global = True
    return (p,) + make_pair(cell_2p5, p, hip, tol=1e-10)


def test_CB_edge_and_charge(pair_2p5):
    p, dev, sim, gb, o = pair_2p5
    # CB edge [J]: |error| <= 1e-9 * q*Vd (both solves converged to ||r||^2 <= 1e-20 of the scaled system)
    assert np.abs(dev.site_CB_edge - o.CB_edge).max() <= 1e-9 * p.q * Vd
    dev.updateCharge(gb); o.update_charge()
    assert np.array_equal(get(gb, "site_charge"), o.charge)              # integer: exact


def test_potential_fields(pair_2p5, hip):
    host, L = hip
    p, dev, sim, gb, o = pair_2p5
    dev.updateCharge(gb); o.update_charge()
    dev.updatePotential(gb, p, Vd, 0); o.update_potential(Vd)
    pb, pc = get(gb, "site_potential_boundary"), get(gb, "site_potential_charge")
    # both CGs stop at ||r||^2 <= 1e-20 (scaled system); K mixes conductances of 1 and 1e-8, so the solution error is
    # cond(K) * residual: tolerance 1e-7 of the applied bias.  The residual itself is checked below.
    assert np.abs(pb - o.pot_boundary).max() <= 1e-7 * Vd
    assert np.abs(pc - o.pot_charge).max() <= 1e-12 * np.abs(o.pot_charge).max()   # same sum order; erfc differs by ulps
    # contacts re-imposed exactly
    n = p.num_atoms_first_layer
    assert (pb[:n] == -Vd / 2).all() and (pb[-n:] == Vd / 2).all()
    # size-independent property: K phi = rhs on the device block (residual of the UNscaled system)
    import scipy.sparse as sp
    rp, ci, data, rhs = o._last_K
    K = sp.csr_matrix((data, ci, rp))
    res = K @ pb[n:dev.N - n] - rhs
    assert np.abs(res).max() <= 1e-9


def test_pair_sum_all_pairs_switch(cell_2p5, hip):
    """dkmc_set_pair_cutoff(0): every (site, charged site) pair is evaluated, exactly the terms the reference sums
    (potential_solver_gpu.cu:908-958); the default screening cut-off (erfc < 3.8e-20 beyond 6.5 sigma sqrt 2) changes no potential by
    more than 1e-15 of the largest one.  Randomised charges so that most sites see charged sites on both sides of the cut-off."""
    from devicekmc_amd import params as pm
    from devicekmc_amd.host import _ptr
    from devicekmc_amd.lib import check
    from oracle import oracle as oc
    host, L = hip
    p = pm.KMCParameters()
    dev, sim, gb, o = make_pair(cell_2p5, p, hip)
    rng = np.random.default_rng(11)
    q = np.where(rng.random(dev.N) < 0.08, rng.choice([-2, 2], dev.N), 0).astype(np.int32)
    put(gb, "site_charge", q); o.charge[:] = q
    oc.lib().okmc_poisson_gridless(o.N, oc._p(o.x), oc._p(o.y), oc._p(o.z), oc._p(o.lattice), 0, C.c_double(p.sigma), C.c_double(p.k),
                                   oc._p(o.charge), oc._p(o.pot_charge))
    got = {}
    try:
        for cut in (0.0, 6.5):
            L.dkmc_set_pair_cutoff(cut)
            L.dkmc_set_profiling(1)
            check(L.dkmc_poisson_gridless_gpu(0, 0, gb.N_, _ptr(gb.lattice), _ptr(gb.sigma), _ptr(gb.k), _ptr(gb.site_x), _ptr(gb.site_y),
                                              _ptr(gb.site_z), _ptr(gb.site_charge), _ptr(gb.site_potential_charge)))
            got[cut] = (get(gb, "site_potential_charge").copy(), host.get_stats()["pair_evaluated"])
    finally:
        L.dkmc_set_pair_cutoff(6.5); L.dkmc_set_profiling(0)
    scale = np.abs(o.pot_charge).max()
    nq = int((q != 0).sum())
    assert got[0.0][1] == dev.N * nq - nq                                  # every pair but the self terms
    assert 0 < got[6.5][1] < got[0.0][1]
    assert np.abs(got[0.0][0] - o.pot_charge).max() <= 1e-12 * scale
    assert np.abs(got[6.5][0] - got[0.0][0]).max() <= 1e-15 * scale


@pytest.mark.parametrize("pbc", [0, 1])
def test_pair_sum_cell_list(cell_2p5, hip, pbc):
    """The cell-list path of the pair sum (taken on the device when the box has >= 3 columns of the cut-off radius along y or z and
    >= 512 sites are charged): the 2.5 nm cell tiled 6 x 6 (338 364 sites, 153 A = 4 columns each way), 2 % of the oxygen / vacancy sites
    charged at random.  Against the all-pairs sum of the same call (dkmc_set_pair_cutoff(0)) to 1e-15 of the largest potential, against the
    oracle to 1e-12; 39 % of the pairs are distance-tested without pbc (edge columns see 2 of 4, inner ones 3 of 4, per axis), 56 % with
    pbc (the columns wrap: 3 of 4); two runs give the same bits."""
    from devicekmc_amd import params as pm, structure
    from devicekmc_amd.host import _ptr
    from devicekmc_amd.lib import check
    from oracle import oracle as oc
    host, L = hip
    k = 6
    s = structure.tile_structure(cell_2p5, k, 25.575, 25.575, 1440)
    p = pm.KMCParameters().for_tiling(k); p.pbc = bool(pbc)
    dev = host.Device(s, p, gpu_neighbors="cuda:0")
    gb = dev.make_gpubuf("cuda:0")
    rng = np.random.default_rng(5 + pbc)
    # charges only where the model puts them (oxygen / vacancy sites): the shipped cell holds one pair of coincident Ti sites per tile
    ok = (dev.site_element == pm.O_EL) | (dev.site_element == pm.VACANCY)
    q = np.where(ok & (rng.random(dev.N) < 0.02), rng.choice([-2, 2], dev.N), 0).astype(np.int32)
    put(gb, "site_charge", q)
    nq = int((q != 0).sum())
    assert nq >= 512
    want = np.zeros(dev.N)
    lat = np.asarray(p.lattice, dtype=np.float64)
    oc.lib().okmc_poisson_gridless(dev.N, oc._p(dev.site_x), oc._p(dev.site_y), oc._p(dev.site_z), oc._p(lat), int(pbc), C.c_double(p.sigma),
                                   C.c_double(p.k), oc._p(q), oc._p(want))
    got = {}
    try:
        for cut in (0.0, 6.5, 6.5):
            L.dkmc_set_pair_cutoff(cut)
            L.dkmc_set_profiling(1)
            check(L.dkmc_poisson_gridless_gpu(0, int(pbc), gb.N_, _ptr(gb.lattice), _ptr(gb.sigma), _ptr(gb.k), _ptr(gb.site_x), _ptr(gb.site_y),
                                              _ptr(gb.site_z), _ptr(gb.site_charge), _ptr(gb.site_potential_charge)))
            st = host.get_stats()
            cur = (get(gb, "site_potential_charge").copy(), st["pair_evaluated"], st["pair_tested"])
            if cut in got:
                assert np.array_equal(cur[0], got[cut][0]) and cur[1:] == got[cut][1:]          # run-to-run: same bits
            got[cut] = cur
    finally:
        L.dkmc_set_pair_cutoff(6.5); L.dkmc_set_profiling(0)
    scale = np.abs(want).max()
    assert got[0.0][2] == dev.N * nq and got[0.0][1] == dev.N * nq - nq             # all pairs: every one tested, every one but the self terms evaluated
    assert got[6.5][2] < (0.62 if pbc else 0.45) * dev.N * nq, got[6.5][2] / (dev.N * nq)      # the cell list was taken
    assert got[6.5][1] <= got[6.5][2]
    assert np.abs(got[0.0][0] - want).max() <= 1e-12 * scale
    assert np.abs(got[6.5][0] - got[0.0][0]).max() <= 1e-15 * scale


def test_event_table_and_loop_exact(pair_2p5, hip):
    host, L = hip
    from devicekmc_amd.host import _ptr
    p, dev, sim, gb, o = pair_2p5
    # identical inputs on both sides: upload the oracle's fields
    for name, arr in (("site_potential_boundary", o.pot_boundary), ("site_potential_charge", o.pot_charge),
                      ("site_charge", o.charge), ("site_element", o.element)):
        put(gb, name, arr)
    N, nn = dev.N, dev.max_num_neighbors
    torch = _torch()
    ev_type = torch.zeros(N * nn, dtype=torch.int32, device="cuda:0"); ev_prob = torch.zeros(N * nn, dtype=torch.float64, device="cuda:0")
    from devicekmc_amd.lib import check
    check(L.dkmc_build_event_list(N, nn, _ptr(gb.neigh_idx), _ptr(gb.site_layer), _ptr(gb.lattice), int(p.pbc), _ptr(gb.T_bg), _ptr(gb.freq),
                                  _ptr(gb.sigma), _ptr(gb.k), _ptr(gb.site_x), _ptr(gb.site_y), _ptr(gb.site_z),
                                  _ptr(gb.site_potential_boundary), _ptr(gb.site_potential_charge), _ptr(gb.site_element),
                                  _ptr(gb.site_charge), _ptr(ev_type), _ptr(ev_prob)))
    torch.cuda.synchronize()
    ot, op = o.build_event_list()
    assert np.array_equal(ev_type.cpu().numpy(), ot)                     # exact
    gp = ev_prob.cpu().numpy()
    nz = op > 0
    assert np.array_equal(gp > 0, nz)
    assert np.abs(gp[nz] / op[nz] - 1).max() <= 1e-11                    # exp() of arguments up to ~150: ulps * 150
    # event loop: same slots, sites, types; same stream consumption
    _, dt = sim.executeKMCStep(gb, dev, want_log=True)
    odt = o.execute_kmc_step(ev=(ot, op))
    assert np.array_equal(sim.last_event_log, o.last_events["log"])
    assert abs(dt / odt - 1) <= 1e-12
    assert o.last_events["margin"].min() > 1e-9
    assert np.array_equal(get(gb, "site_element"), o.element) and np.array_equal(get(gb, "site_charge"), o.charge)
    assert sim.random_generator.uniform() == o.rng_kmc.uniform()         # both streams at the same position


def test_current_solve(pair_2p5, hip, golden_dir):
    host, L = hip
    p, dev, sim, gb, o = pair_2p5
    r = dev.updatePower(gb, p, Vd)
    oi = o.update_power(Vd, heating=True)
    rp, ci, data = host.get_last_X()
    X = o.last_X
    assert np.array_equal(rp, X["row_ptr"]) and np.array_equal(ci, X["col"])      # pattern: exact
    scale = np.abs(X["data"]).max()
    big = np.abs(X["data"]) > 1e-300
    assert np.abs(data[big] / X["data"][big] - 1).max() <= 1e-10                   # WKB values: exp/pow rounding
    assert abs(dev.imacro / oi - 1) <= 1e-7                                        # CG to 1e-10 on both sides
    pw = get(gb, "site_power")
    assert np.abs(pw - o.power).max() <= 1e-6 * np.abs(o.power).max()
    # the reference's own sparsity dump (after the step-0 event; needs test_event_table_and_loop_exact to have run on this fixture), every
    # row: identical except 60 entries (40 only in the dump, 20 only here), all of them vacancy-vacancy pairs -- the dump's revision
    # solved the CB edge on atoms only; with that domain the dump is reproduced entry for entry (test_x_pattern_dump_log_revision below,
    # tests/test_oracle_golden.py)
    g = np.load(os.path.join(golden_dir, "x_pattern_2.5nm_step0.npz"))
    assert len(g["row_ptr"]) == len(rp)
    ael = X["ael"]
    only_dump = only_here = 0
    for r_ in range(len(rp) - 1):
        a_ = g["col_idx"][g["row_ptr"][r_]:g["row_ptr"][r_ + 1]]; b_ = ci[rp[r_]:rp[r_ + 1]]
        if len(a_) == len(b_) and np.array_equal(a_, b_):
            continue
        d1, d2 = np.setdiff1d(a_, b_), np.setdiff1d(b_, a_)
        assert r_ >= 2 and ael[r_ - 2] == 2 and (ael[np.r_[d1, d2] - 2] == 2).all(), r_
        only_dump += len(d1); only_here += len(d2)
    assert (only_dump, only_here) == (40, 20)
    st = host.get_stats()
    assert st["N_atom"] == 6421 and st["X_nnz"] == len(ci)
    # size-independent properties: X symmetric; rows without boundary terms sum to zero
    import scipy.sparse as sp
    A = sp.csr_matrix((data, ci, rp))
    assert abs(A - A.T).max() <= 1e-12 * scale
    # heat: both closed forms
    T0 = float(gb.T_bg.item())
    dev.updateTemperature(gb, p, 1e-13)
    To = o.update_temperature_global(1e-13)
    assert abs(dev.T_bg - To) <= 1e-9
    from devicekmc_amd.host import _ptr
    from devicekmc_amd.lib import check
    Cth = p.A * p.t_ox * p.c_p * 1e6
    a = -p.dissipation_constant / Cth * p.small_step + 1; b = p.dissipation_constant / Cth * p.small_step * p.background_temp
    gb.copy_Tbg_toGPU(300.0)
    check(L.dkmc_update_temperatureglobal_gpu(_ptr(gb.site_power), _ptr(gb.T_bg), gb.N_, a, b, 1e-13 / p.small_step, Cth, p.small_step))
    o.T_bg = 300.0
    To2 = o.update_temperature_global(1e-13, mode=1)
    assert abs(float(gb.T_bg.item()) - To2) <= 1e-9


def test_x_pattern_dump_log_revision(cell_2p5, hip, golden_dir):
    """The reference's own X-pattern dump (timing_2.5nm/fullmatrix_assembly, 467 336 entries, state after the first executed event), HIP
    path under `log_revision()` (CB edge solved on atoms, CG 1e-12): row pointers and column indices IDENTICAL."""
    from devicekmc_amd import params as pm
    host, L = hip
    g = np.load(os.path.join(golden_dir, "x_pattern_2.5nm_step0.npz"))
    p = pm.KMCParameters().log_revision(); p.solve_heating_global = False
    dev, sim, gb, _ = _fresh_device(cell_2p5, p, hip)
    dev.updateCharge(gb); dev.updatePotential(gb, p, Vd, 0)
    sim.executeKMCStep(gb, dev)
    dev.updatePower(gb, p, Vd)
    rp, ci, _ = host.get_last_X()
    assert len(ci) == 467336 and np.array_equal(rp, g["row_ptr"]) and np.array_equal(ci, g["col_idx"])
    L.dkmc_set_cb_edge_domain(0)


def test_superstep_sequence_2p5(cell_2p5, hip, ref_logs):
    """Five coupled supersteps at the reference's tolerance (1e-6): same event sequence as the oracle, fields close,
    and the reference CPU path's own numbers at 6 digits when both are converged."""
    from devicekmc_amd import params as pm
    host, L = hip
    p = pm.KMCParameters(); p.cg_tol = 1e-10
    dev, sim, gb, o = make_pair(cell_2p5, p, hip)
    gold = ref_logs["BASELINE.md#2 (reference CPU path run during the survey)"]["steps"]
    t = 0.0
    for k in range(5):
        dev.updateCharge(gb); dev.updatePotential(gb, p, Vd, k)
        _, dt = sim.executeKMCStep(gb, dev, want_log=True); t += dt
        dev.updatePower(gb, p, Vd)
        out = o.superstep(Vd)
        assert np.array_equal(sim.last_event_log, o.last_events["log"]), k
        assert abs(dt / out["step_time"] - 1) <= 1e-7
        assert abs(dev.imacro / out["imacro"] - 1) <= 1e-6
        if k < 2:
            assert float("%.6g" % (dev.imacro * 1e6)) == gold[k]["Current [uA]"]
            assert float("%.6g" % t) == gold[k]["KMC time"]
    gb.sync_GPUToHost(dev)
    assert np.array_equal(dev.site_element, o.element) and np.array_equal(dev.site_charge, o.charge)


def test_event_stream_exhaustion_resume(cell_2p5, hip):
    """Handing the device fewer random numbers than the step needs must give the same events (resume path)."""
    from devicekmc_amd import params as pm
    host, L = hip
    logs = []
    for batch in (64, 1):
        p = pm.KMCParameters(); p.cg_tol = 1e-8
        dev, sim, gb, o = make_pair(cell_2p5, p, hip)
        sim.batch = batch
        seq = []
        for k in range(3):
            dev.updateCharge(gb); dev.updatePotential(gb, p, Vd, k)
            _, dt = sim.executeKMCStep(gb, dev, want_log=True)
            seq.append((sim.last_event_log.copy(), dt))
        logs.append(seq)
    for (la, da), (lb, db) in zip(*logs):
        assert np.array_equal(la, lb) and da == db
    assert sum(len(l) for l, _ in logs[0]) >= 3


def test_determinism_bitwise(cell_2p5, hip):
    from devicekmc_amd import params as pm
    host, L = hip
    outs = []
    for rep in range(2):
        p = pm.KMCParameters()
        dev, sim, gb, o = make_pair(cell_2p5, p, hip)
        dev.updateCharge(gb); dev.updatePotential(gb, p, Vd, 0)
        _, dt = sim.executeKMCStep(gb, dev); dev.updatePower(gb, p, Vd)
        outs.append((get(gb, "site_potential_boundary"), get(gb, "site_potential_charge"), dt, dev.imacro))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    assert outs[0][2] == outs[1][2] and outs[0][3] == outs[1][3]


def test_cg_solver_random_spd(hip):
    """solve_sparse_CG_Jacobi on a random diagonally dominant SPD system with ragged rows (incl. rows > 192 nnz)."""
    host, L = hip
    import scipy.sparse as sp
    from devicekmc_amd.lib import check
    torch = _torch()
    rng = np.random.default_rng(0)
    m = 3000
    B = sp.random(m, m, density=0.004, random_state=1, format="lil")
    B[5, :400] = rng.random(400) * 0.01           # a long row
    B = sp.csr_matrix(B); A = B + B.T
    A = A + sp.diags(np.abs(A).sum(axis=1).A1 + 1.0)
    A = sp.csr_matrix(A); A.sort_indices()
    xs = rng.standard_normal(m); b = A @ xs
    d = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a, dtype=dt)).cuda()
    data, rp, ci = d(A.data, np.float64), d(A.indptr, np.int32), d(A.indices, np.int32)
    rhs, y = d(b, np.float64), d(np.zeros(m), np.float64)
    L.dkmc_set_cg_tolerance(1e-10)
    it, rr = C.c_int(0), C.c_double(0)
    check(L.dkmc_solve_sparse_CG_Jacobi(data.data_ptr(), rp.data_ptr(), ci.data_ptr(), A.nnz, m, rhs.data_ptr(), y.data_ptr(), C.byref(it), C.byref(rr)))
    torch.cuda.synchronize()
    assert it.value > 0 and rr.value <= 1e-20
    assert np.abs(y.cpu().numpy() - xs).max() <= 1e-8
    # m = 0 and already-converged guess
    y2 = d(xs, np.float64); rhs2 = d(b, np.float64); data2 = d(A.data, np.float64)
    L.dkmc_set_cg_tolerance(1e-3)
    check(L.dkmc_solve_sparse_CG_Jacobi(data2.data_ptr(), rp.data_ptr(), ci.data_ptr(), A.nnz, m, rhs2.data_ptr(), y2.data_ptr(), C.byref(it), C.byref(rr)))
    assert it.value == 0
    L.dkmc_set_cg_tolerance(1e-6)


def test_pbc_fields(cell_2p5, hip):
    """pbc = 1 (y,z wrap with round()): pair potential and event table against the oracle on a sub-block."""
    from devicekmc_amd import params as pm, structure
    from oracle import oracle as oc
    host, L = hip
    from devicekmc_amd.host import _ptr
    from devicekmc_amd.lib import check
    n = 3000
    sub = structure.Structure(cell_2p5.element[1440:1440 + n].copy(), cell_2p5.x[1440:1440 + n].copy(),
                              cell_2p5.y[1440:1440 + n].copy(), cell_2p5.z[1440:1440 + n].copy(), {})
    p = pm.KMCParameters(); p.pbc = True; p.num_atoms_first_layer = 10; p.num_atoms_contact = 10
    dev = host.Device(sub, p)
    gb = dev.make_gpubuf("cuda:0")
    o = oc.OracleKMC(sub.element, sub.x, sub.y, sub.z, p)
    assert np.array_equal(dev.neigh_idx, o.neigh)
    dev.updateCharge(gb); o.update_charge()
    assert np.array_equal(get(gb, "site_charge"), o.charge)
    check(L.dkmc_poisson_gridless_gpu(0, 1, gb.N_, _ptr(gb.lattice), _ptr(gb.sigma), _ptr(gb.k), _ptr(gb.site_x), _ptr(gb.site_y),
                                      _ptr(gb.site_z), _ptr(gb.site_charge), _ptr(gb.site_potential_charge)))
    oc.lib().okmc_poisson_gridless(o.N, oc._p(o.x), oc._p(o.y), oc._p(o.z), oc._p(o.lattice), 1, C.c_double(p.sigma), C.c_double(p.k),
                                   oc._p(o.charge), oc._p(o.pot_charge))
    pc = get(gb, "site_potential_charge")
    assert np.abs(pc - o.pot_charge).max() <= 1e-12 * max(np.abs(o.pot_charge).max(), 1e-30)


def test_kmc_time_vs_reference_cuda_log_7p5(dev_7p5, hip, ref_logs):
    """configs[1]: 85 071 sites.  KMC time of all 19 logged supersteps against the reference's own CUDA-path log, and the same
    event sequence as the oracle.  CG tolerance 1e-12 = the tolerance the log was produced with ("used to be 1e-12",
    iterative_solvers_gpu.cu:322; tests/test_oracle_golden.py::test_cuda_path_log_7p5nm): the log's 6 printed digits are met."""
    host, L = hip
    gold = ref_logs["timing_7.5nm/output_noguess.txt"]["steps"]
    p = params_7p5(); p.cg_tol = 1e-12; p.solve_current = False
    dev, sim, gb, o = make_pair(dev_7p5, p, hip)
    t = 0.0
    for k in range(3):
        dev.updateCharge(gb); dev.updatePotential(gb, p, Vd, k)
        _, dt = sim.executeKMCStep(gb, dev, want_log=True); t += dt
        out = o.superstep(Vd)
        assert np.array_equal(sim.last_event_log, o.last_events["log"]), k
        assert abs(t / gold[k]["KMC time"] - 1) < 1e-5, (k, t, gold[k])
        pb = get(gb, "site_potential_boundary")
        assert np.abs(pb - o.pot_boundary).max() <= 1e-6 * Vd
    # the remaining 16 logged supersteps, HIP path alone against the reference's log
    for k in range(3, len(gold)):
        dev.updateCharge(gb); dev.updatePotential(gb, p, Vd, k)
        _, dt = sim.executeKMCStep(gb, dev); t += dt
        assert abs(t / gold[k]["KMC time"] - 1) < 1e-5, (k, t, gold[k])


@pytest.mark.parametrize("domain", ["sites", "atoms"])
def test_current_7p5_properties(dev_7p5, hip, domain, ref_logs):
    """Full-size current solve under both CB-edge domains (snapshot source: every site; log revision: atoms only): CB edge and X
    pattern identical to the oracle's, X symmetric, I_macro equal to the oracle's.  Under the log revision the current is the
    reference log's 11.8834 uA to its six printed digits (tests/test_oracle_golden.py::test_current_7p5nm_vs_log)."""
    host, L = hip
    p = params_7p5(); p.cg_tol = 1e-10; p.cb_edge_domain = domain
    dev, sim, gb, o = make_pair(dev_7p5, p, hip)
    assert np.abs(dev.site_CB_edge - o.CB_edge).max() <= 1e-8 * p.q * Vd
    if domain == "atoms":
        assert (dev.site_CB_edge[(dev.site_element == 0) | (dev.site_element == 1)] == 0).all()      # unlinked interstitials
    dev.updateCharge(gb); dev.updatePotential(gb, p, Vd, 0)
    o.update_charge(); o.update_potential(Vd)
    _, dt = sim.executeKMCStep(gb, dev, want_log=True); o.execute_kmc_step()
    assert np.array_equal(sim.last_event_log, o.last_events["log"])
    dev.updatePower(gb, p, Vd)
    oi = o.update_power(Vd)
    rp, ci, data = host.get_last_X()
    assert np.array_equal(rp, o.last_X["row_ptr"]) and np.array_equal(ci, o.last_X["col"])
    assert abs(dev.imacro / oi - 1) <= 1e-6
    if domain == "atoms":
        assert abs(dev.imacro * 1e6 - ref_logs["timing_7.5nm/output_noguess.txt"]["steps"][0]["Current [uA]"]) <= 1e-4
    import scipy.sparse as sp
    A = sp.csr_matrix((data, ci, rp))
    assert abs(A - A.T).max() <= 1e-12 * np.abs(data).max()
    L.dkmc_set_cb_edge_domain(0)


def test_reference_log_7p5_currents(dev_7p5, hip, ref_logs):
    """The reference's own CUDA-path log of configs[1] (timing_7.5nm/output_noguess.txt): KMC time AND Current [uA] of all 19 logged
    supersteps, HIP path alone, full coupled step (charge, potential, events, current), under `log_revision()` = CG tolerance 1e-12
    ("used to be 1e-12", iterative_solvers_gpu.cu:322) + CB edge solved on atoms.  Both columns to the log's six printed digits."""
    host, L = hip
    gold = ref_logs["timing_7.5nm/output_noguess.txt"]["steps"]
    p = params_7p5().log_revision()
    dev = host.Device(dev_7p5, p); sim = host.KMCProcess(dev, p.freq)
    gb = dev.make_gpubuf("cuda:0")
    dev.setLaplacePotential(gb, p, Vd)
    t = 0.0
    worst_i = worst_t = 0.0
    for k in range(len(gold)):
        dev.updateCharge(gb); dev.updatePotential(gb, p, Vd, k)
        _, dt = sim.executeKMCStep(gb, dev); t += dt
        dev.updatePower(gb, p, Vd)
        worst_t = max(worst_t, abs(t / gold[k]["KMC time"] - 1))
        worst_i = max(worst_i, abs(dev.imacro * 1e6 - gold[k]["Current [uA]"]))
        assert abs(t / gold[k]["KMC time"] - 1) < 5e-6, (k, t, gold[k])
        assert abs(dev.imacro * 1e6 - gold[k]["Current [uA]"]) <= 1e-4, (k, dev.imacro * 1e6, gold[k])     # 6th printed digit +- 1
    print("worst |dI| %.2e uA, worst rel dt %.1e over %d steps" % (worst_i, worst_t, len(gold)))
    L.dkmc_set_cb_edge_domain(0)


def _d2h_i32(ptr, n):
    """int32 array behind a raw device pointer of GPUBuffers (the pattern arrays are allocated by the library, not by torch)."""
    _torch().cuda.synchronize()
    hiprt = C.CDLL("libamdhip64.so")
    out = np.empty(n, dtype=np.int32)
    rc = hiprt.hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), C.c_size_t(4 * n), 2)       # hipMemcpyDeviceToHost
    assert rc == 0
    return out


@pytest.mark.parametrize("which", ["2.5nm", "7.5nm"])
def test_K_sparsity_patterns_published_in_gpubuffers(cell_2p5, dev_7p5, hip, which):
    """initialize_sparsity (iterative_solvers_gpu.cu:96-109, Assemble_K_sparsity :2158-2208): the six arrays it leaves in GPUBuffers
    (gpu_buffers.h:40-46) -- CSR of the device block incl. the diagonal, CSR of device x left contact and device x right contact --
    read back and compared entry by entry with the oracle's patterns (okmc_k_pattern)."""
    from devicekmc_amd import params as pm
    from oracle import oracle as oc
    host, L = hip
    s, p = (cell_2p5, pm.KMCParameters()) if which == "2.5nm" else (dev_7p5, params_7p5())
    dev = host.Device(s, p)
    gb = dev.make_gpubuf("cuda:0")
    o = oc.OracleKMC(s.element, s.x, s.y, s.z, p)
    nl, m, pats = o.initialize_sparsity()
    assert m == dev.N - 2 * p.num_atoms_first_layer
    c = gb.c
    for (rp_name, ci_name, nnz_name), (rp, ci) in zip((("Device_row_ptr_d", "Device_col_indices_d", "Device_nnz"),
                                                       ("contact_left_row_ptr", "contact_left_col_indices", "contact_left_nnz"),
                                                       ("contact_right_row_ptr", "contact_right_col_indices", "contact_right_nnz")), pats):
        nnz = getattr(c, nnz_name)
        assert nnz == len(ci) == rp[m], (rp_name, nnz, len(ci))
        assert np.array_equal(_d2h_i32(getattr(c, rp_name), m + 1), rp), rp_name
        if nnz:
            assert np.array_equal(_d2h_i32(getattr(c, ci_name), nnz), ci), ci_name
    # shape facts of the reference layout: diagonal present in every row of the device block, columns ascending
    rp, ci = pats[0]
    rows = np.repeat(np.arange(m), np.diff(rp))
    assert (np.bincount(rows[ci == rows], minlength=m) == 1).all()
    assert (np.diff(ci)[np.diff(rows) == 0] > 0).all()


def test_neighbor_index_gpu(cell_2p5, dev_7p5, hip):
    """the HIP cell-list builder reproduces Device.cpp:98-136 + :69-80 (ascending j, -1 padding, nn = max)."""
    from devicekmc_amd import params as pm, structure
    from oracle import oracle as oc
    host, L = hip
    p = pm.KMCParameters()
    for s, lat in ((cell_2p5, p.lattice), (dev_7p5, params_7p5().lattice)):
        neigh, nn = host.build_neighbor_index_gpu(s.x, s.y, s.z, lat, False, p.nn_dist)
        on, onn = oc.build_neighbors(s.x, s.y, s.z, np.array(lat), False, p.nn_dist)
        assert nn == onn and np.array_equal(neigh, on)
    sub = structure.Structure(cell_2p5.element[:3000], cell_2p5.x[:3000], cell_2p5.y[:3000], cell_2p5.z[:3000], {})
    neigh, nn = host.build_neighbor_index_gpu(sub.x, sub.y, sub.z, p.lattice, True, p.nn_dist)
    on, onn = oc.build_neighbors(sub.x, sub.y, sub.z, np.array(p.lattice), True, p.nn_dist)
    assert nn == onn and np.array_equal(neigh, on)
    # a Device built on it is identical to the host-built one
    d1 = host.Device(cell_2p5, p); d2 = host.Device(cell_2p5, p, gpu_neighbors="cuda:0")
    assert d1.max_num_neighbors == d2.max_num_neighbors and np.array_equal(d1.neigh_idx, d2.neigh_idx)


def test_superstep_pbc_and_heating(cell_2p5, hip):
    """The 2.5 nm cell is periodic in y,z with the period of parameters.txt: run it with pbc = 1 (wrapped distances in the
    neighbour list, pair sum, event table and X) and global heating on; same events, T_bg and current as the oracle."""
    from devicekmc_amd import params as pm
    host, L = hip
    p = pm.KMCParameters(); p.pbc = True; p.solve_heating_global = True; p.cg_tol = 1e-9
    dev, sim, gb, o = make_pair(cell_2p5, p, hip)
    assert dev.max_num_neighbors == o.nn and np.array_equal(dev.neigh_idx, o.neigh)
    assert (dev.neigh_idx >= 0).sum() > 9399 * 23.3            # periodic images add neighbours at the lateral faces
    for k in range(3):
        dev.updateCharge(gb); dev.updatePotential(gb, p, Vd, k)
        _, dt = sim.executeKMCStep(gb, dev, want_log=True)
        dev.updatePower(gb, p, Vd); dev.updateTemperature(gb, p, dt)
        out = o.superstep(Vd)
        assert np.array_equal(sim.last_event_log, o.last_events["log"]), k
        assert abs(dt / out["step_time"] - 1) <= 1e-7
        assert abs(dev.imacro / out["imacro"] - 1) <= 1e-6
        assert abs(dev.T_bg - out["T_bg"]) <= 1e-9
    rp, ci, data = host.get_last_X()
    assert np.array_equal(rp, o.last_X["row_ptr"]) and np.array_equal(ci, o.last_X["col"])


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_randomised_event_engine(cell_2p5, hip, seed):
    """All four event classes (generation, recombination, vacancy and ion diffusion) on randomised site states and
    potentials: event table exact, long event sequences identical to the oracle, charges exact."""
    from devicekmc_amd import params as pm
    from devicekmc_amd.host import _ptr
    from devicekmc_amd.lib import check
    host, L = hip
    torch = _torch()
    rng = np.random.default_rng(seed)
    p = pm.KMCParameters(); p.rnd_seed_kmc = seed
    dev, sim, gb, o = make_pair(cell_2p5, p, hip, tol=1e-6)
    el = o.element.copy()
    ox = np.nonzero(el == pm.O_EL)[0]; dd = np.nonzero(el == pm.DEFECT)[0]
    el[rng.choice(ox, 300, replace=False)] = pm.VACANCY                    # many vacancies, some clustered
    el[rng.choice(dd, 200, replace=False)] = pm.OXYGEN_DEFECT              # oxygen ions on interstitial sites
    o.element[:] = el; put(gb, "site_element", el)
    dev.updateCharge(gb); o.update_charge()
    assert np.array_equal(get(gb, "site_charge"), o.charge)
    # rough potentials so that every class gets appreciable rates (|dphi| up to ~2 V between neighbours)
    pb = np.linspace(-2.5, 2.5, dev.N)[np.argsort(np.argsort(o.x))] + 0.3 * rng.standard_normal(dev.N)
    pc = 0.2 * rng.standard_normal(dev.N)
    o.pot_boundary[:] = pb; o.pot_charge[:] = pc
    put(gb, "site_potential_boundary", pb); put(gb, "site_potential_charge", pc)
    N, nn = dev.N, dev.max_num_neighbors
    ev_type = torch.zeros(N * nn, dtype=torch.int32, device="cuda:0"); ev_prob = torch.zeros(N * nn, dtype=torch.float64, device="cuda:0")
    check(L.dkmc_build_event_list(N, nn, _ptr(gb.neigh_idx), _ptr(gb.site_layer), _ptr(gb.lattice), int(p.pbc), _ptr(gb.T_bg), _ptr(gb.freq),
                                  _ptr(gb.sigma), _ptr(gb.k), _ptr(gb.site_x), _ptr(gb.site_y), _ptr(gb.site_z),
                                  _ptr(gb.site_potential_boundary), _ptr(gb.site_potential_charge), _ptr(gb.site_element),
                                  _ptr(gb.site_charge), _ptr(ev_type), _ptr(ev_prob)))
    torch.cuda.synchronize()
    ot, op = o.build_event_list()
    gt = ev_type.cpu().numpy()
    assert np.array_equal(gt, ot)
    for cls in range(4):
        assert (ot == cls).sum() > 0, cls                                 # every class is present
    gp = ev_prob.cpu().numpy(); nz = op > 0
    fin = nz & np.isfinite(op)
    assert np.array_equal(np.isfinite(gp), np.isfinite(op))
    assert np.abs(gp[fin] / op[fin] - 1).max() <= 1e-10
    if np.isfinite(op).all():
        # event loop: with rates this large hundreds of events run before a waiting time exceeds 1/freq
        sim.batch = 64
        _, dt = sim.executeKMCStep(gb, dev, want_log=True)
        odt = o.execute_kmc_step(ev=(ot, op))
        n = o.last_events["n"]
        assert n >= 100 and (np.bincount(o.last_events["log"][:, 3], minlength=4)[:4] > 0).all()   # all four classes EXECUTED
        safe = o.last_events["margin"] > 1e-9                              # draws within rounding distance of a bucket edge are exempt
        k = n if safe.all() else int(np.argmin(safe))
        assert np.array_equal(sim.last_event_log[:k], o.last_events["log"][:k])
        if safe.all():
            assert len(sim.last_event_log) == n and abs(dt / odt - 1) <= 1e-9
            assert np.array_equal(get(gb, "site_element"), o.element) and np.array_equal(get(gb, "site_charge"), o.charge)


def test_current_solve_randomised_negative_bias(cell_2p5, hip):
    """X assembly / solve / power on a state with 300 vacancies (clusters of neutral vacancies -> high_G links) at a
    NEGATIVE bias (the `ical > 0 && Vd < 0` branch of the power formula): pattern exact, values, I_macro, power."""
    from devicekmc_amd import params as pm
    from oracle import oracle as oc
    host, L = hip
    V = -3.0
    p = pm.KMCParameters(); p.cg_tol = 1e-10; p.solve_heating_global = True
    dev = host.Device(cell_2p5, p); sim = host.KMCProcess(dev, p.freq)
    gb = dev.make_gpubuf("cuda:0")
    rng = np.random.default_rng(5)
    el = dev.site_element.copy()
    ox = np.nonzero(el == pm.O_EL)[0]
    el[rng.choice(ox, 300, replace=False)] = pm.VACANCY
    dev.site_element = el
    dev.setLaplacePotential(gb, p, V)
    gb.sync_HostToGPU(dev)
    o = oc.OracleKMC(cell_2p5.element, cell_2p5.x, cell_2p5.y, cell_2p5.z, p)
    o.element[:] = el
    o.set_laplace_potential(V)
    dev.updateCharge(gb); o.update_charge()
    assert np.array_equal(get(gb, "site_charge"), o.charge)
    assert ((o.element == pm.VACANCY) & (o.charge == 0)).sum() > 20          # neutral (clustered / contact-adjacent) vacancies exist
    dev.updatePower(gb, p, V)
    oi = o.update_power(V, heating=True)
    rp, ci, data = host.get_last_X()
    X = o.last_X
    assert np.array_equal(rp, X["row_ptr"]) and np.array_equal(ci, X["col"])
    big = np.abs(X["data"]) > 1e-300
    assert np.abs(data[big] / X["data"][big] - 1).max() <= 1e-10
    assert oi < 0 and abs(dev.imacro / oi - 1) <= 1e-6
    pw = get(gb, "site_power")
    assert np.abs(o.power).max() > 0 and np.abs(pw - o.power).max() <= 1e-6 * np.abs(o.power).max()


def test_crossbar_log(hip, ref_logs, golden_dir):
    """structures/crossbars/timing_10nm_5pitch (110 813 sites, V = 1, solve_current = 0): a second geometry.  Same events as
    the oracle; KMC time of ALL 13 logged supersteps of the reference's CUDA run to its 6 printed digits, at the CG tolerance the
    log was made with (1e-12, tests/test_oracle_golden.py)."""
    from devicekmc_amd import params as pm, structure
    from oracle import oracle as oc
    host, L = hip
    gold = ref_logs["crossbars/timing_10nm_5pitch/output_initial.txt"]["steps"]
    s = structure.load_structure(os.path.join(golden_dir, "crossbar_10nm_5pitch.npz"))
    p = pm.KMCParameters(rnd_seed=5, lattice=tuple(s.meta["lattice"]), num_atoms_first_layer=144, num_atoms_contact=11520)
    p.cg_tol = 1e-12; p.solve_current = False
    V = 1.0
    dev = host.Device(s, p, gpu_neighbors="cuda:0"); sim = host.KMCProcess(dev, p.freq)
    gb = dev.make_gpubuf("cuda:0")
    o = oc.OracleKMC(s.element, s.x, s.y, s.z, p)
    assert dev.N == 110813 and np.array_equal(dev.neigh_idx, o.neigh) and np.array_equal(dev.site_element, o.element)
    t = 0.0
    for k in range(len(gold)):
        dev.updateCharge(gb); dev.updatePotential(gb, p, V, k)
        _, dt = sim.executeKMCStep(gb, dev, want_log=True); t += dt
        if k < 3:
            o.superstep(V)
            assert np.array_equal(sim.last_event_log, o.last_events["log"]), k
            assert np.array_equal(get(gb, "site_charge"), o.charge)
        assert abs(t / gold[k]["KMC time"] - 1) < 1e-5, (k, t, gold[k])


def test_error_paths(cell_2p5, hip):
    """Errors are reported through return codes + dkmc_last_error (the shim prints and carries on like the reference)."""
    from devicekmc_amd import params as pm
    from devicekmc_amd.lib import DeviceKMCError, check
    host, L = hip
    p = pm.KMCParameters()
    dev = host.Device(cell_2p5, p)
    gb = host.GPUBuffers(p.layers, dev.site_layer, p.freq, dev.N, dev.N_atom, dev.site_x, dev.site_y, dev.site_z, dev.max_num_neighbors,
                         dev.sigma, dev.k, dev.lattice, dev.neigh_idx, list(p.metals), "cuda:0")
    gb.sync_HostToGPU(dev)
    with pytest.raises(DeviceKMCError, match="sparsity"):            # solve before initialize_sparsity
        dev.updatePotential(gb, p, Vd, 0)
    check(L.dkmc_initialize_sparsity(C.byref(gb.c), 0, p.nn_dist, p.num_atoms_first_layer))
    with pytest.raises(DeviceKMCError, match="contact sizes"):      # solve with other contact sizes than the pattern
        check(L.dkmc_background_potential_gpu_sparse(C.byref(gb.c), dev.N, 100, 100, Vd, 0, p.high_G, p.low_G, p.nn_dist, 2, 0))
    dev.updatePotential(gb, p, Vd, 0)
    # a device without any possible event: every rate is zero -> reported, not a hang
    put(gb, "site_element", np.full(dev.N, pm.Hf_EL, dtype=np.int32))
    sim = host.KMCProcess(dev, p.freq)
    with pytest.raises(DeviceKMCError, match="positive rate"):
        sim.executeKMCStep(gb, dev)
    assert L.dkmc_last_error() == b""                                 # cleared by the raising wrapper


def test_local_temperature_model(cell_2p5, hip):
    """Local heating (heat_solver.cpp:40-246, 286-308, 354-513; host-only and dense in the reference): the sparse device
    solves against the dense numpy restatement (explicit inverses), transient sub-steps and steady state.  The oracle for
    this model is unpinned (no reference fixture exists); tolerance 1e-7 K on temperatures of 300..1e3 K (CG stops at a
    scaled residual of 1e-10, the dense inverse carries ~1e-12 relative error itself)."""
    from devicekmc_amd import params as pm
    from oracle import heat_local as hl
    host, L = hip
    p = pm.KMCParameters(); p.solve_heating_local = True
    dev, sim, gb, o = make_pair(cell_2p5, p, hip, tol=1e-10)
    dev.updateCharge(gb); dev.updatePotential(gb, p, Vd, 0)
    sim.executeKMCStep(gb, dev)
    dev.updatePower(gb, p, Vd)                                    # site_power of a real step
    power = get(gb, "site_power").copy()
    assert power.max() > 0
    element = get(gb, "site_element").copy()
    dev.site_element = element                                    # host copy after the step's events (contact counting)
    dev.constructLaplacian(gb, p)
    orc = hl.LocalHeatOracle(element, dev.neigh_idx, p.metals, p.num_atoms_contact, p.nn_dist, p.delta, p.delta_t, p.tau,
                             p.k_th_interface, p.k_th_metal)
    assert (dev.N_left_tot, dev.N_right_tot, dev.N_interface) == (orc.N_left_tot, orc.N_right_tot, orc.N_interface)
    assert 0 < orc.N_interface < dev.N
    # the power of one step heats by micro-kelvins; the model is linear in the power: scale it so that one transient update
    # raises the hottest site by 30 K and the comparison resolves kelvins
    trial = np.full(dev.N, p.background_temp)
    orc.update_local_temperature(trial, power, element, p.background_temp, p.delta_t, p.tau, p.k_th_interface, p.k_th_vacancies,
                                 p.num_atoms_contact)
    power *= 30.0 / np.abs(trial - p.background_temp).max()
    put(gb, "site_power", power)
    T_ref = np.full(dev.N, p.background_temp)
    put(gb, "site_temperature", T_ref)
    # (a) transient: 2.5 delta_t -> 3 solves, then again from the heated state (warm history)
    for step_time in (2.5 * p.delta_t, 0.3 * p.delta_t):
        want_Tbg, want_n, want_ss = orc.update_temperature_local(T_ref, power, element, step_time, p.background_temp, p.delta_t, p.tau,
                                                                 p.k_th_interface, p.k_th_vacancies, p.num_atoms_contact)
        res = dev.updateTemperature(gb, p, step_time)
        T = get(gb, "site_temperature")
        assert (dev.last_heat_solves, dev.last_heat_steady) == (want_n, bool(want_ss))
        assert np.abs(T - T_ref).max() <= 1e-7, np.abs(T - T_ref).max()
        assert abs(res["Global temperature [K]"] - want_Tbg) <= 1e-9
        assert abs(float(gb.T_bg.item()) - want_Tbg) <= 1e-9
    assert np.abs(T_ref - p.background_temp).max() > 1.0          # the comparison is not vacuous
    # (b) steady state: step_time > 1e3 delta_t
    want_Tbg, want_n, want_ss = orc.update_temperature_local(T_ref, power, element, 2e3 * p.delta_t, p.background_temp, p.delta_t, p.tau,
                                                             p.k_th_interface, p.k_th_vacancies, p.num_atoms_contact)
    res = dev.updateTemperature(gb, p, 2e3 * p.delta_t)
    T = get(gb, "site_temperature")
    assert dev.last_heat_steady and want_ss == 1
    rel = np.abs(T - T_ref).max() / max(1.0, np.abs(T_ref - p.background_temp).max())
    assert rel <= 1e-8, rel                                       # steady-state temperatures can be large: relative to the rise
    assert abs(res["Global temperature [K]"] - want_Tbg) <= 1e-8 * max(1.0, abs(want_Tbg))
    # sites outside the interface keep their temperature
    out = np.r_[0:orc.N_left_tot, dev.N - orc.N_right_tot:dev.N]
    assert np.all(T[out] == p.background_temp)
    put(gb, "site_temperature", np.full(dev.N, p.background_temp))


def _scaled_residual(rp, ci, data, m_scaled, G0, loop_G, Vd=Vd):
    """||S (X m - b)||_2 with S = diag(X)^-1/2: the quantity solve_sparse_CG_Jacobi's stop test bounds (iterative_solvers_gpu.cu:448)."""
    import scipy.sparse as sp
    n = len(rp) - 1
    X = sp.csr_matrix((data, ci, rp), shape=(n, n))
    b = np.zeros(n); b[0] = -loop_G * Vd; b[1] = loop_G * Vd
    s = 1.0 / np.sqrt(X.diagonal())
    return float(np.linalg.norm(s * (X @ (m_scaled[:n] / G0) - b)))


def test_tiled_X_agrees_with_csr_X(dev_7p5, hip):
    """Current solve of the 85 k-site device on the tiled X (default) against the same solve on the CSR X, both asked for a scaled
    residual of 1e-10.  Rounding-independent criteria: both solutions satisfy the stop test in the TRUE scaled residual of the
    CSR matrix (computed on the host), same entry count, I_macro / solution / dissipated power equal to 1e-8 relative (the
    paths differ by rounding only; cond(X) amplifies it), and the tiles carry the matrix (>= 90 % of X)."""
    host, L = hip
    p = params_7p5()
    p.cg_tol = 1e-10
    out = {}
    try:
        for fmt in (0, 1):
            L.dkmc_set_x_format(fmt)
            p.solve_heating_global = False
            dev, sim, gb, _ = _fresh_device(dev_7p5, p, hip)
            dev.updateCharge(gb); dev.updatePotential(gb, p, Vd, 0)
            dev.updatePower(gb, p, Vd)
            st = host.get_stats()
            rec = dict(imacro=dev.imacro, m=get(gb, "atom_virtual_potentials").copy(), nnz=st["X_nnz"], tiles=st["spmv_tiles"],
                       tile_entries=st["spmv_tile_entries"], iters=st["cg_iters_X"])
            if fmt == 0:
                rec["X"] = host.get_last_X()
            p.solve_heating_global = True
            put(gb, "atom_virtual_potentials", np.zeros(dev.N_atom + 2))
            dev.updatePower(gb, p, Vd)
            rec["power"] = get(gb, "site_power").copy()
            out[fmt] = rec
    finally:
        L.dkmc_set_x_format(1)
    a, b = out[0], out[1]
    assert a["tiles"] == 0 and b["tiles"] > 0 and 2 * b["tile_entries"] >= 0.9 * b["nnz"] and a["nnz"] == b["nnz"]
    rp, ci, data = a["X"]
    for rec in (a, b):
        assert _scaled_residual(rp, ci, data, rec["m"], p.G0, p.X_loop_G) <= 10 * p.cg_tol, rec["iters"]
    assert abs(b["imacro"] - a["imacro"]) <= 1e-8 * abs(a["imacro"])
    n = min(len(a["m"]), len(b["m"]))
    assert np.abs(b["m"][:n] - a["m"][:n]).max() <= 1e-8 * np.abs(a["m"][:n]).max()
    assert np.abs(b["power"] - a["power"]).max() <= 1e-8 * np.abs(a["power"]).max()


def test_uncached_coefficients_same_bits(dev_7p5, hip):
    """The tunnelling-coefficient cache against its fallback (a device whose cache does not fit evaluates the contact->trap integrals
    directly while the tiles are filled): with the budget forced to zero the same two supersteps give the same bits -- solution, current,
    dissipated power -- as with the cache, on the tiled and on the CSR form of X."""
    host, L = hip
    out = {}
    try:
        for fmt in (1, 0):
            for budget in (-1, 0):
                L.dkmc_set_x_format(fmt)
                L.dkmc_set_tcache_budget(budget)
                p = params_7p5(); p.cg_tol = 1e-8; p.solve_heating_global = True
                dev, sim, gb, _ = _fresh_device(dev_7p5, p, hip)          # setLaplacePotential invalidates the cache: rebuilt with this budget
                rec = []
                for k in range(2):
                    dev.updateCharge(gb); dev.updatePotential(gb, p, Vd, k)
                    sim.executeKMCStep(gb, dev)
                    dev.updatePower(gb, p, Vd)
                    rec.append((dev.imacro, get(gb, "atom_virtual_potentials").copy(), get(gb, "site_power").copy(), host.get_stats()["cg_iters_X"]))
                out[(fmt, budget)] = rec
    finally:
        L.dkmc_set_x_format(1); L.dkmc_set_tcache_budget(-1)
    for fmt in (1, 0):
        for (ia, ma, pa, na), (ib, mb, pb_, nb) in zip(out[(fmt, -1)], out[(fmt, 0)]):
            assert ia == ib and na == nb and np.array_equal(ma, mb) and np.array_equal(pa, pb_), fmt


def test_default_tolerance_superstep_7p5(dev_7p5, hip):
    """The configuration bench.py times: 85 071 sites at the library's DEFAULT CG tolerance 1e-6 (the snapshot's hard-coded value,
    iterative_solvers_gpu.cu:322).  At a loose tolerance two correct CG implementations stop on different iterates (cond(K) ~ 1e8,
    cond(X) ~ 1e13), so the fields are not compared value by value with the oracle's; checked instead, over three coupled supersteps:
      * K: the TRUE residual of the unscaled system K phi = rhs (the oracle's K, assembled from the same elements / charges) is
        bounded by the stop test: ||S (K phi - rhs)||_2 <= 10 tol;
      * events: the oracle, FED the GPU's two potentials, builds the same event table and selects the same (slot, i, j, type)
        sequence and the same dt -- the event path is exact given its inputs;
      * X: the GPU's solution meets the stop test in the true scaled residual of the oracle's X, assembled from the same state
        (||S (X m - b)||_2 <= 10 tol), and the pattern is identical."""
    import scipy.sparse as sp
    host, L = hip
    p = params_7p5(); p.solve_heating_global = False
    assert p.cg_tol == 1e-6 and p.cb_edge_domain == "sites"
    dev, sim, gb, o = make_pair(dev_7p5, p, hip)
    nl = p.num_atoms_first_layer
    for k in range(3):
        dev.updateCharge(gb); dev.updatePotential(gb, p, Vd, k)
        o.update_charge()
        assert np.array_equal(get(gb, "site_charge"), o.charge)
        pb, pc = get(gb, "site_potential_boundary"), get(gb, "site_potential_charge")
        # K of this state from the oracle's assembly (values only; nothing is solved on the CPU)
        o._solve_K(o.pot_boundary.copy(), -Vd / 2, Vd / 2, 0, 1e300)
        rp, ci, data, rhs = o._last_K
        K = sp.csr_matrix((data, ci, rp))
        sK = 1.0 / np.sqrt(K.diagonal())
        assert np.linalg.norm(sK * (K @ pb[nl:dev.N - nl] - rhs)) <= 10 * p.cg_tol, k
        # the oracle takes the GPU's potentials as its own and runs the event path
        o.pot_boundary[:] = pb; o.pot_charge[:] = pc
        _, dt = sim.executeKMCStep(gb, dev, want_log=True)
        odt = o.execute_kmc_step()
        assert np.array_equal(sim.last_event_log, o.last_events["log"]), k
        assert abs(dt - odt) <= 1e-12 * odt
        assert o.last_events["margin"].min() > 1e-9
        assert np.array_equal(get(gb, "site_element"), o.element)
        dev.updatePower(gb, p, Vd)
        X = o.assemble_X()
        m = get(gb, "atom_virtual_potentials")
        if k == 0:
            hrp, hci, _ = host.get_last_X()
            assert np.array_equal(hrp, X["row_ptr"]) and np.array_equal(hci, X["col"])
        assert host.get_stats()["X_nnz"] == len(X["col"])
        assert _scaled_residual(X["row_ptr"], X["col"], X["data"], m, p.G0, p.X_loop_G) <= 10 * p.cg_tol, k


@pytest.mark.parametrize("case", ["small_bias", "few_vacancies", "no_vacancies", "negative_bias", "empty_S"])
def test_tiled_X_edge_cases(cell_2p5, hip, case):
    """Tiled X against CSR X where the tunnelling block degenerates: |Vd| so small that contact-contact pairs fall below the 0.01 eV
    threshold (only ragged, sparse cells are left), very few / no vacancies (S = contact metals only), negative bias, and an EMPTY
    tunnelling set (no vacancies and an empty inner-contact window; X is its neighbour part alone).  Same
    column-sorted CSR out of both (pattern exact, values to 1e-12), same I_macro and dissipated power to 1e-8."""
    host, L = hip
    from devicekmc_amd import params as pm
    p = pm.KMCParameters(); p.solve_heating_global = True; p.cg_tol = 1e-10
    vd = {"small_bias": 0.03, "negative_bias": -3.0}.get(case, Vd)
    if case == "few_vacancies":
        p.initial_vacancy_concentration = 0.002
    if case in ("no_vacancies", "empty_S"):
        p.initial_vacancy_concentration = 0.0
    if case == "empty_S":
        p.num_layers_contact = 100          # (nlc - 1) * n_src exceeds the atom count: the inner-contact window is empty
    out = {}
    try:
        # two STORAGE forms of X under the same CG: the single-vector loop on both sides (the CSR form has no block loop; block against
        # single-vector is tests/test_gpu_block_cg.py, with the tolerances two different iterate sequences need)
        L.dkmc_set_x_block(1)
        for fmt in (0, 1):
            L.dkmc_set_x_format(fmt)
            dev = host.Device(cell_2p5, p); gb = dev.make_gpubuf("cuda:0")
            dev.setLaplacePotential(gb, p, vd); gb.sync_HostToGPU(dev)
            dev.updateCharge(gb); dev.updatePotential(gb, p, vd, 0)
            p.solve_heating_global = False                     # (with heating on the node potentials are shifted after the solve)
            dev.updatePower(gb, p, vd)
            st = host.get_stats()
            rec = [dev.imacro, None, host.get_last_X(), st["X_nnz"], get(gb, "atom_virtual_potentials").copy()]
            p.solve_heating_global = True
            put(gb, "atom_virtual_potentials", np.zeros(dev.N_atom + 2))
            dev.updatePower(gb, p, vd)
            rec[1] = get(gb, "site_power").copy()
            out[fmt] = rec
    finally:
        L.dkmc_set_x_format(1); L.dkmc_set_x_block(16)
    (i0, pw0, (rp0, ci0, d0), nnz0, m0), (i1, pw1, (rp1, ci1, d1), nnz1, m1) = out[0], out[1]
    if case == "empty_S":
        assert host.get_stats()["xt_ns"] == 0
    assert nnz0 == nnz1 == len(ci0) == len(ci1)
    assert np.array_equal(rp0, rp1) and np.array_equal(ci0, ci1)
    assert np.abs(d1 - d0).max() <= 1e-12 * np.abs(d0).max() and np.all(np.abs(d1 - d0) <= 1e-10 * np.abs(d0))
    n = len(rp0) - 1
    # both solutions satisfy the stop test in the true scaled residual of the same matrix; the scaled X couples nodes through
    # conductances from 1e-8 to 1e7 (cond ~ 1e13), so two roundings of a 1e-10 residual may differ by more than 1e-8 on the
    # weakly coupled nodes
    for m_ in (m0, m1):
        assert _scaled_residual(rp0, ci0, d0, m_, p.G0, p.X_loop_G, vd) <= 10 * p.cg_tol
    loose = case == "small_bias"         # nearly floating nodes: differences of potentials carry the conditioning
    assert np.abs(m1[:n] - m0[:n]).max() <= (1e-5 if loose else 1e-7) * np.abs(m0[:n]).max()
    # I_macro = sum of n_src terms -high_G (m_c - m_1) that cancel (by 9 digits at 0.03 V): tolerance relative to the terms
    assert abs(i1 - i0) <= 1e-8 * abs(i0) + 1e-9 * p.X_high_G * p.num_atoms_first_layer * np.abs(m0[:n]).max()
    assert np.abs(pw1 - pw0).max() <= (1e-4 if loose else 1e-8) * max(np.abs(pw0).max(), 1e-300)


def _fresh_device(structure, p, hip, warm=None):
    """(Device, KMCProcess, GPUBuffers, None) right after setLaplacePotential, without the oracle twin (large devices)."""
    host, L = hip
    dev = host.Device(structure, p)
    sim = host.KMCProcess(dev, p.freq)
    gb = dev.make_gpubuf("cuda:0")
    if warm is not None:
        L.dkmc_set_current_warm_start(warm)
    dev.setLaplacePotential(gb, p, Vd)
    gb.sync_HostToGPU(dev)
    return dev, sim, gb, None


def test_supersteps_tiled_structure(cell_2p5, hip):
    """37 596 sites (the 2.5 nm cell tiled 2 x 2 laterally): a tunnelling block of a few hundred tiles, small enough for the CPU
    oracle.  Three full supersteps with global heating: the tile layout is rebuilt every step as the vacancies
    move; events identical to the oracle's, KMC time / current / temperature within the tolerances of the other superstep tests
    (both solves converged to a scaled residual of 1e-12)."""
    from devicekmc_amd import params as pm
    from devicekmc_amd import structure
    host, L = hip
    s = structure.tile_structure(cell_2p5, 2, 25.575, 25.575, 1440)
    # converged solves (1e-12): at a looser tolerance the comparison would measure whether two implementations stop on the same
    # iterate (cond(K) ~ 1e8: at 1e-9 the KMC time of neighbouring iterates still differs by 1e-3), not whether they solve the same system
    p = pm.KMCParameters().for_tiling(2); p.solve_heating_global = True; p.cg_tol = 1e-12
    dev, sim, gb, o = make_pair(s, p, hip)
    for k in range(3):
        dev.updateCharge(gb); dev.updatePotential(gb, p, Vd, k)
        _, dt = sim.executeKMCStep(gb, dev, want_log=True)
        dev.updatePower(gb, p, Vd); dev.updateTemperature(gb, p, dt)
        st = host.get_stats()
        assert st["spmv_tiles"] > 0 and 2 * st["spmv_tile_entries"] >= 0.8 * st["X_nnz"]      # the tiles carry the solve
        out = o.superstep(Vd)
        assert np.array_equal(sim.last_event_log, o.last_events["log"]), k
        # K mixes conductances of 1 and 1e-8 (cond ~1e8): potentials, hence rates and the KMC time, agree to cond(K) x residual
        assert abs(dt / out["step_time"] - 1) <= 1e-5
        assert abs(dev.imacro / out["imacro"] - 1) <= 1e-5
        assert abs(dev.T_bg - out["T_bg"]) <= 1e-7
    rp, ci, data = host.get_last_X()
    assert np.array_equal(rp, o.last_X["row_ptr"]) and np.array_equal(ci, o.last_X["col"])


def test_crossbar_with_current_solve(hip, golden_dir):
    """configs[4] geometry (crossbar, 110 813 sites) with the current solve switched ON (every shipped crossbar parameter set has
    solve_current = 0, so the reference holds no number for it), V = 1, two supersteps.  The oracle follows the potential / event
    path (same events); the current solve is checked the way the ~1e6-site test does it: the solved node potentials satisfy
    X m = b on sampled rows that the oracle generates on the fly (assembling this X and its WKB integrals on the CPU takes minutes)."""
    from devicekmc_amd import params as pm, structure
    from oracle import oracle as oc
    host, L = hip
    s = structure.load_structure(os.path.join(golden_dir, "crossbar_10nm_5pitch.npz"))
    p = pm.KMCParameters(rnd_seed=5, lattice=tuple(s.meta["lattice"]), num_atoms_first_layer=144, num_atoms_contact=11520)
    p.cg_tol = 1e-12
    V = 1.0
    dev = host.Device(s, p, gpu_neighbors="cuda:0"); sim = host.KMCProcess(dev, p.freq)
    gb = dev.make_gpubuf("cuda:0")
    dev.setLaplacePotential(gb, p, V); gb.sync_HostToGPU(dev)
    po = pm.KMCParameters(rnd_seed=5, lattice=tuple(s.meta["lattice"]), num_atoms_first_layer=144, num_atoms_contact=11520)
    po.cg_tol = 1e-12; po.solve_current = False
    o = oc.OracleKMC(s.element, s.x, s.y, s.z, po, neigh=dev.neigh_idx)
    o.set_laplace_potential(V)
    # the CB-edge system couples the oxide through conductances of 1e-8 (cond ~ 1e8 and more in this geometry): two converged
    # solves agree to ~1e-6 there; the row check below uses the CB edge the GPU solve used
    assert np.abs(get(gb, "site_CB_edge") - o.CB_edge).max() <= 1e-5 * np.abs(o.CB_edge).max()
    o.CB_edge[:] = get(gb, "site_CB_edge")
    rng = np.random.default_rng(11)
    for k in range(2):
        dev.updateCharge(gb); dev.updatePotential(gb, p, V, k)
        _, dt = sim.executeKMCStep(gb, dev, want_log=True)
        dev.updatePower(gb, p, V)
        out = o.superstep(V)
        st = host.get_stats()
        assert np.array_equal(sim.last_event_log, o.last_events["log"]), k
        assert abs(dt / out["step_time"] - 1) <= 1e-5
        assert st["spmv_tiles"] > 0 and st["cg_rr_X"] <= p.cg_tol ** 2 and dev.imacro > 0
        gb.sync_GPUToHost(dev)
        assert np.array_equal(dev.site_element, o.element) and np.array_equal(dev.site_charge, o.charge)
        m = get(gb, "atom_virtual_potentials")                 # G0-scaled, not shifted (no heating)
        el = o.element
        atom_site = np.flatnonzero((el != 0) & (el != 1))
        Na = len(atom_site); a = np.arange(Na); ael = el[atom_site]
        n1, nlc = p.num_atoms_first_layer, p.num_layers_contact
        inner = np.isin(ael, list(p.metals)) & (a > (nlc - 1) * n1) & (a < Na - (nlc - 1) * n1) & (a < Na - 1)
        vac = (ael == 2) & (a < Na - 1)
        plain = ~inner & ~vac & (a < Na - 1)
        rows = np.concatenate([rng.choice(np.flatnonzero(vac), 8, replace=False), rng.choice(np.flatnonzero(inner), 8, replace=False),
                               rng.choice(np.flatnonzero(plain), 16, replace=False)]).astype(np.int32) + 2
        diag, xm = o.x_rows_apply(rows, m)
        assert (np.abs(xm) / p.G0 / np.sqrt(diag)).max() <= 1e-9          # scaled residual of the sampled rows (b = 0 on atom rows)


def test_restart_continues_the_same_event_sequence(cell_2p5, hip, tmp_path):
    """f2: three supersteps, snapshot + sidecar (host.save_restart), fresh objects from the files (host.load_restart), three more
    supersteps: same (slot, i, j, type) event log, KMC time, I_macro and T_bg -- bit for bit -- as the uninterrupted run."""
    from devicekmc_amd import params as pm
    host, L = hip
    p = pm.KMCParameters(); p.solve_heating_global = True

    def steps(dev, sim, gb, k0, n):
        out = []
        for k in range(k0, k0 + n):
            dev.updateCharge(gb); dev.updatePotential(gb, p, Vd, k)
            _, dt = sim.executeKMCStep(gb, dev, want_log=True)
            dev.updatePower(gb, p, Vd); dev.updateTemperature(gb, p, dt)
            out.append((sim.last_event_log.copy(), dt, dev.imacro, dev.T_bg))
        return out

    dev = host.Device(cell_2p5, p); sim = host.KMCProcess(dev, p.freq); gb = dev.make_gpubuf("cuda:0")
    dev.setLaplacePotential(gb, p, Vd); gb.sync_HostToGPU(dev)
    full = steps(dev, sim, gb, 0, 6)
    del gb, sim, dev
    dev = host.Device(cell_2p5, p); sim = host.KMCProcess(dev, p.freq); gb = dev.make_gpubuf("cuda:0")
    dev.setLaplacePotential(gb, p, Vd); gb.sync_HostToGPU(dev)
    first = steps(dev, sim, gb, 0, 3)
    path = str(tmp_path / "snapshot_3.xyz")
    host.save_restart(path, dev, sim, gb, kmc_time=sum(f[1] for f in first), kmc_step_count=3)
    del gb, sim, dev
    dev2, sim2, gb2, state = host.load_restart(path, p, "cuda:0")
    assert state["kmc_step_count"] == 3 and state["kmc_rng_raw_draws"] > 0
    rest = steps(dev2, sim2, gb2, 3, 3)
    for a, b in zip(full[:3], first):
        assert np.array_equal(a[0], b[0]) and a[1:] == b[1:]
    for a, b in zip(full[3:], rest):
        assert np.array_equal(a[0], b[0])
        assert a[1:] == b[1:], (a[1:], b[1:])


def test_two_devices_in_one_process_keep_their_own_solver_state(cell_2p5, hip):
    """Two GPUBuffers of the same size alive in one process, stepped alternately (the "one device per crossbar cell" use): the
    coefficient cache and the private warm-start copy are kept per GPUBuffers, so each trajectory equals the one of a process
    that runs that device alone -- bit for bit, also with dkmc_set_current_warm_start(1)."""
    from devicekmc_amd import params as pm
    host, L = hip
    pa = pm.KMCParameters(); pa.solve_heating_global = True
    pb = pm.KMCParameters(); pb.solve_heating_global = True; pb.rnd_seed = 7; pb.rnd_seed_kmc = 3
    VA, VB = 5.0, 3.0

    def make(p, V):
        dev = host.Device(cell_2p5, p); sim = host.KMCProcess(dev, p.freq); gb = dev.make_gpubuf("cuda:0")
        dev.setLaplacePotential(gb, p, V); gb.sync_HostToGPU(dev)
        return dev, sim, gb

    def step(dev, sim, gb, p, V, k):
        dev.updateCharge(gb); dev.updatePotential(gb, p, V, k)
        _, dt = sim.executeKMCStep(gb, dev)
        dev.updatePower(gb, p, V); dev.updateTemperature(gb, p, dt)
        return (dt, dev.imacro, dev.T_bg)

    try:
        L.dkmc_set_current_warm_start(1)
        alone = {}
        for name, p, V in (("a", pa, VA), ("b", pb, VB)):
            d = make(p, V)
            alone[name] = [step(*d, p, V, k) for k in range(3)]
            del d
        da, db = make(pa, VA), make(pb, VB)
        both = {"a": [], "b": []}
        for k in range(3):
            both["a"].append(step(*da, pa, VA, k))
            both["b"].append(step(*db, pb, VB, k))
        assert both["a"] == alone["a"] and both["b"] == alone["b"]
    finally:
        L.dkmc_set_current_warm_start(1)


@pytest.mark.parametrize("offset", [0, 2])
def test_split_matrix_cg(hip, offset):
    """solve_sparse_CG_splitmatrix (iterative_solvers_gpu.cu:656-821, unfinished in the reference): unpreconditioned CG on
    (A + P^T M P) y = x with A in CSR and M dense on an index subset, against a direct solve of the assembled system.  Ragged CSR
    rows, an odd dense-block size (unaligned rows of M), a warm start; offset 2 = the reference's node = atom + 2 convention."""
    import ctypes as C
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    import torch
    host, L = hip
    rng = np.random.default_rng(5 + offset)
    m, msub = 4000, 777
    B = sp.random(m, m, density=3e-3, random_state=3, data_rvs=rng.standard_normal).tocsr()
    A = (B + B.T) * 0.1
    A = (A + sp.diags(np.abs(A).sum(axis=1).A1 + 1.0)).tocsr()           # symmetric, diagonally dominant
    A.sort_indices()
    G = rng.standard_normal((msub, 40))
    M = G @ G.T / 40.0 + 0.05 * rng.standard_normal((msub, msub)); M = 0.5 * (M + M.T) + 2.0 * np.eye(msub)      # symmetric positive definite
    idx = np.sort(rng.choice(m - offset, msub, replace=False)).astype(np.int32)
    P = sp.csr_matrix((np.ones(msub), (np.arange(msub), idx + offset)), shape=(msub, m))
    full = (A + P.T @ sp.csr_matrix(M) @ P).tocsc()
    x = rng.standard_normal(m)
    want = spl.spsolve(full, x)
    dev = torch.device("cuda:0")
    tt = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a, dtype=dt)).to(dev)
    dA, drp, dci, dM, didx, dx = tt(A.data, np.float64), tt(A.indptr, np.int32), tt(A.indices, np.int32), tt(M, np.float64), tt(idx, np.int32), tt(x, np.float64)
    dy = tt(0.5 * want + 0.1 * rng.standard_normal(m), np.float64)      # some start vector
    it, rn = C.c_int(0), C.c_double(0.0)
    ptr = lambda t: C.c_void_p(t.data_ptr())
    tol = 1e-9
    from devicekmc_amd.lib import check
    check(L.dkmc_set_stream(C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    check(L.dkmc_solve_sparse_CG_splitmatrix(ptr(dM), msub, ptr(dA), ptr(drp), ptr(dci), A.nnz, m, ptr(didx), offset, ptr(dx), ptr(dy), tol,
                                              C.byref(it), C.byref(rn)))
    torch.cuda.synchronize()
    got = dy.cpu().numpy()
    assert 0 < it.value < 2000 and rn.value <= tol
    assert np.linalg.norm(full @ got - x) <= 10 * tol                   # the true residual meets the stop test
    assert np.abs(got - want).max() <= 1e-8 * np.abs(want).max()
    # inputs are read only
    assert np.array_equal(dM.cpu().numpy(), M) and np.array_equal(dA.cpu().numpy(), A.data) and np.array_equal(dx.cpu().numpy(), x)


def test_pair_sum_site_grouping_follows_the_structure(cell_2p5, hip):
    """The cached grouping of the sites by (y, z) column (cell-list pair sum) must not outlive the structure it was built for:
    a GPUBuffers is freed and one of the SAME size with the sites in another order is created -- hipMalloc returns the old addresses -- and
    the pair sum of the second structure equals the oracle's; then positions are rewritten in place behind the same pointers and
    dkmc_reset_pair_sum_cache() is called: again the oracle's sums.  (Before the fix the first case summed sites over the columns of the
    old structure.)"""
    from devicekmc_amd import params as pm, structure
    from devicekmc_amd.host import _ptr
    from devicekmc_amd.lib import check
    from oracle import oracle as oc
    host, L = hip
    k = 6
    s = structure.tile_structure(cell_2p5, k, 25.575, 25.575, 1440)
    p = pm.KMCParameters().for_tiling(k)
    lat = np.asarray(p.lattice, dtype=np.float64)
    rng = np.random.default_rng(23)

    def oracle_sum(x, y, z, q):
        want = np.zeros(len(x))
        oc.lib().okmc_poisson_gridless(len(x), oc._p(x), oc._p(y), oc._p(z), oc._p(lat), 0, C.c_double(p.sigma), C.c_double(p.k), oc._p(q), oc._p(want))
        return want

    def gpu_sum(gb):
        check(L.dkmc_poisson_gridless_gpu(0, 0, gb.N_, _ptr(gb.lattice), _ptr(gb.sigma), _ptr(gb.k), _ptr(gb.site_x), _ptr(gb.site_y),
                                          _ptr(gb.site_z), _ptr(gb.site_charge), _ptr(gb.site_potential_charge)))
        return get(gb, "site_potential_charge").copy()

    dev = host.Device(s, p, gpu_neighbors="cuda:0")
    ok = (dev.site_element == pm.O_EL) | (dev.site_element == pm.VACANCY)
    q = np.where(ok & (rng.random(dev.N) < 0.02), rng.choice([-2, 2], dev.N), 0).astype(np.int32)
    gb = dev.make_gpubuf("cuda:0")
    put(gb, "site_charge", q)
    want = oracle_sum(dev.site_x, dev.site_y, dev.site_z, q)
    assert np.abs(gpu_sum(gb) - want).max() <= 1e-12 * np.abs(want).max()
    ptrs = (_ptr(gb.site_x), _ptr(gb.site_y), _ptr(gb.site_z))
    # ---- same size, the lateral coordinates mirrored: every site changes its column ----
    y2 = np.ascontiguousarray(lat[1] - dev.site_y - 1e-3); z2 = np.ascontiguousarray(lat[2] - dev.site_z - 1e-3)
    want2 = oracle_sum(dev.site_x, y2, z2, q)
    del gb                      # (GPUBuffers.__del__ -> dkmc_free_sparsity; torch's caching allocator hands the next buffers the same addresses)
    import gc
    gc.collect()
    dev2 = host.Device(s, p, gpu_neighbors="cuda:0")
    dev2.site_y[:] = y2; dev2.site_z[:] = z2
    gb2 = dev2.make_gpubuf("cuda:0")
    put(gb2, "site_charge", q)
    got2 = gpu_sum(gb2)
    assert np.abs(got2 - want2).max() <= 1e-12 * np.abs(want2).max(), "stale grouping after free + create (same addresses: %s)" % (ptrs == (_ptr(gb2.site_x), _ptr(gb2.site_y), _ptr(gb2.site_z)))
    # ---- in place, behind the same pointers ----
    put(gb2, "site_y", dev.site_y); put(gb2, "site_z", dev.site_z)
    L.dkmc_reset_pair_sum_cache()
    assert np.abs(gpu_sum(gb2) - want).max() <= 1e-12 * np.abs(want).max()


@pytest.mark.parametrize("m", [1000, 85071, 700001])
def test_step_kernel_stop_word_is_stamped_with_the_iteration(hip, m):
    """The CG step kernel both reads and (workgroup 0) writes the stop word.  With a plain 0 / 1 flag a workgroup scheduled after workgroup 0
    had published convergence skipped y += alpha p on the converging iteration: a partially updated, run-dependent solution (round 2).  The
    word is stamped instead: d > 0 stops the kernels of iterations >= d - 1, and the value the converging iteration publishes is it + 2.
    Deterministically (dkmc_debug_step_stop_word: one launch of iteration `it` with the word preset): 0 and it + 2 update EVERY element --
    whichever workgroups see the word; it + 1 and anything below stop the launch as a whole (an earlier iteration converged)."""
    host, L = hip
    from devicekmc_amd.lib import check
    for it in (0, 7):
        for word, expect in ((0, m), (it + 2, m), (it + 3, m), (it + 1, 0), (1, 0)):
            n, after = C.c_int(-1), C.c_int(-1)
            check(L.dkmc_debug_step_stop_word(m, it, word, C.byref(n), C.byref(after)))
            assert n.value == expect, (m, it, word, n.value)
            assert after.value == word                       # r'.r' of the harness stays above the tolerance: the launch publishes nothing


@pytest.mark.parametrize("which", ["2.5nm", "7.5nm"])
def test_K_blocked_form_matches_csr_positions(cell_2p5, dev_7p5, hip, which):
    """The CG on K in its internal blocked order (rows sorted by x, one block per CU, q window in LDS: csrc/kcg.hip, the default up to
    262 144 rows) against the same solve on the CSR positions of the pattern (dkmc_set_k_blocked(0)): background potential and CB edge in
    SITE order agree to 1e-8 of the bias at a converged tolerance (1e-12; cond(K) x residual: 7e-9 V measured at 85 k sites, where either
    solve is within 1e-7 of the oracle's), contacts re-imposed exactly, and the unscaled residual of the
    oracle's K is met by both (2.5 nm).  Nothing outside the solve sees the internal order."""
    from devicekmc_amd import params as pm
    host, L = hip
    structure, p = (cell_2p5, pm.KMCParameters()) if which == "2.5nm" else (dev_7p5, params_7p5())
    p.cg_tol = 1e-12
    res = {}
    try:
        for on in (0, 1):
            L.dkmc_set_k_blocked(on)
            dev, sim, gb, _ = _fresh_device(structure, p, hip)          # (initialize_sparsity reads the switch)
            cb = dev.site_CB_edge.copy()
            used_cb = host.get_stats()["kcg_blocked"]
            dev.updateCharge(gb); dev.updatePotential(gb, p, Vd, 0)
            st = host.get_stats()
            assert st["kcg_blocked"] == on and used_cb == on
            res[on] = (get(gb, "site_potential_boundary").copy(), cb, st["cg_iters_K"])
        n = p.num_atoms_first_layer
        assert np.abs(res[1][0] - res[0][0]).max() <= 1e-8 * Vd, (res[0][2], res[1][2])
        assert np.abs(res[1][1] - res[0][1]).max() <= 1e-8 * p.q * Vd
        assert (res[1][0][:n] == -Vd / 2).all() and (res[1][0][-n:] == Vd / 2).all()
        assert abs(res[1][2] - res[0][2]) <= 0.05 * res[0][2] + 2           # same Krylov sequence up to rounding
        if which == "2.5nm":
            import scipy.sparse as sp
            from oracle import oracle as oc
            o = oc.OracleKMC(structure.element, structure.x, structure.y, structure.z, p)
            o.set_laplace_potential(Vd); o.update_charge(); o.update_potential(Vd)
            rp, ci, data, rhs = o._last_K
            K = sp.csr_matrix((data, ci, rp))
            for on in (0, 1):
                assert np.abs(K @ res[on][0][n:dev.N - n] - rhs).max() <= 1e-9, on
    finally:
        L.dkmc_set_k_blocked(1); L.dkmc_set_cg_tolerance(1e-6)


def test_soak_60_supersteps_against_the_oracle(cell_2p5, hip):
    """60 coupled supersteps of the 2.5 nm device with global heating on, HIP path against the CPU oracle step by step (the long-trajectory
    check that used to live in tools/soak_vs_oracle.py): the same executed events (slot, i, j, type), charges and elements bit for bit,
    current / KMC time / temperature within the CG tolerance (1e-10), at the library's defaults otherwise (block-CG on X)."""
    from devicekmc_amd import params as pm
    host, L = hip
    p = pm.KMCParameters(); p.cg_tol = 1e-10; p.solve_heating_global = True
    dev, sim, gb, o = make_pair(cell_2p5, p, hip)
    worst = dict(I=0.0, T=0.0, dt=0.0, margin=1.0)
    nev = 0
    for k in range(60):
        dev.updateCharge(gb); dev.updatePotential(gb, p, Vd, k)
        _, dt = sim.executeKMCStep(gb, dev, want_log=True)
        dev.updatePower(gb, p, Vd); dev.updateTemperature(gb, p, dt)
        out = o.superstep(Vd)
        assert np.array_equal(sim.last_event_log, o.last_events["log"]), k
        assert np.array_equal(get(gb, "site_element"), o.element) and np.array_equal(get(gb, "site_charge"), o.charge), k
        worst["I"] = max(worst["I"], abs(dev.imacro / out["imacro"] - 1)); worst["T"] = max(worst["T"], abs(dev.T_bg - out["T_bg"]))
        worst["dt"] = max(worst["dt"], abs(dt / out["step_time"] - 1)); worst["margin"] = min(worst["margin"], float(o.last_events["margin"].min()))
        nev += len(sim.last_event_log)
        assert worst["I"] <= 1e-6 and worst["dt"] <= 1e-6 and worst["T"] <= 1e-7, (k, worst)
    assert nev > 60 and worst["margin"] > 1e-9, (nev, worst)
