"""Pins the CPU oracle (oracle/kmc_oracle.c) against everything the reference ships for this path:
its CPU-path numbers, its CUDA-path run logs and its X-sparsity dump."""
import os

import numpy as np
import pytest

from conftest import params_7p5
from devicekmc_amd import params as pm
from oracle import oracle as oc


def test_rng_matches_std_mt19937():
    # std::mt19937 default seed 5489: first two outputs 3499211612, 581869302 (well-known test vector);
    # libstdc++ uniform_real_distribution<double> = (x1 + x2 * 2^32) / 2^64
    r = oc.OracleRNG(5489)
    u = r.uniform()
    assert u == (3499211612 + 581869302 * 4294967296.0) / 18446744073709551616.0
    # 10000th 32-bit output of mt19937(5489) is 4123659995 (C++ standard, [rand.predef]); 2 draws per uniform
    r = oc.OracleRNG(5489)
    for _ in range(4999):
        r.uniform()
    u = r.uniform()          # consumes outputs 9999 and 10000
    x2 = int(u * 18446744073709551616.0) >> 32
    assert abs(x2 - 4123659995) <= 1


def test_cpu_path_numbers_2p5nm(cell_2p5, ref_logs):
    """Reference CPU path (dense LU), test_2.5nm, V=5: Current [uA], KMC time, charged vacancies."""
    gold = ref_logs["BASELINE.md#2 (reference CPU path run during the survey)"]["steps"]
    p = pm.KMCParameters(); p.cg_tol = 1e-10           # LU-equivalent accuracy
    for sem in ("cpu", "cuda"):
        o = oc.OracleKMC(cell_2p5.element, cell_2p5.x, cell_2p5.y, cell_2p5.z, p, semantics=sem)
        o.set_laplace_potential(5.0)
        t = 0.0
        for k in range(2):
            out = o.superstep(5.0)
            t += out["step_time"]
            assert float("%.6g" % (out["imacro"] * 1e6)) == gold[k]["Current [uA]"]
            assert float("%.6g" % t) == gold[k]["KMC time"]
            if k == 0:
                pass
        # 88 charged / 12 uncharged vacancies were logged for step 0; charges are re-evaluated every step
    o = oc.OracleKMC(cell_2p5.element, cell_2p5.x, cell_2p5.y, cell_2p5.z, p)
    o.update_charge()
    vac = o.element == pm.VACANCY
    assert int((vac & (o.charge != 0)).sum()) == gold[0]["Charged vacancies"]
    assert int((vac & (o.charge == 0)).sum()) == gold[0]["Uncharged vacancies"]


def _pattern_diff(X, g):
    ael = X["ael"]
    ndiff = 0
    for r in range(len(g["row_ptr"]) - 1):
        a = g["col_idx"][g["row_ptr"][r]:g["row_ptr"][r + 1]]
        b = X["col"][X["row_ptr"][r]:X["row_ptr"][r + 1]]
        if len(a) == len(b) and np.array_equal(a, b):
            continue
        d = np.setxor1d(a, b)
        assert r >= 2 and ael[r - 2] == pm.VACANCY and (ael[d - 2] == pm.VACANCY).all(), r
        ndiff += len(d)
    return ndiff


def test_x_pattern_vs_reference_dump(cell_2p5, golden_dir):
    """X sparsity after step 0 vs the CSR the reference CUDA path dumped (timing_2.5nm/fullmatrix_assembly, 467 336 entries).
    With the CB edge solved on ATOMS (`log_revision()`: interstitial sites carry no link in the CB-edge system) the dump is
    reproduced entry for entry: row pointers and column indices are IDENTICAL.  With the snapshot's source (every site in the system,
    potential_solver_gpu.cu:595-694) 60 entries differ, all vacancy-vacancy pairs: the interstitial links shift the CB edge of
    single vacancies by up to 0.1 eV, across the 0.01 eV threshold of the tunnelling rule (iterative_solvers_gpu.cu:903-908)."""
    g = np.load(os.path.join(golden_dir, "x_pattern_2.5nm_step0.npz"))
    for p, want in ((pm.KMCParameters().log_revision(), 0), (pm.KMCParameters(), 60)):
        o = oc.OracleKMC(cell_2p5.element, cell_2p5.x, cell_2p5.y, cell_2p5.z, p)
        o.set_laplace_potential(5.0)
        o.update_charge(); o.update_potential(5.0); o.execute_kmc_step()
        X = o.assemble_X()
        assert len(X["row_ptr"]) == len(g["row_ptr"])
        if want == 0:
            assert np.array_equal(X["row_ptr"], g["row_ptr"]) and np.array_equal(X["col"], g["col_idx"])
            assert len(X["col"]) == 467336
        assert _pattern_diff(X, g) == want
        # rows 0 and 1: 144 = {0,1} + 142 extraction columns; 146 = num_source_inj + 2
        assert X["row_ptr"][1] == 144 and X["row_ptr"][2] - X["row_ptr"][1] == 146


def test_cuda_path_log_7p5nm(dev_7p5, ref_logs):
    """KMC time of the reference's CUDA-path log (85 071 sites), potential + event path.  With the CG tolerance the reference
    used before it was relaxed to 1e-6 ("used to be 1e-12", iterative_solvers_gpu.cu:322) the oracle reproduces every printed
    digit of the log; at 1e-9 it agrees to 1e-4, at 1e-6 to 4.5 % (tools/pin_current.py): the log was produced at 1e-12."""
    gold = ref_logs["timing_7.5nm/output_noguess.txt"]["steps"]
    p = params_7p5(); p.cg_tol = 1e-12; p.solve_current = False
    o = oc.OracleKMC(dev_7p5.element, dev_7p5.x, dev_7p5.y, dev_7p5.z, p)
    assert o.N == 85071 and o.nn == 52 and int((o.element == pm.VACANCY).sum()) == 900
    t = 0.0
    for k in range(4):
        t += o.superstep(5.0)["step_time"]
        assert abs(t / gold[k]["KMC time"] - 1) < 5e-6, (k, t, gold[k])      # 6 printed digits
        assert o.last_events["margin"].min() > 1e-9      # no draw within rounding distance of a bucket edge


def test_current_7p5nm_vs_log(dev_7p5, ref_logs):
    """Current [uA] of the reference's CUDA-path log (timing_7.5nm/output_noguess.txt, 19 supersteps) under `log_revision()` (CG
    tolerance 1e-12, CB edge solved on atoms): step 0 here -- 11.88338 against the log's 11.8834, all six printed digits, with the
    KMC time of the same step; all 19 steps by tools/pin_current_constants.py --steps 19 (its output: profiles/r03_pin_current.txt)
    and by tests/test_gpu_parity.py::test_reference_log_7p5_currents on the HIP path.  The same state with the snapshot's CB-edge
    domain (every site) gives 11.78485 uA, 0.83 % lower at every step: that is the difference between the snapshot and the revision
    that wrote the log, not an error of either restatement."""
    gold = ref_logs["timing_7.5nm/output_noguess.txt"]["steps"]
    assert len(gold) == 19 and all("Current [uA]" in g for g in gold)
    p = params_7p5().log_revision()
    o = oc.OracleKMC(dev_7p5.element, dev_7p5.x, dev_7p5.y, dev_7p5.z, p)
    o.set_laplace_potential(5.0)
    out = o.superstep(5.0)
    assert o.stats["X_nnz"] == 30605002
    assert abs(out["imacro"] * 1e6 - gold[0]["Current [uA]"]) <= 0.6e-4          # half a unit of the 6th printed digit
    assert abs(out["step_time"] / gold[0]["KMC time"] - 1) < 5e-6
    # snapshot domain on the same event state: the documented 0.83 %
    p2 = params_7p5(); p2.cg_tol = 1e-9
    o2 = oc.OracleKMC(dev_7p5.element, dev_7p5.x, dev_7p5.y, dev_7p5.z, p2)
    o2.set_laplace_potential(5.0)
    o2.element[:] = o.element; o2.charge[:] = o.charge
    im2 = o2.update_power(5.0, heating=False) * 1e6
    assert abs(im2 / 11.78485 - 1) < 2e-6
    assert abs(im2 / (out["imacro"] * 1e6) - 1 + 8.29e-3) < 2e-4


def test_cuda_path_log_crossbar(ref_logs, golden_dir):
    """Second geometry: crossbars/timing_10nm_5pitch (110 813 sites, V = 1, solve_current = 0).  At the tolerance the reference's
    logs were made with (1e-12, see test_cuda_path_log_7p5nm) ALL 13 logged supersteps agree to the 6 printed digits; at 1e-9 the
    first 11 agree to 5e-3 and the trajectories then part at one event."""
    from devicekmc_amd import structure
    gold = ref_logs["crossbars/timing_10nm_5pitch/output_initial.txt"]["steps"]
    s = structure.load_structure(os.path.join(golden_dir, "crossbar_10nm_5pitch.npz"))
    p = pm.KMCParameters(rnd_seed=5, lattice=tuple(s.meta["lattice"]), num_atoms_first_layer=144, num_atoms_contact=11520)
    p.cg_tol = 1e-12; p.solve_current = False
    o = oc.OracleKMC(s.element, s.x, s.y, s.z, p)
    assert o.N == 110813 and int((o.element == pm.VACANCY).sum()) == 1600
    t = 0.0
    for k in range(len(gold)):
        t += o.superstep(1.0)["step_time"]
        assert abs(t / gold[k]["KMC time"] - 1) < 5e-6, (k, t, gold[k])


def test_x_rows_on_the_fly_match_assembled_X(cell_2p5):
    """okmc_x_rows_apply (rows of X generated on the fly; the full-size check of tests/test_gpu_scale.py) against the assembled CSR
    of the same state: diagonal and X m of sampled rows, vacancy / inner-contact (tunnelling) rows and plain rows alike."""
    import scipy.sparse as sp
    from devicekmc_amd import params
    from oracle import oracle as oc
    p = params.KMCParameters()
    o = oc.OracleKMC(cell_2p5.element, cell_2p5.x, cell_2p5.y, cell_2p5.z, p)
    o.set_laplace_potential(5.0); o.update_charge(); o.update_potential(5.0)
    X = o.assemble_X()
    n = X["Na"] + 1
    A = sp.csr_matrix((X["data"], X["col"], X["row_ptr"]), shape=(n, n))
    rng = np.random.default_rng(3)
    m = rng.standard_normal(X["Na"] + 2)
    lens = np.diff(X["row_ptr"])
    long_rows = np.flatnonzero(lens > 200); long_rows = long_rows[long_rows >= 2]
    rows = np.concatenate([rng.choice(long_rows, 24, replace=False), rng.choice(np.arange(2, n), 40, replace=False)]).astype(np.int32)
    diag, axm = o.x_rows_apply(rows, m)
    want = (A @ m[:n])[rows]
    assert np.allclose(diag, A.diagonal()[rows], rtol=1e-12, atol=0)
    assert np.all(np.abs(axm - want) <= 1e-10 * np.abs(A[rows]).dot(np.abs(m[:n])))
