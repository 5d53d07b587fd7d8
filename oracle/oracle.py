"""Python driver of the CPU oracle (oracle/kmc_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg as the checker / reported baseline, never by the product path.

``OracleKMC`` holds the same state the reference's Device + GPUBuffers hold and exposes one method
per reference entry point (same names minus the ``_gpu`` suffix), each restating the CUDA path's
semantics.  ``semantics="cpu"`` flips the handful of documented CPU-vs-CUDA
differences and is used only to pin the oracle against CPU-path numbers.
"""
import ctypes as C
import os
import subprocess
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_int_p = C.POINTER(C.c_int)
c_dbl_p = C.POINTER(C.c_double)


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libkmc_oracle.so")
        src = os.path.join(_HERE, "kmc_oracle.c")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            build()
        L = C.CDLL(path)
        L.okmc_site_dist.restype = C.c_double
        L.okmc_v_solve.restype = C.c_double
        L.okmc_v_solve.argtypes = [C.c_double, C.c_int, C.c_double, C.c_double]
        L.okmc_rng_uniform.restype = C.c_double
        L.okmc_x_pattern.restype = C.c_longlong
        L.okmc_imacro_row1.restype = C.c_double
        L.okmc_imacro_row0.restype = C.c_double
        L.okmc_imacro_row0.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double]
        L.okmc_temperature_global.restype = C.c_double
        L.okmc_cg_iter_bench.restype = C.c_double
        L.okmc_cg_iter_bench.argtypes = [C.c_int, C.c_int, C.c_longlong, C.c_longlong, C.c_int]
        L.okmc_temperature_global.argtypes = [C.c_int, C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_double,
                                              C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_void_p]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class OracleRNG:
    """std::mt19937 + uniform_real_distribution<double>(0,1) (random_num.h:4-23)."""

    def __init__(self, seed):
        self.buf = C.create_string_buffer(lib().okmc_rng_sizeof())
        lib().okmc_rng_seed(self.buf, C.c_uint32(seed))

    def uniform(self):
        return lib().okmc_rng_uniform(self.buf)


def build_neighbors(x, y, z, lattice, pbc, nn_dist):
    L = lib()
    N = len(x)
    lat = _f64(lattice)
    nn = L.okmc_build_neighbors(N, _p(x), _p(y), _p(z), _p(lat), int(pbc), C.c_double(nn_dist), 0, None)
    out = np.empty((N, nn), dtype=np.int32)
    L.okmc_build_neighbors(N, _p(x), _p(y), _p(z), _p(lat), int(pbc), C.c_double(nn_dist), nn, _p(out))
    return out, nn


def cg_jacobi(row_ptr, col, data, rhs, guess, tol=1e-6, max_iter=0):
    """solve_sparse_CG_Jacobi on copies; returns (solution, iterations, final ||r||^2)."""
    a = _f64(data).copy(); x = _f64(rhs).copy(); y = _f64(guess).copy()
    rr = C.c_double(0)
    it = lib().okmc_cg_jacobi(len(x), _p(row_ptr), _p(col), _p(a), _p(x), _p(y), C.c_double(tol), int(max_iter), C.byref(rr))
    return y, it, rr.value


def cg_iter_bench(m, n_long, nnz_long, nnz_short, niter=3):
    """Seconds per iteration of the oracle's CG loop body on a CSR of X's shape (see okmc_cg_iter_bench)."""
    return lib().okmc_cg_iter_bench(int(m), int(n_long), int(nnz_long), int(nnz_short), int(niter))


class OracleKMC:
    def __init__(self, element, x, y, z, p, semantics="cuda", neigh=None):
        """element/x/y/z: site arrays (pre-substoichiometry); p: KMCParameters."""
        L = lib()
        self.p = p
        self.sem = semantics
        self.N = len(element)
        self.element = _i32(element).copy()
        self.x, self.y, self.z = _f64(x).copy(), _f64(y).copy(), _f64(z).copy()
        self.lattice = _f64(p.lattice)
        self.metals = _i32(p.metals)
        if neigh is None:
            neigh, nn = build_neighbors(self.x, self.y, self.z, self.lattice, p.pbc, p.nn_dist)
        self.neigh = _i32(neigh)
        self.nn = self.neigh.shape[1]
        nl = len(p.layers)
        self.layer = np.empty(self.N, dtype=np.int32)
        sx = _f64([l.start_x for l in p.layers]); ex = _f64([l.end_x for l in p.layers])
        bad = L.okmc_site_layers(self.N, _p(self.x), nl, _p(sx), _p(ex), _p(self.layer))
        if bad:
            raise ValueError("Site #%d is not inside the device!" % (bad - 1))
        self.E_gen = _f64([l.E_gen_0 for l in p.layers]); self.E_rec = _f64([l.E_rec_1 for l in p.layers])
        self.E_Vdiff = _f64([l.E_diff_2 for l in p.layers]); self.E_Odiff = _f64([l.E_diff_3 for l in p.layers])
        if p.pristine:
            rng = OracleRNG(p.rnd_seed)
            L.okmc_make_substoichiometric(self.N, _p(self.element), C.c_double(p.initial_vacancy_concentration), rng.buf)
        self.rng_kmc = OracleRNG(p.rnd_seed_kmc)
        self.charge = np.zeros(self.N, dtype=np.int32)
        self.CB_edge = np.zeros(self.N)
        self.pot_boundary = np.zeros(self.N)
        self.pot_charge = np.zeros(self.N)
        self.power = np.zeros(self.N)
        self.T_bg = float(p.background_temp)
        self.Na = int(((self.element != 0) & (self.element != 1)).sum())
        self.virtual_potentials = np.zeros(self.Na + 2)
        self.imacro = 0.0
        self.timing = {}
        self.stats = {}
        self._K = None

    # --- sparsity of K, once (initialize_sparsity, iterative_solvers_gpu.cu:96-109) ---------
    def initialize_sparsity(self, n_contact=None):
        L = lib()
        nl = self.p.num_atoms_first_layer if n_contact is None else n_contact
        N = self.N
        m = N - 2 * nl
        pats = []
        for which in (0, 1, 2):
            rp = np.empty(m + 1, dtype=np.int32)
            nnz = L.okmc_k_pattern(N, self.nn, _p(self.neigh), nl, nl, which, _p(rp), None)
            ci = np.empty(max(nnz, 1), dtype=np.int32)
            L.okmc_k_pattern(N, self.nn, _p(self.neigh), nl, nl, which, _p(rp), _p(ci))
            pats.append((rp, ci[:nnz]))
        self._K = (nl, m, pats)
        return self._K

    def _solve_K(self, field, VL, VR, cb, tol):
        L = lib()
        if self._K is None:
            self.initialize_sparsity()
        nl, m, ((rp, ci), (lrp, lci), (rrp, rci)) = self._K
        data = np.zeros(len(ci)); rhs = np.zeros(m)
        L.okmc_k_assemble(self.N, nl, nl, _p(self.element), _p(self.charge), _p(self.metals), len(self.metals),
                          C.c_double(self.p.high_G), C.c_double(self.p.low_G), int(cb),
                          _p(rp), _p(ci), _p(lrp), _p(lci), _p(rrp), _p(rci), C.c_double(VL), C.c_double(VR),
                          _p(data), _p(rhs))
        self._last_K = (rp, ci, data.copy(), rhs.copy())
        guess = field[nl:nl + m].copy()
        sol, it, rr = cg_jacobi(rp, ci, data, rhs, guess, tol=tol)
        field[nl:nl + m] = sol
        return it, rr

    # --- update_CB_edge_gpu_sparse (potential_solver_gpu.cu:595-694) -------------------------
    def set_laplace_potential(self, Vd, tol=None):
        tol = self.p.cg_tol if tol is None else tol
        nl = self.p.num_atoms_first_layer
        # cb_edge_domain "atoms": the log revision's CB edge (k_conductance cb == 2), see params.KMCParameters.cb_edge_domain
        it, rr = self._solve_K(self.CB_edge, Vd / 2, -Vd / 2, 2 if self.p.cb_edge_domain == "atoms" else 1, tol)
        self.CB_edge[:nl] = Vd / 2
        self.CB_edge[self.N - nl:] = -Vd / 2
        self.CB_edge *= 1.60217663e-19
        self.stats["cg_iters_CB"] = it
        return it

    # --- update_charge_gpu (potential_solver_gpu.cu:10-63) ----------------------------------
    def update_charge(self):
        t0 = time.perf_counter()
        lib().okmc_update_charge(self.N, self.nn, _p(self.neigh), _p(self.element), _p(self.charge),
                                 _p(self.metals), len(self.metals))
        self.timing["charge"] = time.perf_counter() - t0

    # --- background_potential_gpu_sparse + poisson_gridless_gpu ------------------------------
    def update_potential(self, Vd, tol=None):
        tol = self.p.cg_tol if tol is None else tol
        t0 = time.perf_counter()
        nl = self.p.num_atoms_first_layer if self.sem == "cuda" else self.p.num_atoms_contact
        if self._K is not None and self._K[0] != nl:
            self._K = None
        if self._K is None:
            self.initialize_sparsity(nl)
        it, rr = self._solve_K(self.pot_boundary, -Vd / 2, Vd / 2, 0, tol)
        self.pot_boundary[:nl] = -Vd / 2
        self.pot_boundary[self.N - nl:] = Vd / 2
        t1 = time.perf_counter()
        lib().okmc_poisson_gridless(self.N, _p(self.x), _p(self.y), _p(self.z), _p(self.lattice), int(self.p.pbc),
                                    C.c_double(self.p.sigma), C.c_double(self.p.k), _p(self.charge), _p(self.pot_charge))
        t2 = time.perf_counter()
        self.timing["potential_boundary"] = t1 - t0
        self.timing["potential_charge"] = t2 - t1
        self.stats["cg_iters_K"] = it
        self.stats["n_charged"] = int((self.charge != 0).sum())
        return it

    # --- execute_kmc_step_gpu (kmc_events.cu:146-365) ----------------------------------------
    def build_event_list(self):
        total = self.N * self.nn
        ev_type = np.empty(total, dtype=np.int32); ev_prob = np.empty(total)
        lib().okmc_build_event_list(self.N, self.nn, _p(self.neigh), _p(self.layer), _p(self.lattice), int(self.p.pbc),
                                    C.c_double(self.T_bg), C.c_double(self.p.freq), C.c_double(self.p.sigma), C.c_double(self.p.k),
                                    _p(self.x), _p(self.y), _p(self.z), _p(self.pot_boundary), _p(self.pot_charge),
                                    _p(self.element), _p(self.charge),
                                    _p(self.E_gen), _p(self.E_rec), _p(self.E_Vdiff), _p(self.E_Odiff),
                                    1 if self.sem == "cpu" else 0, _p(ev_type), _p(ev_prob))
        return ev_type, ev_prob

    def execute_kmc_step(self, max_events=0, ev=None):
        t0 = time.perf_counter()
        ev_type, ev_prob = self.build_event_list() if ev is None else ev
        cap = max_events if max_events > 0 else 100000
        log = np.zeros((cap, 4), dtype=np.int32); margin = np.zeros(cap); psum = np.zeros(cap)
        et = C.c_double(0)
        n = lib().okmc_execute_events(self.N, self.nn, _p(self.neigh), _p(ev_type), _p(ev_prob), C.c_double(self.p.freq),
                                      _p(self.element), _p(self.charge), self.rng_kmc.buf, cap,
                                      _p(log), _p(margin), _p(psum), C.byref(et))
        self.timing["events"] = time.perf_counter() - t0
        self.last_events = dict(n=n, log=log[:n].copy(), margin=margin[:n].copy(), psum=psum[:n].copy())
        self.stats["n_events"] = n
        return et.value

    # --- update_power_gpu_sparse (current_solver_gpu.cu:854-1147) ----------------------------
    def assemble_X(self):
        L = lib(); p = self.p
        atom_site = np.empty(self.N, dtype=np.int32)
        Na = L.okmc_compact_atoms(self.N, _p(self.element), _p(atom_site))
        atom_site = atom_site[:Na].copy()
        ax, ay, az = self.x[atom_site].copy(), self.y[atom_site].copy(), self.z[atom_site].copy()
        ael = self.element[atom_site].copy(); aq = self.charge[atom_site].copy(); acb = self.CB_edge[atom_site].copy()
        an = np.empty((Na, self.nn), dtype=np.int32)
        L.okmc_atom_neighbors(self.N, self.nn, _p(self.neigh), Na, _p(atom_site), _p(an))
        n_src = n_gnd = p.num_atoms_first_layer
        rp = np.empty(Na + 2, dtype=np.int32)
        args = (Na, self.nn, _p(an), _p(ael), _p(acb), _p(self.metals), len(self.metals), C.c_double(p.X_tol),
                n_src, n_gnd, p.num_layers_contact)
        nnz = L.okmc_x_pattern(*args, _p(rp), None)
        if nnz < 0:
            raise OverflowError("X nnz exceeds int32")
        ci = np.empty(nnz, dtype=np.int32)
        L.okmc_x_pattern(*args, _p(rp), _p(ci))
        data = np.empty(nnz)
        L.okmc_x_values(Na, _p(ax), _p(ay), _p(az), _p(ael), _p(aq), _p(acb), _p(self.lattice), int(p.pbc),
                        C.c_double(p.nn_dist), _p(self.metals), len(self.metals), C.c_double(p.X_tol),
                        C.c_double(p.X_high_G), C.c_double(p.X_low_G), C.c_double(p.X_loop_G),
                        C.c_double(p.m_e), C.c_double(p.V0), n_src, n_gnd, p.num_layers_contact,
                        _p(rp), _p(ci), _p(data))
        return dict(Na=Na, atom_site=atom_site, ael=ael, row_ptr=rp[:Na + 2], col=ci, data=data)

    def x_rows_apply(self, rows, m):
        """(diag_i, (X m)_i) of the listed atom rows (node index >= 2) computed on the fly from the current state, without
        assembling X (okmc_x_rows_apply): a full-size check of a solution on sampled rows."""
        L = lib(); p = self.p
        atom_site = np.empty(self.N, dtype=np.int32)
        Na = L.okmc_compact_atoms(self.N, _p(self.element), _p(atom_site))
        atom_site = atom_site[:Na].copy()
        ax, ay, az = self.x[atom_site].copy(), self.y[atom_site].copy(), self.z[atom_site].copy()
        ael = self.element[atom_site].copy(); aq = self.charge[atom_site].copy(); acb = self.CB_edge[atom_site].copy()
        an = np.empty((Na, self.nn), dtype=np.int32)
        L.okmc_atom_neighbors(self.N, self.nn, _p(self.neigh), Na, _p(atom_site), _p(an))
        rows = _i32(rows); m = _f64(m)
        assert rows.min() >= 2 and rows.max() <= Na
        diag = np.empty(len(rows)); axm = np.empty(len(rows))
        n = p.num_atoms_first_layer
        L.okmc_x_rows_apply(Na, self.nn, _p(an), _p(ax), _p(ay), _p(az), _p(ael), _p(aq), _p(acb), _p(self.lattice), int(p.pbc),
                            C.c_double(p.nn_dist), _p(self.metals), len(self.metals), C.c_double(p.X_tol), C.c_double(p.X_high_G),
                            C.c_double(p.X_low_G), C.c_double(p.m_e), C.c_double(p.V0), n, n, p.num_layers_contact,
                            len(rows), _p(rows), _p(m), _p(diag), _p(axm))
        return diag, axm

    def update_power(self, Vd, tol=None, heating=None):
        L = lib(); p = self.p
        tol = p.cg_tol if tol is None else tol
        t0 = time.perf_counter()
        X = self.assemble_X()
        t1 = time.perf_counter()
        Na = X["Na"]; Nsub = Na + 1
        rhs = np.zeros(Nsub); rhs[0] = -p.X_loop_G * Vd; rhs[1] = p.X_loop_G * Vd
        guess = self.virtual_potentials[:Nsub].copy()
        sol, it, rr = cg_jacobi(X["row_ptr"], X["col"], X["data"], rhs, guess, tol=tol)
        self.virtual_potentials[:Nsub] = sol
        t2 = time.perf_counter()
        self.virtual_potentials[:Na + 2] *= p.G0                     # :1015-1016 (in place: next warm start is G0 * m)
        m = self.virtual_potentials
        if self.sem == "cuda":
            self.imacro = L.okmc_imacro_row1(_p(X["row_ptr"]), _p(X["col"]), _p(X["data"]), _p(m))
        else:
            self.imacro = L.okmc_imacro_row0(_p(X["row_ptr"]), _p(X["col"]), _p(X["data"]), _p(m), C.c_double(p.X_high_G))
        heating = (p.solve_heating_global or p.solve_heating_local) if heating is None else heating
        if heating:
            L.okmc_dissipated_power(Na, _p(X["row_ptr"]), _p(X["col"]), _p(X["data"]), _p(m), C.c_double(Vd),
                                    _p(X["ael"]), _p(X["atom_site"]), _p(self.metals), len(self.metals),
                                    C.c_double(1.0), _p(self.power))
        t3 = time.perf_counter()
        self.timing["current"] = t3 - t0
        self.timing["current_assemble"] = t1 - t0
        self.timing["current_solve"] = t2 - t1
        self.stats["cg_iters_X"] = it
        self.stats["X_nnz"] = int(len(X["col"]))
        self.last_X = X
        return self.imacro

    # --- global temperature (heat_solver.cpp:316-350 as run; heat_solver_gpu.cu:42-48 unused) -
    def update_temperature_global(self, step_time, mode=0):
        p = self.p
        t0 = time.perf_counter()
        P = C.c_double(0)
        self.T_bg = lib().okmc_temperature_global(self.N, _p(self.power), self.T_bg, step_time, mode,
                                                  p.dissipation_constant, p.background_temp, p.t_ox, p.A, p.c_p,
                                                  p.small_step, C.byref(P))
        self.timing["heat"] = time.perf_counter() - t0
        return self.T_bg

    # --- one KMC superstep (kmc_main.cpp:175-279) --------------------------------------------
    def superstep(self, Vd):
        p = self.p
        out = {}
        if p.solve_potential:
            self.update_charge()
            self.update_potential(Vd)
        step_time = 0.0
        if p.perturb_structure:
            step_time = self.execute_kmc_step()
        out["step_time"] = step_time
        if p.solve_current:
            out["imacro"] = self.update_power(Vd)
            if p.solve_heating_global:
                out["T_bg"] = self.update_temperature_global(step_time)
        return out
