#!/usr/bin/env python3
"""CPU experiment (round 5, for DESIGN.md "Open points"): where does the ill-conditioning of X live -- in its sparse neighbour part or in the dense
tunnelling block?  The oracle's X of the 2.5 nm / 7.5 nm device, Jacobi-scaled as the solver scales it, split into
    Xn = the entries between neighbouring atoms (+ the two driver rows / columns + the full diagonal),    Xt = X - Xn (tunnelling),
and solved by preconditioned CG from a zero start to the reference's stop test (||r||_2 <= tol on the scaled system) with
    jacobi   no further preconditioner (the scaled system as the product solves it)
    Xn       M = Xn, applied exactly (sparse LU): the best any preconditioner built on the neighbour part alone can do
    ic(Xn)   M = Xn approximated by k sweeps of a (damped) Jacobi iteration on Xn (what a device kernel could afford per sweep)
usage: python tools/precond_proto.py [2.5nm|7.5nm] [tol]"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spl

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import blockcg_proto as bp  # noqa: E402
from oracle import oracle as oc  # noqa: E402


def pcg(A, b, Minv, tol, maxit=20000):
    x = np.zeros_like(b); r = b.copy(); z = Minv(r); p = z.copy(); rz = r @ z; it = 0
    while np.sqrt(r @ r) > tol and it < maxit:
        t = A @ p; al = rz / (p @ t); x += al * p; r -= al * t; z = Minv(r); rzn = r @ z; p = z + (rzn / rz) * p; rz = rzn; it += 1
    return x, it


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "2.5nm"
    tol = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-6
    t0 = time.perf_counter()
    As, bs, sc, o = bp.system(name)
    m = As.shape[0]
    # neighbour pattern among atoms (node = atom + 2), from the oracle's neighbour index
    L = oc.lib()
    atom_site = np.empty(o.N, dtype=np.int32)
    Na = L.okmc_compact_atoms(o.N, oc._p(o.element), oc._p(atom_site)); atom_site = atom_site[:Na].copy()
    an = np.empty((Na, o.nn), dtype=np.int32)
    L.okmc_atom_neighbors(o.N, o.nn, oc._p(o.neigh), Na, oc._p(atom_site), oc._p(an))
    rows = np.repeat(np.arange(Na), o.nn); cols = an.ravel(); keep = cols >= 0
    Pn = sp.csr_matrix((np.ones(keep.sum()), (rows[keep] + 2, cols[keep] + 2)), shape=(Na + 2, Na + 2))[:m, :m]
    Pn = ((Pn + Pn.T) > 0).astype(np.float64).tolil()
    Pn[0:2, :] = 1.0; Pn[:, 0:2] = 1.0; Pn.setdiag(1.0)
    Pn = Pn.tocsr()
    Xn = As.multiply(Pn).tocsr(); Xt = (As - Xn).tocsr(); Xt.eliminate_zeros()
    print("%s: %d rows, nnz %d = neighbour part %d + tunnelling %d  (set-up %.1f s)" % (name, m, As.nnz, Xn.nnz, Xt.nnz, time.perf_counter() - t0), flush=True)
    print("  ||Xt||_F / ||Xn||_F = %.3e" % (spl.norm(Xt) / spl.norm(Xn)), flush=True)
    res = {}
    t1 = time.perf_counter(); _, res["jacobi"] = pcg(As, bs, lambda r: r, tol); print("  jacobi (the product's system): %d iterations (%.1f s)" % (res["jacobi"], time.perf_counter() - t1), flush=True)
    t1 = time.perf_counter(); lu = spl.splu(Xn.tocsc()); _, res["Xn"] = pcg(As, bs, lu.solve, tol); print("  M = Xn exactly (sparse LU):    %d iterations (%.1f s)" % (res["Xn"], time.perf_counter() - t1), flush=True)
    for k in (4, 16):
        def mk(r, k=k):
            z = r.copy()                                     # Xn has unit diagonal (Jacobi-scaled): z <- z + w (r - Xn z)
            for _ in range(k):
                z = z + 0.7 * (r - Xn @ z)
            return z
        t1 = time.perf_counter(); _, it = pcg(As, bs, mk, tol); res["jac%d" % k] = it
        print("  M^-1 = %d damped Jacobi sweeps on Xn: %d iterations (%.1f s)" % (k, it, time.perf_counter() - t1), flush=True)
    # what an ITERATIVE solve of Xn costs: CG on Xn alone (unit diagonal) for the first residual, to three relative tolerances
    def cg_its(A, b, rtol, maxit=200000):
        x = np.zeros_like(b); r = b.copy(); p_ = r.copy(); rr = r @ r; r0 = np.sqrt(rr); it = 0
        while np.sqrt(rr) > rtol * r0 and it < maxit:
            t = A @ p_; al = rr / (p_ @ t); x += al * p_; r -= al * t; rn = r @ r; p_ = r + (rn / rr) * p_; rr = rn; it += 1
        return x, it
    for rtol in (1e-2, 1e-4, 1e-8):
        _, it = cg_its(Xn, bs, rtol); res["cg_Xn_%g" % rtol] = it
        print("  CG on Xn alone, right-hand side b, relative residual %g: %d iterations" % (rtol, it), flush=True)
    # inexact preconditioning: k CG iterations on Xn per application (a different operator every time: flexible outer CG, Polak-Ribiere beta)
    def fpcg(A, b, Minv, tol, maxit=5000):
        x = np.zeros_like(b); r = b.copy(); z = Minv(r); p_ = z.copy(); rz = r @ z; it = 0
        while np.sqrt(r @ r) > tol and it < maxit:
            t = A @ p_; al = rz / (p_ @ t); x += al * p_; rn = r - al * t; z = Minv(rn); rzn = rn @ z; beta = (z @ (rn - r)) / rz; r = rn; p_ = z + beta * p_; rz = rzn; it += 1
        return x, it
    for k in (10, 30, 100, 300):
        def mk2(r, k=k):
            x = np.zeros_like(r); rr_ = r.copy(); p2 = rr_.copy(); q = rr_ @ rr_
            for _ in range(k):
                if q == 0.0: break
                t = Xn @ p2; al = q / (p2 @ t); x += al * p2; rr_ -= al * t; qn = rr_ @ rr_; p2 = rr_ + (qn / q) * p2; q = qn
            return x
        t1 = time.perf_counter(); _, it = fpcg(As, bs, mk2, tol); res["inner_cg%d" % k] = it
        print("  M^-1 = %d CG iterations on Xn (flexible outer CG): %d outer iterations = %d products with X + %d with Xn (%.1f s)" % (k, it, it, it * k, time.perf_counter() - t1), flush=True)
    # the other way round: M = I + Xt restricted to its own block diagonal?  (tunnelling block alone, exact)
    S = np.flatnonzero(np.asarray(abs(Xt).sum(axis=1)).ravel() > 0)
    print("  tunnelling set |S| = %d" % len(S), flush=True)
    if len(S) <= 12000:
        TS = (As[S][:, S]).toarray()
        import scipy.linalg as sl
        cf = sl.cho_factor(TS)
        def mt(r):
            z = r.copy(); z[S] = sl.cho_solve(cf, r[S]); return z
        t1 = time.perf_counter(); _, it = pcg(As, bs, mt, tol); res["TS"] = it
        print("  M = diag outside S, X[S,S] exactly inside: %d iterations (%.1f s)" % (it, time.perf_counter() - t1), flush=True)
    print(res)


if __name__ == "__main__":
    main()
