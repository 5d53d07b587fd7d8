#!/usr/bin/env python3
"""CPU experiment behind the block-CG of csrc/xtb.hip: the oracle's X of the 2.5 nm / 7.5 nm device (or a k x k
tiling), Jacobi-scaled as solve_sparse_CG_Jacobi scales it (iterative_solvers_gpu.cu:349-459), solved by

  cg      the reference's single-vector loop (sign convention r = A y - b, p = -r)
  bcg     block-CG over s columns: column 0 = the physical right-hand side and start vector, columns 1..s-1 = fixed-seed auxiliary
          right-hand sides with a zero start; the stop test is the reference's, on column 0 only.
          One sweep T = A P per iteration; every s x s matrix the iteration needs (P'T, P'R, T'R, T'T, R'R) comes from ONE pass over
          the panels, so that the device loop needs no second global reduction:
              c    = -(P'T)^-1 P'R            Y += P c ; R+ = R + T c
              beta =  (P'T)^-1 (T'R + T'T c)  (= (P'T)^-1 T'R+)
              P+   = (-R+ + P beta) W         W = inverse Cholesky factor of the Gram matrix of (-R+ + P beta), formed from the
                                              same s x s matrices (no pass over the new panel): directions stay orthonormal.
usage: python tools/blockcg_proto.py [2.5nm|7.5nm|tile:K] [s ...]"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from devicekmc_amd import params, structure  # noqa: E402
from oracle import oracle as oc  # noqa: E402


def load(name):
    g = os.path.join(ROOT, "tests", "golden")
    if name == "7.5nm":
        s = structure.load_structure(os.path.join(g, "device_7.5nm.npz"))
        p = params.KMCParameters(rnd_seed=5, lattice=tuple(s.meta["lattice"]), num_atoms_first_layer=1296, num_atoms_contact=12960,
                                 A=76.725e-10 * 76.725e-10)
    elif name == "2.5nm":
        s = structure.load_structure(os.path.join(g, "device_2.5nm.npz")); p = params.KMCParameters()
    else:
        k = int(name.split(":")[1])
        cell = structure.load_structure(os.path.join(g, "device_2.5nm.npz"))
        s = structure.tile_structure(cell, k, 25.575, 25.575, 1440); p = params.KMCParameters().for_tiling(k)
    return s, p


def system(name, Vd=5.0, steps=1):
    s, p = load(name)
    o = oc.OracleKMC(s.element, s.x, s.y, s.z, p)
    o.set_laplace_potential(Vd)
    for _ in range(steps):
        o.update_charge(); o.update_potential(Vd); o.execute_kmc_step()
        X = o.assemble_X()
    Na = X["Na"]; m = Na + 1
    A = sp.csr_matrix((X["data"], X["col"], X["row_ptr"][:m + 1]), shape=(m, Na + 2))[:, :m].tocsr()
    b = np.zeros(m); b[0] = -p.X_loop_G * Vd; b[1] = p.X_loop_G * Vd
    d = A.diagonal(); sc = 1.0 / np.sqrt(d)
    As = sp.diags(sc) @ A @ sp.diags(sc)
    return As.tocsr(), b * sc, sc, o


def cg(A, b, y0, tol=1e-6, maxit=100000):
    y = y0.copy(); r = A @ y - b; p = -r; rr = r @ r; it = 0
    if not np.sqrt(rr) > tol * tol:
        return y, 0
    while True:
        t = A @ p; al = rr / (p @ t); y += al * p; r += al * t; rn = r @ r; it += 1
        if not rn > tol * tol or it >= maxit:
            return y, it
        p = (rn / rr) * p - r; rr = rn


def aux_rhs(m, s, seed=12345):
    """fixed-seed auxiliary columns: a hash of (row, column) -> uniform in [-1, 1) (the device generator of xtb.hip uses the same)"""
    i = np.arange(m, dtype=np.uint64)[:, None]; j = np.arange(1, s, dtype=np.uint64)[None, :]
    h = (i * np.uint64(0x9E3779B97F4A7C15) + j * np.uint64(0xC2B2AE3D27D4EB4F) + np.uint64(seed)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    h ^= h >> np.uint64(29); h = (h * np.uint64(0xBF58476D1CE4E5B9)) & np.uint64(0xFFFFFFFFFFFFFFFF); h ^= h >> np.uint64(32)
    return (h >> np.uint64(11)).astype(np.float64) / 2.0 ** 52 - 1.0


def bcg(A, b, y0, s, tol=1e-6, maxit=100000, ortho=True, verbose=False):
    m = len(b)
    B = np.empty((m, s)); B[:, 0] = b; B[:, 1:] = aux_rhs(m, s) * np.linalg.norm(b) / np.sqrt(m)
    Y = np.zeros((m, s)); Y[:, 0] = y0
    R = A @ Y - B
    if not np.sqrt(R[:, 0] @ R[:, 0]) > tol * tol:
        return Y[:, 0], 0
    Grr = R.T @ R
    L = np.linalg.cholesky(Grr)
    P = np.linalg.solve(L, -R.T).T                      # orthonormal directions: P'P = I
    Gpp = np.eye(s)
    it = 0
    while True:
        T = A @ P
        # ---- everything below needs only s x s matrices formed in ONE pass over P, R, T ----
        Gpt = P.T @ T; Gpr = P.T @ R; Gtr = T.T @ R; Gtt = T.T @ T
        Gpt = 0.5 * (Gpt + Gpt.T)
        c = -np.linalg.solve(Gpt, Gpr)
        Y += P @ c
        Rn = R + T @ c
        it += 1
        Grr_n = Grr + c.T @ Gtr + Gtr.T @ c + c.T @ Gtt @ c          # = Rn'Rn without a pass over Rn
        Grr_direct = Rn.T @ Rn
        if verbose and it % 20 == 0:
            print("  it %4d  rr0 %.3e  recurrence err %.1e  cond(P'T) %.1e" % (it, Grr_direct[0, 0], abs(Grr_n - Grr_direct).max() / abs(Grr_direct).max(), np.linalg.cond(Gpt)))
        rr0 = Grr_direct[0, 0]
        if not rr0 > tol * tol or it >= maxit:
            return Y[:, 0], it
        beta = np.linalg.solve(Gpt, Gtr + Gtt @ c)
        Pn = -Rn + P @ beta
        if ortho:
            Gpr_n = Gpr + Gpt @ c                                    # P'Rn (zero in exact arithmetic)
            G = Grr_direct - beta.T @ Gpr_n - Gpr_n.T @ beta + beta.T @ Gpp @ beta
            G = 0.5 * (G + G.T)
            Lc = np.linalg.cholesky(G)
            Pn = np.linalg.solve(Lc, Pn.T).T
        P = Pn; R = Rn; Grr = Grr_direct


if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "2.5nm"
    ss = [int(x) for x in sys.argv[2:]] or [2, 4, 8, 16]
    t0 = time.time()
    A, b, sc, o = system(name)
    m = len(b)
    print("%s: %d rows, %d nnz  (assembled in %.0f s)" % (name, m, A.nnz, time.time() - t0), flush=True)
    y0 = np.zeros(m)
    t0 = time.time(); y1, it1 = cg(A, b, y0); print("cg: %d sweeps  [%.0f s]" % (it1, time.time() - t0), flush=True)
    for s in ss:
        for ortho in (True,):
            t0 = time.time(); ys, its = bcg(A, b, y0, s, ortho=ortho, verbose=bool(os.environ.get("V")))
            print("bcg s=%2d ortho=%d: %d sweeps, rel diff to cg %.2e, true ||r0|| %.2e  [%.0f s]" %
                  (s, ortho, its, np.linalg.norm(ys - y1) / np.linalg.norm(y1), np.linalg.norm(A @ ys - b), time.time() - t0), flush=True)
