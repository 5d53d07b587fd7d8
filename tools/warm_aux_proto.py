#!/usr/bin/env python3
"""CPU experiment (round 5): what should the AUXILIARY columns of the block-CG start from, once column 0 starts from the previous solution?
The oracle's X of consecutive supersteps (Jacobi-scaled), solved by the block-CG of tools/blockcg_proto.py with
  ref    every column from zero (column 0: the reference's start)
  warm0  column 0 from the previous step's solution, auxiliary columns from zero     (the library default of round 5)
  warmA  ALL columns from the previous step's solutions (the auxiliary right-hand sides do not change between steps)
usage: python tools/warm_aux_proto.py [2.5nm|7.5nm|tile:K] [steps] [s]"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from blockcg_proto import aux_rhs, load  # noqa: E402
from oracle import oracle as oc  # noqa: E402


def smooth_rhs(o, X, s, sc):
    """the smooth auxiliary set of csrc/xtb.hip (xtb_rhs / k_xtb_modes): column v = cos(kx pi xi) cos(ky pi eta) cos(kz pi zeta) / s_row with
    (kx, ky, kz) the v-th lowest Laplacian mode of the bounding box of the atoms"""
    a = X["atom_site"]
    pos = np.stack([o.x[a], o.y[a], o.z[a]], axis=1)
    lo, hi = pos.min(axis=0), pos.max(axis=0)
    L = np.maximum(hi - lo, 1e-6)
    cand = []
    for code in range(1, 512):
        kx, ky, kz = code >> 6, (code >> 3) & 7, code & 7
        cand.append(((kx / L[0]) ** 2 + (ky / L[1]) ** 2 + (kz / L[2]) ** 2, -code, (kx, ky, kz)))
    cand.sort()
    modes = [c[2] for c in cand[:s - 1]]
    m = len(sc)
    u = np.empty((m, 3)); u[0] = (1.0, 0.5, 0.5); u[1] = (0.0, 0.5, 0.5)
    u[2:] = ((pos - lo) / L)[:m - 2]
    B = np.empty((m, s - 1))
    for v, (kx, ky, kz) in enumerate(modes):
        B[:, v] = np.cos(np.pi * kx * u[:, 0]) * np.cos(np.pi * ky * u[:, 1]) * np.cos(np.pi * kz * u[:, 2]) / sc
    return B


def assemble(o, p, Vd, want_X=False):
    X = o.assemble_X()
    if want_X:
        assemble.X = X
    Na = X["Na"]; m = Na + 1
    A = sp.csr_matrix((X["data"], X["col"], X["row_ptr"][:m + 1]), shape=(m, Na + 2))[:, :m].tocsr()
    b = np.zeros(m); b[0] = -p.X_loop_G * Vd; b[1] = p.X_loop_G * Vd
    sc = 1.0 / np.sqrt(A.diagonal())
    return (sp.diags(sc) @ A @ sp.diags(sc)).tocsr(), b * sc, sc


def bcg(A, B, Y0, tol=1e-6, maxit=20000):
    """block-CG on all columns of B from Y0; stop test on column 0; returns (Y, sweeps, breakdowns)"""
    s = B.shape[1]
    Y = Y0.copy()
    R = A @ Y - B
    if not np.sqrt(R[:, 0] @ R[:, 0]) > tol * tol:
        return Y, 0, 0
    bad = 0

    def orth(M):
        nonlocal bad
        G = M.T @ M; G = 0.5 * (G + G.T)
        try:
            L = np.linalg.cholesky(G)
            return np.linalg.solve(L, M.T).T
        except np.linalg.LinAlgError:
            bad += 1
            return M / np.sqrt(np.maximum(np.diag(G), 1e-300))
    P = orth(-R)
    it = 0
    while True:
        T = A @ P
        Gpt = P.T @ T; Gpt = 0.5 * (Gpt + Gpt.T)
        try:
            c = -np.linalg.solve(Gpt, P.T @ R)
        except np.linalg.LinAlgError:
            return Y, it, -1
        Y += P @ c
        R = R + T @ c
        it += 1
        if not R[:, 0] @ R[:, 0] > tol * tol or it >= maxit:
            return Y, it, bad
        beta = np.linalg.solve(Gpt, T.T @ R)
        P = orth(-R + P @ beta)


if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "2.5nm"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    s = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    Vd = 5.0
    st, p = load(name)
    o = oc.OracleKMC(st.element, st.x, st.y, st.z, p)
    o.set_laplace_potential(Vd)
    Yprev = None
    for k in range(steps):
        o.update_charge(); o.update_potential(Vd); o.execute_kmc_step()
        t0 = time.time()
        A, b, sc = assemble(o, p, Vd, want_X=True)
        m = len(b)
        B = np.empty((m, s)); B[:, 0] = b
        aux_mode = os.environ.get("AUX", "hash")
        if aux_mode == "smooth":
            B[:, 1:] = smooth_rhs(o, assemble.X, s, sc)
        elif aux_mode == "mix":            # columns 1 .. s/2 - 1 smooth (always from zero), the rest hash (warm-started in warmA)
            h = s // 2
            B[:, 1:h] = smooth_rhs(o, assemble.X, h, sc)
            B[:, h:] = (aux_rhs(m, s) * np.linalg.norm(b) / np.sqrt(m))[:, h - 1:]
        else:
            B[:, 1:] = aux_rhs(m, s) * np.linalg.norm(b) / np.sqrt(m)
        Z = np.zeros((m, s))
        Yr, it_ref, _ = bcg(A, B, Z)
        line = "step %d (%d events, %d rows): ref %d" % (k, o.last_events["n"], m, it_ref)
        if Yprev is not None and Yprev.shape[0] == m:
            # the previous SCALED solutions belong to the previous scaling: unscale with the old, rescale with the new
            Yw = Yprev / sc_prev[:, None] * sc[:, None]
            Y0 = Z.copy(); Y0[:, 0] = Yw[:, 0]
            _, it_w0, _ = bcg(A, B, Y0)
            if aux_mode == "mix":
                Yw = Yw.copy(); Yw[:, 1:s // 2] = 0.0
            _, it_wa, bad = bcg(A, B, Yw)
            r0 = A @ Yw - B
            line += "  warm0 %d  warmA %d (rescaled directions: %d)   |r0| col0 %.2e, aux %.2e..%.2e" % (
                it_w0, it_wa, bad, np.linalg.norm(r0[:, 0]), np.linalg.norm(r0[:, 1:], axis=0).min(), np.linalg.norm(r0[:, 1:], axis=0).max())
        print(line + "  [%.0f s]" % (time.time() - t0), flush=True)
        Yprev, sc_prev = Yr, sc
