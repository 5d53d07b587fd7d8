#!/usr/bin/env python3
"""CPU experiment (round 5): the block-CG of csrc/xtb.hip on the SPLIT-preconditioned operator L A L, L = the degree-d truncation of the series of
(I - N)^(-1/2), N = I - An (An = neighbour part + full diagonal of the Jacobi-scaled X: unit diagonal) -- a preconditioner that costs 2 d sparse panel
products per sweep and leaves the block loop's algebra untouched (it just sees another SPD operator).  Sweeps of the block loop (width 16, zero start,
stop on the TRUE residual of the unpreconditioned scaled system) for d = 0 (the product's system), 1, 2, 4, 8.
usage: python tools/precond_block_proto.py [2.5nm|7.5nm] [width]"""
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


class SplitOp:
    """v -> L (A (L v)), L = sum_j c_j N^j (c = binomial series of (1 - x)^(-1/2))"""
    def __init__(self, A, N, d):
        self.A, self.N, self.d = A, N, d
        c = [1.0]
        for j in range(1, d + 1):
            c.append(c[-1] * (2 * j - 1) / (2 * j))
        self.c = c
        self.shape = A.shape
        self.nnz = A.nnz

    def L(self, V):
        out = self.c[0] * V; W = V
        for j in range(1, self.d + 1):
            W = self.N @ W; out = out + self.c[j] * W
        return out

    def __matmul__(self, V):
        return self.L(self.A @ self.L(V))


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "2.5nm"
    s = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    As, bs, sc, o = bp.system(name)
    m = As.shape[0]
    L_ = oc.lib()
    atom_site = np.empty(o.N, dtype=np.int32)
    Na = L_.okmc_compact_atoms(o.N, oc._p(o.element), oc._p(atom_site)); atom_site = atom_site[:Na].copy()
    an = np.empty((Na, o.nn), dtype=np.int32)
    L_.okmc_atom_neighbors(o.N, o.nn, oc._p(o.neigh), Na, oc._p(atom_site), oc._p(an))
    rows = np.repeat(np.arange(Na), o.nn); cols = an.ravel(); keep = cols >= 0
    Pn = sp.csr_matrix((np.ones(keep.sum()), (rows[keep] + 2, cols[keep] + 2)), shape=(Na + 2, Na + 2))[:m, :m]
    Pn = ((Pn + Pn.T) > 0).astype(np.float64).tolil()
    if not os.environ.get("NO_DRIVERS"):
        Pn[0:2, :] = 1.0; Pn[:, 0:2] = 1.0
    else:
        Pn[0:2, :] = 0.0; Pn[:, 0:2] = 0.0                    # the two driver nodes' couplings stay outside the preconditioner
    Pn.setdiag(1.0); Pn = Pn.tocsr()
    An = As.multiply(Pn).tocsr()
    N = (sp.identity(m, format="csr") - An).tocsr(); N.eliminate_zeros()
    ev = spl.eigsh(N, k=1, which="LA", return_eigenvectors=False)[0]
    print("%s: %d rows; neighbour part %d nnz of %d; largest eigenvalue of N = I - An: %.6f" % (name, m, An.nnz, As.nnz, ev), flush=True)
    y0 = np.zeros(m)
    for d in (0, 1, 2, 4, 8):
        op = SplitOp(As, N, d)
        bh = op.L(bs)
        t0 = time.time()
        # stop on the recurrence residual of the preconditioned system at a tolerance tightened until the TRUE residual of the product's system meets 1e-6
        tol = 1e-6
        for attempt in range(4):
            yh, its = bp.bcg(op, bh, y0, s, tol=tol)
            y = op.L(yh)
            true_r = np.linalg.norm(As @ y - bs)
            if true_r <= 1e-6:
                break
            tol *= 0.3
        print("  d = %d: %3d sweeps (each: 1 product with X + %2d with the neighbour part, 16 columns); true residual %.2e (inner tol %.1e)  [%.0f s]"
              % (d, its, 2 * d, true_r, tol, time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
