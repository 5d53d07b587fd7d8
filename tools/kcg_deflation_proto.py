"""CPU experiment (round 4): deflated CG on the oracle's K with the lowest Laplacian modes of the bounding box (over the Jacobi
scaling) as the coarse space -- the K-side counterpart of the smooth auxiliary columns of the block-CG on X (csrc/xtb.hip).
(The GPU version was built and removed in round 4: on the warm-started production solves the gain was 742 -> 543 iterations at 2.7 x the cost per iteration.)
usage: python tools/kcg_deflation_proto.py [crossbar|7.5nm|tile:K] [tol]"""
import os, sys, time
import numpy as np, scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from devicekmc_amd import params, structure
from oracle import oracle as oc
import ctypes as C
name = sys.argv[1] if len(sys.argv) > 1 else "crossbar"
if name == "crossbar":
    import bench
    s, p = bench.make_workload("crossbar_10nm_5pitch"); p = p.log_revision(); Vd = 1.0
elif name == "7.5nm":
    import bench; s, p = bench.make_workload("7.5nm"); Vd = 5.0
else:
    import bench; s, p = bench.make_workload(name); Vd = 5.0
tol = float(sys.argv[2]) if len(sys.argv) > 2 else p.cg_tol
o = oc.OracleKMC(s.element, s.x, s.y, s.z, p)
o.set_laplace_potential(Vd); o.update_charge()
if o._K is None: o.initialize_sparsity()
nl, m, ((rp, ci), (lrp, lci), (rrp, rci)) = o._K
data = np.zeros(len(ci)); rhs = np.zeros(m)
_p = oc._p
oc.lib().okmc_k_assemble(o.N, nl, nl, _p(o.element), _p(o.charge), _p(o.metals), len(o.metals), C.c_double(p.high_G), C.c_double(p.low_G), 0,
                         _p(rp), _p(ci), _p(lrp), _p(lci), _p(rrp), _p(rci), C.c_double(-Vd / 2), C.c_double(Vd / 2), _p(data), _p(rhs))
K = sp.csr_matrix((data, ci, rp), shape=(m, m))
d = K.diagonal(); sc = 1 / np.sqrt(d)
A = (sp.diags(sc) @ K @ sp.diags(sc)).tocsr(); b = rhs * sc
print(name, m, "rows", K.nnz, "nnz; tol", tol, flush=True)
def cg(A, b, x0, tol, W=None):
    x = x0.copy()
    if W is not None:
        AW = A @ W; E = W.T @ AW; Ei = np.linalg.inv(E)
        r = b - A @ x; x += W @ (Ei @ (W.T @ r))
    r = b - A @ x
    proj = (lambda v: v - W @ (Ei @ (AW.T @ v))) if W is not None else (lambda v: v)
    pdir = proj(r); rr = r @ r; it = 0
    while rr > tol * tol and it < 20000:
        t = A @ pdir; al = rr / (pdir @ t); x += al * pdir; r -= al * t; rn = r @ r; it += 1
        pdir = proj(r) + (rn / rr) * pdir; rr = rn
    return x, it, np.linalg.norm(b - A @ x)
x0 = np.zeros(m)
t0 = time.time(); x1, it1, res = cg(A, b, x0, tol); print("plain CG (zero start): %d iterations, ||r|| %.2e [%.0f s]" % (it1, res, time.time() - t0), flush=True)
X = np.c_[o.x[nl:nl + m], o.y[nl:nl + m], o.z[nl:nl + m]]
lo, hi = X.min(0), X.max(0); U = (X - lo) / (hi - lo); L = hi - lo
cands = sorted([((kx / L[0]) ** 2 + (ky / L[1]) ** 2 + (kz / L[2]) ** 2, -kx, -ky, -kz) for kx in range(8) for ky in range(8) for kz in range(8)])
for nm in (15, 31, 63):
    modes = [(-c[1], -c[2], -c[3]) for c in cands[:nm + 1]]        # includes the constant
    W = np.c_[[np.cos(kx * np.pi * U[:, 0]) * np.cos(ky * np.pi * U[:, 1]) * np.cos(kz * np.pi * U[:, 2]) / sc for kx, ky, kz in modes]].T
    W, _ = np.linalg.qr(W)
    t0 = time.time(); x2, it2, res2 = cg(A, b, x0, tol, W); print("deflated CG, %d box modes / s: %d iterations, ||r|| %.2e, rel diff %.1e [%.0f s]" % (nm + 1, it2, res2, np.linalg.norm(x2 - x1) / np.linalg.norm(x1), time.time() - t0), flush=True)
