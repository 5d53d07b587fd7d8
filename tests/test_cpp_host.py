"""The reference's host language is C++: run the C++ driver that goes through the drop-in shim
(include/gpu_buffers.h + include/gpu_solvers.h -> C ABI) and compare it with the oracle."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_driver_matches_oracle(cell_2p5, tmp_path):
    import __graft_entry__ as g
    g.build()
    from devicekmc_amd import params as pm, structure
    from oracle import oracle as oc
    p = pm.KMCParameters(); p.solve_heating_global = True
    Vd, steps = 5.0, 3
    element, neigh, nn, layer = structure.prepare_device(cell_2p5, p)
    N = cell_2p5.N
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    from devicekmc_amd import io as kio
    kio.write_host_bundle(fin, cell_2p5, p, Vd, element, neigh, nn, layer)
    exe = os.path.join(ROOT, "devicekmc_amd", "host", "kmc_superstep")
    r = subprocess.run([exe, fin, fout, str(steps)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    assert "GPUassert" not in r.stderr, r.stderr
    raw = np.fromfile(fout, dtype=np.uint8)
    off = 0
    def take(dt, n):
        nonlocal off
        a = np.frombuffer(raw, dtype=dt, count=n, offset=off); off += a.nbytes; return a
    dt, I, T, next_u = take(np.float64, steps), take(np.float64, steps), take(np.float64, steps), take(np.float64, 1)[0]
    el, q = take(np.int32, N), take(np.int32, N)
    pb, pc = take(np.float64, N), take(np.float64, N)
    I_sparse, I_split = take(np.float64, 1)[0], take(np.float64, 1)[0]
    o = oc.OracleKMC(cell_2p5.element, cell_2p5.x, cell_2p5.y, cell_2p5.z, p)
    o.set_laplace_potential(Vd)
    for k in range(steps):
        out = o.superstep(Vd)
        assert abs(dt[k] / out["step_time"] - 1) <= 1e-6, k           # both at the reference's CG tolerance (1e-6)
        assert abs(I[k] / out["imacro"] - 1) <= 1e-4, k
        assert abs(T[k] - out["T_bg"]) <= 1e-9
    assert np.array_equal(el, o.element) and np.array_equal(q, o.charge)      # same events executed
    assert next_u == o.rng_kmc.uniform()                                       # caller's RNG left where the reference leaves it
    assert np.abs(pc - o.pot_charge).max() <= 1e-12 * np.abs(o.pot_charge).max()
    # update_power_gpu_split (declared by the reference, gpu_solvers.h:167-172) on the final state agrees with update_power_gpu_sparse
    assert I_sparse > 0 and abs(I_split / I_sparse - 1) <= 1e-4 and abs(I_sparse / I[-1] - 1) <= 1e-3
