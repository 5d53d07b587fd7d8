#!/usr/bin/env python3
"""Development aid (GPU): tiled X (dkmc_set_x_format(1)) against the CSR X path and the oracle on one workload.
usage: python tools/check_xt.py [2.5nm|7.5nm|tile:K] [steps]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bench import make_workload  # noqa: E402
from devicekmc_amd import host, lib  # noqa: E402


def run(name, fmt, nsteps, want_x):
    L = lib.load()
    L.dkmc_set_x_format(fmt)
    s, p = make_workload(name)
    dev = host.Device(s, p, gpu_neighbors="cuda:0")
    sim = host.KMCProcess(dev, p.freq)
    gb = dev.make_gpubuf("cuda:0")
    dev.setLaplacePotential(gb, p, 5.0)
    gb.sync_HostToGPU(dev)
    out = []
    for k in range(nsteps):
        dev.updateCharge(gb); dev.updatePotential(gb, p, 5.0, k)
        _, dt = sim.executeKMCStep(gb, dev)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        dev.updatePower(gb, p, 5.0)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        dev.updateTemperature(gb, p, dt)
        st = host.get_stats()
        rec = dict(dt=dt, imacro=dev.imacro, T=dev.T_bg, iters=st["cg_iters_X"], nnz=st["X_nnz"], ms=(t1 - t0) * 1e3,
                   m=gb.atom_virtual_potentials.cpu().numpy().copy(), power=gb.site_power.cpu().numpy().copy(), st=st)
        if want_x and k == 0:
            rec["X"] = host.get_last_X()
        out.append(rec)
    del gb, sim, dev
    torch.cuda.empty_cache()
    return out


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "2.5nm"
    nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    want_x = name in ("2.5nm", "tile:2")
    a = run(name, 0, nsteps, want_x)
    b = run(name, 1, nsteps, want_x)
    ok = True
    for k, (ra, rb) in enumerate(zip(a, b)):
        dm = np.max(np.abs(ra["m"] - rb["m"])) / max(np.max(np.abs(ra["m"])), 1e-300)
        dp = np.max(np.abs(ra["power"] - rb["power"])) / max(np.max(np.abs(ra["power"])), 1e-300)
        di = abs(ra["imacro"] - rb["imacro"]) / abs(ra["imacro"])
        print("step %d: csr I=%.9e iters=%d nnz=%d %.2f ms | tiled I=%.9e iters=%d nnz=%d %.2f ms | rel dI=%.2e dm=%.2e dP=%.2e dt_equal=%s"
              % (k, ra["imacro"], ra["iters"], ra["nnz"], ra["ms"], rb["imacro"], rb["iters"], rb["nnz"], rb["ms"], di, dm, dp, ra["dt"] == rb["dt"]))
        ok &= ra["nnz"] == rb["nnz"] and di < 1e-5 and ra["dt"] == rb["dt"]
    st = b[-1]["st"]
    print("tiled stats: ns=%d tiles=%d subblocks=%d items=%d kc=%d sparse_nnz=%d tile_entries=%d" %
          (st["xt_ns"], st["spmv_tiles"], st["xt_subblocks"], st["xt_items"], st["xt_kc"], st["xt_sparse_nnz"], st["spmv_tile_entries"]))
    if want_x:
        (rpa, cia, da), (rpb, cib, db) = a[0]["X"], b[0]["X"]
        same_pat = np.array_equal(rpa, rpb) and np.array_equal(cia, cib)
        dv = np.max(np.abs(da - db) / np.maximum(np.abs(da), 1e-300)) if same_pat else float("nan")
        print("get_last_X: pattern identical = %s, max rel value diff = %.3e" % (same_pat, dv))
        ok &= same_pat and dv < 1e-10
    print("CHECK_XT", "OK" if ok else "FAILED")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
