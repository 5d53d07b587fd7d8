#!/usr/bin/env python3
"""Soak run on a GPU box: N coupled supersteps of the 2.5 nm device (global heating on), HIP path against the CPU oracle step by step --
same executed events (slot, i, j, type), same RNG position, charges and elements bit for bit, currents / temperature / KMC time within
the CG tolerance.  Not part of the test suite (a minute of oracle time per 50 steps); exits non-zero at the first disagreement.
usage: python tools/soak_vs_oracle.py [steps] [kmc_seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

g.build()
from devicekmc_amd import host, params, structure  # noqa: E402
from oracle import oracle as oc  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
Vd = 5.0
s = structure.load_structure(os.path.join(ROOT, "tests", "golden", "device_2.5nm.npz"))
p = params.KMCParameters(); p.cg_tol = 1e-10; p.solve_heating_global = True; p.rnd_seed_kmc = seed
dev = host.Device(s, p); sim = host.KMCProcess(dev, p.freq)
gb = dev.make_gpubuf("cuda:0")
dev.setLaplacePotential(gb, p, Vd); gb.sync_HostToGPU(dev)
o = oc.OracleKMC(s.element, s.x, s.y, s.z, p)
o.set_laplace_potential(Vd)
worst = dict(I=0.0, T=0.0, dt=0.0, margin=1.0)
nev = 0
for k in range(steps):
    dev.updateCharge(gb); dev.updatePotential(gb, p, Vd, k)
    _, dt = sim.executeKMCStep(gb, dev, want_log=True)
    dev.updatePower(gb, p, Vd); dev.updateTemperature(gb, p, dt)
    out = o.superstep(Vd)
    ok = (np.array_equal(sim.last_event_log, o.last_events["log"])
          and np.array_equal(gb.t["site_element"].cpu().numpy(), o.element) and np.array_equal(gb.t["site_charge"].cpu().numpy(), o.charge))
    worst["I"] = max(worst["I"], abs(dev.imacro / out["imacro"] - 1)); worst["T"] = max(worst["T"], abs(dev.T_bg - out["T_bg"]))
    worst["dt"] = max(worst["dt"], abs(dt / out["step_time"] - 1)); worst["margin"] = min(worst["margin"], float(o.last_events["margin"].min()))
    nev += len(sim.last_event_log)
    if not ok or worst["I"] > 1e-6 or worst["dt"] > 1e-6 or worst["T"] > 1e-7:
        print("DISAGREEMENT at step", k, worst, flush=True)
        sys.exit(1)
print("soak OK: %d supersteps, %d events, seed %d; worst rel dI %.1e, rel d(dt) %.1e, |dT| %.1e K, smallest bucket-edge margin %.1e"
      % (steps, nev, seed, worst["I"], worst["dt"], worst["T"], worst["margin"]))
