#!/usr/bin/env python3
"""Same-box A/B (round 5): the hash auxiliary columns of the block-CG started from the previous solve's solutions (dkmc_set_x_aux_warm(1), with
half of the auxiliary set smooth / zero-started) against every auxiliary column from zero (0): sweeps and ms per superstep.
usage: python tools/ab_aux_warm.py [workload ...] (default 7.5nm tile:5 tile:10)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

if __name__ == "__main__":
    names = sys.argv[1:] or ["7.5nm", "tile:5", "tile:10"]
    out = {}
    for name in names:
        nst = 12 if name == "7.5nm" else 8
        for mode in (0, 1):
            from devicekmc_amd import lib
            lib.load().dkmc_set_x_aux_warm(mode)
            lib.load().dkmc_set_x_aux(int(os.environ.get("AUXMODE", "2")))          # 2 = default set; 0 = all hash (then every auxiliary column is warm-started)
            sim = bench.Sim(name, "cuda:0")
            el, n = sim.run(nst, 1)
            st = sim.host.get_stats()
            out["%s aux_warm=%d" % (name, mode)] = {"ms_per_step": round(el / n * 1e3, 2), "sweeps": [i for _, i in sim.step_log], "xb_aux": int(st["xb_aux"]), "fallback": int(st["xb_fallback"]),
                                                    "trace_I": [t[1] for t in sim.trace[-3:]]}
            print(name, mode, out["%s aux_warm=%d" % (name, mode)], flush=True)
            sim.close()
    lib.load().dkmc_set_x_aux_warm(1)
    print(json.dumps(out))
