"""Round 5: the split polynomial preconditioner of the block-CG (dkmc_set_x_poly(d)) against the plain block loop: same simulation (same seed, same events),
block-CG sweeps and seconds per superstep, the trace (dt, I_macro) of both."""
import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
wls = sys.argv[1:] or ["7.5nm", "tile:5"]
for wl in wls:
    nsteps = 12 if wl != "tile:10" else 8
    res = {}
    degs = [int(x) for x in os.environ.get("DEGREES", "1,2,4").split(",")]
    for d in [0] + degs:
        sim = bench.Sim(wl, "cuda:0")
        sim.L.dkmc_set_x_poly(d)
        for k in range(nsteps):
            sim.step(True)
        res[d] = {"sweeps": [n for _, n in sim.step_log], "seconds": [round(t, 4) for t, _ in sim.step_log], "trace": [(float(a), float(b)) for a, b, _ in sim.trace]}
        sim.L.dkmc_set_x_poly(0)
        del sim
    for d in degs:
        dev = max(abs(a[1] - b[1]) / max(abs(a[1]), 1e-300) for a, b in zip(res[0]["trace"], res[d]["trace"]))
        same_dt = all(a[0] == b[0] for a, b in zip(res[0]["trace"], res[d]["trace"]))
        print(json.dumps({"workload": wl, "degree": d, "sweeps_plain": res[0]["sweeps"], "sweeps": res[d]["sweeps"], "s_plain": round(sum(res[0]["seconds"][2:]), 4),
                          "s": round(sum(res[d]["seconds"][2:]), 4), "same_dt_sequence": same_dt, "max_rel_dev_I_macro": dev}), flush=True)
