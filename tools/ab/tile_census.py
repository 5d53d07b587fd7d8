import ctypes as C, json, sys
sys.path.insert(0, ".")
import bench
from devicekmc_amd.lib import check
for wl in ("tile:10", "tile:5", "7.5nm"):
    sim = bench.Sim(wl, "cuda:0", cg_tol=1e-3)
    sim.step(False)
    h = (C.c_longlong * 11)()
    check(sim.L.dkmc_xt_tile_census(h))
    h = list(h)
    tiles = sum(h[:9]); subs = sum(c * h[c] for c in range(9))
    print(json.dumps({"workload": wl, "tiles_by_present_subblocks": h[:9], "tiles": tiles, "subblocks": subs, "subblocks_in_full_tiles": 8 * h[8],
                      "frac_subblocks_in_full_tiles": round(8 * h[8] / max(subs, 1), 4), "full_tiles_in_chains_ge2": h[9], "runs": h[10]}), flush=True)
    del sim
