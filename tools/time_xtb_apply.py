#!/usr/bin/env python3
"""Where the time of the block-CG's tile x panel kernel goes: the kernel as it runs, without its matrix instructions, and without
re-reading the tile stream, on one workload (dkmc_xtb_time_apply).  usage: python tools/time_xtb_apply.py tile:10 [width ...]
(the measurement variants exist only in a library built with DKMC_MEASURE_VARIANTS=1 python __graft_entry__.py)"""
import ctypes as C
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from devicekmc_amd.lib import check  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "tile:5"
widths = [int(x) for x in sys.argv[2:]] or [16, 8]
sim = bench.Sim(name, "cuda:0", cg_tol=1e-3)
sim.L.dkmc_set_x_block(1)
sim.step(False)
st = sim.host.get_stats()
out = {"workload": name, "sites": int(sim.s.N), "subblocks": int(st["xt_subblocks"]), "tile_bytes": 8192 * int(st["xt_subblocks"])}
for w in widths:
    for variant, what in ((0, "kernel"), (1, "no_matrix_instructions"), (2, "no_tile_stream"), (3, "kernel_one_kpair_stages"), (4, "no_lds_traffic")):
        if variant >= 3 and w <= 12:
            continue
        us = C.c_double(0)
        check(sim.L.dkmc_xtb_time_apply(w, variant, 5, C.byref(us)))
        out["s%d_%s_us" % (w, what)] = round(us.value, 1)
    out["s%d_kernel_GBps" % w] = round(out["tile_bytes"] / out["s%d_kernel_us" % w] / 1e3, 1)
print(json.dumps(out))
