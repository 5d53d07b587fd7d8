"""round 5 (tools/ab/run_r05_apply_clock.sh): launches k_xtb_apply's product kernel and its measurement variants at tile:10 so that a rocprofv3
--pmc GRBM_GUI_ACTIVE pass can turn busy cycles / duration into the shader clock each of them actually runs at"""
import ctypes as C, sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from devicekmc_amd.lib import check
sim = bench.Sim("tile:10", "cuda:0", cg_tol=1e-3)
sim.L.dkmc_set_x_block(1)
sim.step(False)
us = C.c_double(0)
for form in (1, 0):
    sim.L.dkmc_set_x_apply_form(form)
    check(sim.L.dkmc_xtb_time_apply(16, 0, 4, C.byref(us)))
    check(sim.L.dkmc_xtb_time_apply(8, 0, 4, C.byref(us)))
sim.L.dkmc_set_x_apply_form(0)
for v in (2, 4, 7):
    check(sim.L.dkmc_xtb_time_apply(16, v, 4, C.byref(us)))
