import ctypes as C, json, sys
sys.path.insert(0, ".")
import bench
from devicekmc_amd.lib import check
for wl in ("tile:10", "tile:5"):
    sim = bench.Sim(wl, "cuda:0", cg_tol=1e-3)
    sim.L.dkmc_set_x_block(1)
    sim.step(False)
    for form in (0, 1):
        sim.L.dkmc_set_x_apply_form(form)
        for w in (16, 12, 8, 4, 2):
            d, a = C.c_double(-1), C.c_double(-1)
            check(sim.L.dkmc_xtb_check_product(w, C.byref(d), C.byref(a)))
            print(wl, "form", form, "width", w, "rel", d.value / a.value, flush=True)
            assert d.value <= 1e-12 * a.value
    sim.L.dkmc_set_x_apply_form(0)
    del sim
print("all ok")
