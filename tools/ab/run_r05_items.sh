# round 5: same-box A/B of the nominal run length kc of the tile kernels' run list (dkmc_set_x_items; profiles/r05_ab_tile_run_lists.jsonl also holds
# the sub-block-balanced run list and the generic loop form measured from the working tree while they existed: tools/attic/xtb_apply_generic.hip)
# on the tile x panel kernel's time (dkmc_xtb_time_apply, width 16 and 8, product form), tile:10
mkdir -p gpurun_out/r05
python3 - <<'PY' 2>gpurun_out/r05/items_ab.err | tee -a gpurun_out/r05/items_ab.log
import ctypes as C, json, sys
sys.path.insert(0, ".")
import bench
from devicekmc_amd import lib
from devicekmc_amd.lib import check
L = lib.load()
for rep in (1, 2):
    for kc in (16, 32, 8, 64):
        L.dkmc_set_x_items(kc)
        sim = bench.Sim("tile:10", "cuda:0", cg_tol=1e-3)
        sim.L.dkmc_set_x_block(1)
        sim.step(False)
        out = {"kc": kc, "rep": rep}
        h = (C.c_longlong * 11)(); check(sim.L.dkmc_xt_tile_census(h)); out["runs"] = h[10]
        for form in (0,):
            sim.L.dkmc_set_x_apply_form(form)
            for w in (16, 8):
                us = C.c_double(0)
                check(sim.L.dkmc_xtb_time_apply(w, 0, 8, C.byref(us)))
                out["form%d_s%d_us" % (form, w)] = round(us.value, 1)
        sim.L.dkmc_set_x_apply_form(0)
        d, a = C.c_double(-1), C.c_double(-1)
        check(sim.L.dkmc_xtb_check_product(16, C.byref(d), C.byref(a)))
        out["product_check_rel"] = d.value / a.value
        print(json.dumps(out), flush=True)
        del sim
L.dkmc_set_x_items(0)
PY
