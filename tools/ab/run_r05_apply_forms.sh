# round 5: same-box A/B of the tile x panel kernel's product form (dkmc_set_x_apply_form(0)) against the round-4 form of its loop (1) in ONE library:
# kernel time via dkmc_xtb_time_apply (width 16 and 8), then the product check of each against the single-vector kernel.
# (profiles/r05_ab_xtb_apply_forms.jsonl also holds two intermediate forms and k_xtb_apply3 -- tools/attic/xtb_apply3.hip --, measured with
# this script from the working tree while they existed.)
mkdir -p gpurun_out/r05
python3 - <<'PY' 2>gpurun_out/r05/apply_forms.err | tee -a gpurun_out/r05/apply_forms.log
import ctypes as C, json, sys
sys.path.insert(0, ".")
import bench
from devicekmc_amd.lib import check
sim = bench.Sim("tile:10", "cuda:0", cg_tol=1e-3)
sim.L.dkmc_set_x_block(1)
sim.step(False)
for rep in (1, 2, 3):
    for form in (1, 0):
        sim.L.dkmc_set_x_apply_form(form)
        out = {"x_apply_form": form, "rep": rep}
        for w in (16, 8):
            us = C.c_double(0)
            check(sim.L.dkmc_xtb_time_apply(w, 0, 8, C.byref(us)))
            out["s%d_us" % w] = round(us.value, 1)
        d, a = C.c_double(-1), C.c_double(-1)
        check(sim.L.dkmc_xtb_check_product(16, C.byref(d), C.byref(a)))
        out["product_check_rel"] = d.value / a.value
        print(json.dumps(out), flush=True)
PY
