# round 5: what k_xtb_apply is made of, same box, same launch: dkmc_xtb_time_apply with the measurement variants (2 = no stream
# re-read, 4 = no LDS traffic, 7 = matrix instructions alone, 10 = the product form without the stream re-read) beside the product kernel and the round-4 form of its loop (dkmc_set_x_apply_form 0 / 1); the variants ride on the round-4 form
mkdir -p gpurun_out/r05      # (the library must have been built with DKMC_MEASURE_VARIANTS=1 python __graft_entry__.py before the gpurun call)
python3 - <<'PY' 2>gpurun_out/r05/apply_parts.err | tee -a gpurun_out/r05/apply_parts.log
import ctypes as C, json, sys
sys.path.insert(0, ".")
import bench
from devicekmc_amd.lib import check
sim = bench.Sim("tile:10", "cuda:0", cg_tol=1e-3)
sim.L.dkmc_set_x_block(1)
sim.step(False)
def t(w, v):
    us = C.c_double(0)
    check(sim.L.dkmc_xtb_time_apply(w, v, 8, C.byref(us)))
    return round(us.value, 1)
for rep in (1, 2):
    out = {"rep": rep}
    for form in (1, 0):
        sim.L.dkmc_set_x_apply_form(form)
        out["product_form_%d_us" % form] = t(16, 0)
    sim.L.dkmc_set_x_apply_form(0)
    for v in (2, 4, 7, 10, 12):
        out["variant_%d_us" % v] = t(16, v)
    print(json.dumps(out), flush=True)
PY
