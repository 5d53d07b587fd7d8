# round 5: same-box A/B of two builds of k_xtb_apply (tools/ab/lib_a.so = before, lib_b.so = after): kernel time via dkmc_xtb_time_apply, then the product check
mkdir -p gpurun_out/r05
cp devicekmc_amd/libdevicekmc_hip.so tools/ab/orig.so
for rep in 1 2; do for v in lib_a.so lib_b.so; do
  cp tools/ab/$v devicekmc_amd/libdevicekmc_hip.so
  python3 - "$v" "$rep" <<'PY' 2>/dev/null | tee -a gpurun_out/r05/apply_ab.log
import ctypes as C, json, sys
sys.path.insert(0, ".")
import bench
from devicekmc_amd.lib import check
sim = bench.Sim("tile:10", "cuda:0", cg_tol=1e-3)
sim.L.dkmc_set_x_block(1)
sim.step(False)
out = {"lib": sys.argv[1], "rep": int(sys.argv[2])}
for w in (16, 8):
    us = C.c_double(0)
    check(sim.L.dkmc_xtb_time_apply(w, 0, 8, C.byref(us)))
    out["s%d_us" % w] = round(us.value, 1)
d, a = C.c_double(-1), C.c_double(-1)
check(sim.L.dkmc_xtb_check_product(16, C.byref(d), C.byref(a)))
out["product_check_rel"] = d.value / a.value
print(json.dumps(out))
PY
done; done
cp tools/ab/orig.so devicekmc_amd/libdevicekmc_hip.so
