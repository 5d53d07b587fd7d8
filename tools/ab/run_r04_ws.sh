mkdir -p gpurun_out/r04
python - <<'PY' > gpurun_out/r04/ws_check.txt 2>&1
import ctypes as C, sys, os
sys.path.insert(0, os.getcwd())
import bench
from devicekmc_amd.lib import check
sim = bench.Sim("7.5nm", "cuda:0", cg_tol=1e-3, x_block=1)
sim.step(False)
for ws in (1, 2):
    sim.L.dkmc_debug_xtb_waves_per_run(ws)
    for w in (16, 8):
        d, a = C.c_double(-1), C.c_double(-1)
        check(sim.L.dkmc_xtb_check_product(w, C.byref(d), C.byref(a)))
        print("ws", ws, "width", w, "max diff", d.value, "of", a.value, "rel", d.value / a.value)
PY
cat gpurun_out/r04/ws_check.txt | tail -5
B="--no-cpu-baseline --scale-points none --no-alt --no-pmc --no-cpp-host --no-log-tolerance --no-device --no-reference-order"
for ws in 1 2; do
timeout -k 10 300 python bench.py --workload tile:10 --steps 2 --warmup 1 --x-waves $ws $B > gpurun_out/r04/b7_t10_ws$ws.json 2> gpurun_out/r04/b7_t10_ws$ws.err; python -c "
import json;d=json.load(open('gpurun_out/r04/b7_t10_ws$ws.json'));print('tile:10 ws$ws',d['ms_per_step'],d['per_step']['cg_iters_X'],d['roofline']['avg_launch_us'],d['roofline']['row_kernel_us'])"
timeout -k 10 300 python bench.py --workload 7.5nm --steps 10 --warmup 2 --x-waves $ws $B > gpurun_out/r04/b7_7p5_ws$ws.json 2> gpurun_out/r04/b7_7p5_ws$ws.err; python -c "
import json;d=json.load(open('gpurun_out/r04/b7_7p5_ws$ws.json'));print('7.5nm ws$ws',d['ms_per_step'],d['per_step']['cg_iters_X'],d['roofline']['avg_launch_us'],d['roofline']['row_kernel_us'])"
done
timeout -k 10 300 python bench.py --workload tile:10 --steps 1 --warmup 0 --x-waves 2 --cg-tol 1e-3 --no-cpu-baseline --scale-points none --no-alt --no-cpp-host --no-log-tolerance --no-device --no-reference-order > gpurun_out/r04/b7_t10_ws2_pmc.json 2> gpurun_out/r04/b7_t10_ws2_pmc.err; python -c "
import json;d=json.load(open('gpurun_out/r04/b7_t10_ws2_pmc.json'));r=d['roofline'];print('pmc ws2', r.get('traffic'), r.get('algorithmic_bytes_per_launch'), r.get('traffic_over_algorithmic'))"
