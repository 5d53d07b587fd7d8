# auxiliary columns of the block-CG: hash set against the smooth set (dkmc_set_x_aux 0 / 2), same binary
mkdir -p gpurun_out/r04
cat > /tmp/xaux.py <<'PY'
import os, sys, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from devicekmc_amd import lib
L = lib.load()
out = {}
for name, steps in (("2.5nm", 10), ("7.5nm", 10), ("tile:2", 6), ("tile:5", 3)):
    for mode in (0, 2):
        L.dkmc_set_x_aux(mode)
        sim = bench.Sim(name, "cuda:0", x_format=1)
        el, n = sim.run(steps, 1, budget_s=120.0)
        st = sim.host.get_stats()
        r = bench.summary(sim, el, n)
        out["%s aux%d" % (name, mode)] = {"ms_per_step": r["ms_per_step"], "sweeps": r["per_step"]["cg_iters_X"], "xb_aux": st["xb_aux"], "fallback": st["xb_fallback"], "trace": sim.trace[:3]}
        print(name, "aux", mode, "->", st["xb_aux"], "ms/step %.2f" % r["ms_per_step"], "sweeps %.1f" % r["per_step"]["cg_iters_X"], "fallback", st["xb_fallback"], "I", [t[1] for t in sim.trace[:3]], flush=True)
        sim.close()
json.dump(out, open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out/r04/x1_aux.json"), "w"), indent=1)
PY
timeout -k 10 900 python /tmp/xaux.py 2>&1 | grep -v amdgpu.ids
