"""Diagnostic: one 7.5 nm superstep with profiling on; prints the SpMV layout statistics of the current solve."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from devicekmc_amd import host, lib
L = lib.load()
wl = sys.argv[1] if len(sys.argv) > 1 else "7.5nm"
s, p = bench.make_workload(wl)
dev = host.Device(s, p, gpu_neighbors="cuda:0"); sim = host.KMCProcess(dev, p.freq); gb = dev.make_gpubuf("cuda:0")
dev.setLaplacePotential(gb, p, 5.0); gb.sync_HostToGPU(dev)
L.dkmc_set_profiling(1)
for k in range(2):
    dev.updateCharge(gb); dev.updatePotential(gb, p, 5.0, k); _, dt = sim.executeKMCStep(gb, dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    dev.updatePower(gb, p, 5.0); torch.cuda.synchronize(); t1 = time.perf_counter()
    st = host.get_stats()
    keys = ["cg_iters_X", "X_nnz", "spmv_segments", "spmv_segment_entries", "spmv_tiles", "spmv_tile_entries", "spmv_long_rows", "spmv_short_rows",
            "spmv_short_nnz", "spmv_long_ms", "spmv_long_launches", "spmv_short_ms", "spmv_short_launches"]
    print("step", k, "current ms", round((t1 - t0) * 1e3, 2), {q: st[q] for q in keys})
    if st["spmv_long_launches"]:
        print("  seg/tile kernel us", st["spmv_long_ms"] / st["spmv_long_launches"] * 1e3, " row kernel us", st["spmv_short_ms"] / max(1, st["spmv_short_launches"]) * 1e3)
