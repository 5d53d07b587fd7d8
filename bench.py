#!/usr/bin/env python3
"""KMC supersteps/s of the DeviceKMC hot path on MI355X.

One "step" = one KMC superstep (kmc_main.cpp:175-279): charge update, background potential (K assembly +
Jacobi-CG) + screened-Coulomb pair sum, event table + residence-time event loop, current solve (X assembly
+ Jacobi-CG + I_macro + dissipated power) and the global temperature update, on a synthetic device whose
site fields are resident in HBM before the timed region.

Workloads (config.workload):
  7.5nm   the reference's own 85 071-site single device (structures/single_devices/timing_7.5nm; the
          configuration its only full-step timing log is quoted on), V = 5, rnd_seed = 5   [default]
  2.5nm   9 399 sites (configs[0], plumbing)
  tile:K  the 2.5 nm cell tiled K x K laterally, e.g. tile:3 = 84 591 sites

Contract: python bench.py --gpus N --steps K --warmup W ; prints ONE JSON line on rank 0.
N > 1 (launched with torch.distributed.run; the reference has no multi-GPU path): `value` is measured in replica mode --
every rank runs an independent replica of the workload (own KMC random stream) on its own GPU, value = aggregate
steps/s, "scaling": "weak".  The same run then measures the sharded current solve (csrc/comm.hip: all ranks advance ONE
simulation in lockstep, the segment stage of A*p is dealt to the ranks, one RCCL all-gather per CG iteration) on the
default workload and on a larger one, against the single-GPU time of the same steps, and checks bit-identity; that goes
into the extra "sharded_solve" block (strong scaling, never `value`).  A watchdog prints the line without that block if
the sharded part does not finish in time.  `--mode sharded` makes the sharded run the measured one instead (one simulation,
`value` = its steps/s, "scaling": "strong").
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak (MI355X_MICROARCH.md)


def make_workload(name):
    from devicekmc_amd import params, structure
    g = os.path.join(ROOT, "tests", "golden")
    if name == "7.5nm":
        s = structure.load_structure(os.path.join(g, "device_7.5nm.npz"))
        p = params.KMCParameters(rnd_seed=5, lattice=tuple(s.meta["lattice"]), num_atoms_first_layer=1296,
                                 num_atoms_contact=12960, A=76.725e-10 * 76.725e-10)
    elif name == "2.5nm":
        s = structure.load_structure(os.path.join(g, "device_2.5nm.npz"))
        p = params.KMCParameters()
    elif name.startswith("tile:"):
        k = int(name.split(":")[1])
        cell = structure.load_structure(os.path.join(g, "device_2.5nm.npz"))
        s = structure.tile_structure(cell, k, 25.575, 25.575, 1440)
        p = params.KMCParameters().for_tiling(k)
    else:
        raise SystemExit("unknown workload " + name)
    p.solve_heating_global = True
    return s, p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="7.5nm")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--warm-start", type=int, default=0, help="dkmc_set_current_warm_start mode (0 = reference)")
    ap.add_argument("--x-format", type=int, default=1, help="1: tiled X (default); 0: CSR X as the reference stores it")
    ap.add_argument("--mode", choices=["replicas", "sharded"], default="replicas",
                    help="N > 1: replicas (weak scaling, default) or ONE simulation with the sharded current solve (strong scaling)")
    ap.add_argument("--no-sharded", action="store_true", help="N > 1: skip the sharded-solve block")
    ap.add_argument("--sharded-workloads", default=None, help="comma list; default: the main workload and tile:5")
    ap.add_argument("--sharded-steps", type=int, default=2)
    ap.add_argument("--sharded-timeout", type=float, default=420.0, help="watchdog for the sharded-solve block [s]")
    args = ap.parse_args()

    import numpy as np
    import torch

    from devicekmc_amd import parallel
    # rehearsal on a one-GPU box: DKMC_BENCH_BACKEND=gloo DKMC_BENCH_SINGLE_DEVICE=1 lets several ranks share cuda:0
    backend = os.environ.get("DKMC_BENCH_BACKEND", "nccl")
    if os.environ.get("DKMC_BENCH_SINGLE_DEVICE"):
        os.environ["LOCAL_RANK_REAL"] = os.environ.get("LOCAL_RANK", "0")
        os.environ["LOCAL_RANK"] = "0"
    rank, world, local_rank = parallel.init(backend)
    torch.cuda.set_device(local_rank)
    devname = "cuda:%d" % local_rank

    from devicekmc_amd import host, lib
    L = lib.load()
    Vd = 5.0
    s, p = make_workload(args.workload)
    sharded_main = args.mode == "sharded" and world > 1
    if sharded_main:
        parallel.attach_solver_comm()          # every rank advances the same simulation (same seeds); the current solve is sharded
    else:
        p.rnd_seed_kmc = parallel.replica_kmc_seed(p.rnd_seed_kmc, rank)   # replicas follow different event streams
    dev = host.Device(s, p, gpu_neighbors=devname)       # HIP cell-list neighbour index (setup, outside the timed region)
    sim = host.KMCProcess(dev, p.freq)
    gb = dev.make_gpubuf(devname)
    L.dkmc_set_current_warm_start(args.warm_start)
    L.dkmc_set_x_format(args.x_format)
    dev.setLaplacePotential(gb, p, Vd)
    gb.sync_HostToGPU(dev)

    phases = {"charge": 0.0, "potential": 0.0, "rates": 0.0, "current": 0.0, "heat": 0.0}
    counters = {"events": 0, "cg_iters_K": 0, "cg_iters_X": 0}
    prof = {"long_ms": 0.0, "long_n": 0, "short_ms": 0.0, "short_n": 0}

    def sync():
        torch.cuda.synchronize()

    def step(k, timed):
        t0 = time.perf_counter()
        dev.updateCharge(gb)
        if timed: sync()
        t1 = time.perf_counter()
        dev.updatePotential(gb, p, Vd, k)
        if timed: sync()
        t2 = time.perf_counter()
        _, dt = sim.executeKMCStep(gb, dev)
        t3 = time.perf_counter()
        dev.updatePower(gb, p, Vd)
        t4 = time.perf_counter()
        dev.updateTemperature(gb, p, dt)
        sync()
        t5 = time.perf_counter()
        if timed:
            st = host.get_stats()
            for key, v in zip(phases, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
                phases[key] += v
            counters["events"] += sim.last_n_events
            counters["cg_iters_K"] += st["cg_iters_K"]
            counters["cg_iters_X"] += st["cg_iters_X"]
            prof["long_ms"] += st["spmv_long_ms"]; prof["long_n"] += st["spmv_long_launches"]
            prof["short_ms"] += st["spmv_short_ms"]; prof["short_n"] += st["spmv_short_launches"]
        return dt

    for k in range(args.warmup):
        step(k, False)
    L.dkmc_set_profiling(1)
    sync()
    if world > 1:
        parallel.barrier()
    sync()
    t_start = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k, True)
    sync()
    if world > 1:
        parallel.barrier()
    sync()
    elapsed = time.perf_counter() - t_start
    L.dkmc_set_profiling(0)
    elapsed = parallel.max_over_ranks(elapsed, devname if backend == "nccl" else "cpu")

    # ---- same workload with the optional unscaled warm start of the current solve (dkmc_set_current_warm_start(1)):
    #      reported next to the reference-faithful number, never as `value` ----
    alt = None
    if world == 1 and args.warm_start == 0:
        L.dkmc_set_current_warm_start(1)
        for k in range(2):
            step(args.warmup + args.steps + k, False)
        sync()
        t0 = time.perf_counter(); it0 = 0
        for k in range(args.steps):
            step(args.warmup + args.steps + 2 + k, False); it0 += host.get_stats()["cg_iters_X"]
        sync()
        ta = time.perf_counter() - t0
        alt = {"current_warm_start": 1, "value": round(args.steps / ta, 4), "ms_per_step": round(ta / args.steps * 1e3, 3),
               "cg_iters_X": it0 / args.steps}
        L.dkmc_set_current_warm_start(0)

    st = host.get_stats()
    # ---- roofline of the dominant kernel: k_spmv_ap (CSR SpMV t = X p of the current solve's CG, fused p.t) ----
    roof = None
    if prof["long_n"] > 0:
        avg_ms = prof["long_ms"] / prof["long_n"]
        # algorithmic bytes per launch: 12 B per stored non-zero (value + column) + per row 8 B of row pointers,
        # 8 B result written, 8 B of p read for the fused dot; one launch covers every row of X
        nnz_all = st["spmv_long_nnz"] + st["spmv_short_nnz"]
        rows_all = st["spmv_long_rows"] + st["spmv_short_rows"]
        if st["xt_subblocks"] > 0:
            # tiled X (default): the dominant kernel is k_xt_apply -- one wave per run of tiles of the tunnelling block (8 KiB per stored
            # 32 x 32 sub-block, read once for both triangles; 16 B descriptor, 32 row sums written per tile; 256 column sums per
            # run) plus, in the same launch, the neighbour part Xs in CSR form (12 B per non-zero, 8 B row pointer, 8 B result,
            # 8 B scale and 4 B class per row)
            kname = "k_xt_apply"
            bytes_per_launch = (8192.0 * st["xt_local_subblocks"] + (16.0 + 256.0) * st["spmv_tiles"] + (16.0 + 2048.0) * st["xt_items"]
                                + 12.0 * st["xt_sparse_nnz"] + 28.0 * rows_all)
            csr_equiv = 12.0 * nnz_all + 24.0 * rows_all
        elif st["spmv_segments"] > 0:
            # dense-run mode: the dominant kernel is k_spmv_segs (one wave per <= 2048-entry segment of a tunnelling row, and
            # one wave per symmetric tile); its layout moves 8 B per entry read (value only; the direction vector is
            # compacted over S and stays in L2), 16 B per segment descriptor and 8 B per segment result.  The CSR
            # formulation of the same product would move 12 B per stored entry.
            kname = "k_spmv_segs"
            # the same launch also carries the short rows of X in CSR form (12 B per non-zero + 24 B per row)
            bytes_per_launch = (8.0 * st["spmv_segment_entries"] + 24.0 * st["spmv_segments"]
                                + 12.0 * st["spmv_short_nnz"] + 24.0 * st["spmv_short_rows"])
            # symmetric tiles (same launch): each tile is 32 x 256 doubles of tile-major storage (zero where X has no entry), read
            # once for both triangles, + its 8 B descriptor and the partial sums it writes (32 row sums, 256 column sums)
            bytes_per_launch += st["spmv_tiles"] * (32 * 256 * 8.0 + 8.0 + 32 * 8.0 + 256 * 8.0)
            csr_equiv = 12.0 * nnz_all + 24.0 * rows_all
        else:
            kname = "k_spmv_ap"
            bytes_per_launch = 12.0 * nnz_all + 24.0 * rows_all
            csr_equiv = bytes_per_launch
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_r01.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(args.workload, {}).get(kname + "_bytes_per_launch")
            except Exception:
                traffic = None
        roof = {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "avg_launch_us": round(avg_ms * 1e3, 2), "launches": prof["long_n"],
                "algorithmic_bytes_per_launch": bytes_per_launch,
                "csr_equivalent_GBps": round(csr_equiv / (avg_ms * 1e-3) / 1e9, 1),
                "symmetric_tiles": int(st["spmv_tiles"]), "tile_entries": int(st["spmv_tile_entries"]),
                "segment_entries": int(st["spmv_segment_entries"]), "subblocks": int(st["xt_subblocks"]), "tile_runs": int(st["xt_items"]),
                "row_kernel_us": round(prof["short_ms"] / max(prof["short_n"], 1) * 1e3, 2)}

    # ---- CPU baseline: the oracle (own OpenMP port of the same step) on this box's host cores, rank 0, N=1 only ----
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        ncores = min(16, os.cpu_count() or 1)         # the box's CPU share for one GPU; more threads only add contention
        os.environ["OMP_NUM_THREADS"] = str(ncores)   # read by libgomp when the oracle library is loaded
        from oracle import oracle as oc
        o = oc.OracleKMC(s.element, s.x, s.y, s.z, p)
        o.set_laplace_potential(Vd)
        o.superstep(Vd)                              # untimed: cold-start CG of the first step
        t0 = time.perf_counter()
        nsamp = 1 if s.N > 30000 else 5
        for _ in range(nsamp):
            o.superstep(Vd)
        tc = (time.perf_counter() - t0) / nsamp
        cpu = {"value": round(1.0 / tc, 5), "unit": "KMC steps/s", "cores": ncores, "kind": "port",
               "sample": "%d superstep(s) of the same workload after one untimed step (oracle/kmc_oracle.c, OpenMP)" % nsamp,
               "ms_per_step": round(tc * 1e3, 1),
               "split_ms": {k: round(v * 1e3, 2) for k, v in o.timing.items()}}

    out = None
    if rank == 0:
        n = args.steps
        out = {
            "metric": "KMC steps/sec", "value": round(n / elapsed if sharded_main else parallel.aggregate_rate(n, world, elapsed), 4),
            "unit": "KMC steps/s",
            "n_gpus": world, "steps": n, "warmup": args.warmup, "ms_per_step": round(elapsed / n * 1e3, 3),
            "higher_is_better": True, "scaling": "strong" if sharded_main else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": args.workload, "sites": int(s.N), "nn": int(dev.max_num_neighbors), "atoms": int(dev.N_atom),
                       "Vd": Vd, "phases": "charge+potential+rates+current+heat", "parallelism": ("sharded current solve x%d" if sharded_main else "replicas x%d") % world,
                       "current_warm_start": args.warm_start, "cg_tol": p.cg_tol},
            "split_ms": {k: round(v / n * 1e3, 3) for k, v in phases.items()},
            "per_step": {"events": counters["events"] / n, "cg_iters_K": counters["cg_iters_K"] / n,
                         "cg_iters_X": counters["cg_iters_X"] / n, "X_nnz": int(st["X_nnz"]), "n_charged": int(st["n_charged"]),
                         "K_rows": int(s.N - 2 * p.num_atoms_first_layer), "K_nnz": int(gb.c.Device_nnz)},
            "roofline": roof, "cpu_baseline": cpu, "alt_warm_start": alt,
        }

    # ---- the one JSON line; printed exactly once, by the normal path or by the watchdog of the sharded block ----
    import threading
    emit_lock = threading.Lock()
    emitted = []

    def emit(extra):
        with emit_lock:
            if emitted:
                return
            emitted.append(1)
            if rank == 0:
                if extra is not None:
                    out["sharded_solve"] = extra
                print(json.dumps(out), flush=True)

    if sharded_main:
        parallel.detach_solver_comm()
    if world > 1 and not args.no_sharded and not sharded_main:
        def on_timeout():
            emit({"error": "sharded-solve block did not finish within %.0f s" % args.sharded_timeout})
            os._exit(0)             # a rank stuck in a collective cannot be unwound
        wd = threading.Timer(args.sharded_timeout, on_timeout); wd.daemon = True; wd.start()
        try:
            del gb, sim, dev
            torch.cuda.empty_cache()
            names = args.sharded_workloads.split(",") if args.sharded_workloads else [args.workload, "tile:5"]
            extra = sharded_block(names, args.sharded_steps, devname, backend, rank, world)
        except Exception as exc:         # the replica measurement above stays valid
            extra = {"error": repr(exc)[:300]}
        wd.cancel()
        emit(extra)
    else:
        emit(None)
    parallel.finalize()


def lockstep_run(name, nsteps, devname, Vd=5.0, tiles=1):
    """nsteps supersteps of workload `name` from a fresh state with the reference seeds (identical on every rank);
    returns (seconds for the steps after the first, trace, stats of the last step).  The first step is untimed: it fills the
    tunnelling-coefficient cache and sizes the scratch buffers."""
    import torch
    from devicekmc_amd import host, lib
    L = lib.load()
    s, p = make_workload(name)
    dev = host.Device(s, p, gpu_neighbors=devname)
    sim = host.KMCProcess(dev, p.freq)
    gb = dev.make_gpubuf(devname)
    L.dkmc_set_current_warm_start(0)
    L.dkmc_set_symmetric_tiles(tiles)
    dev.setLaplacePotential(gb, p, Vd)
    gb.sync_HostToGPU(dev)
    trace, iters = [], 0
    t0 = None
    L.dkmc_set_profiling(1)
    for k in range(nsteps + 1):
        if k == 1:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        dev.updateCharge(gb); dev.updatePotential(gb, p, Vd, k)
        _, dt = sim.executeKMCStep(gb, dev)
        dev.updatePower(gb, p, Vd); dev.updateTemperature(gb, p, dt)
        trace.append((dt, dev.imacro, dev.T_bg))
        if k >= 1:
            iters += host.get_stats()["cg_iters_X"]
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    L.dkmc_set_profiling(0)
    L.dkmc_set_symmetric_tiles(1)
    st = dict(host.get_stats())
    st["cg_iters_X_per_step"] = iters / nsteps
    st["sites"] = int(s.N)
    del gb, sim, dev
    torch.cuda.empty_cache()
    return el, trace, st


def sharded_block(names, nsteps, devname, backend, rank, world):
    """Strong scaling of one simulation over the ranks: single-GPU time (every rank runs it, slowest counts) against the
    lockstep run with the sharded current solve, same steps, same seeds; bit-identity of (dt, I_macro, T_bg) checked."""
    import torch
    import torch.distributed as dist
    from devicekmc_amd import parallel
    red_dev = devname if backend == "nccl" else "cpu"
    res = {"ranks": world, "steps": nsteps, "workloads": {}}
    for name in names:
        parallel.barrier()
        t_single, trace_single, st1 = lockstep_run(name, nsteps, devname)
        t_single = parallel.max_over_ranks(t_single, red_dev)
        parallel.barrier()
        res["transport"] = parallel.attach_solver_comm()
        try:
            parallel.barrier()
            t_shard, trace_shard, st2 = lockstep_run(name, nsteps, devname)
        finally:
            parallel.detach_solver_comm()
        t_shard = parallel.max_over_ranks(t_shard, red_dev)
        # default arithmetic on both sides (symmetric tiles where they apply): the sharded solve then completes the row sums with an
        # all-reduce and equals the single-GPU run to rounding; all ranks must hold the same bits.  (The runs-only variant,
        # dkmc_set_symmetric_tiles(0), is bit-identical to the single-GPU run: tests/test_dist_sharded.py.)
        same_here = 1.0 if trace_shard == trace_single else 0.0
        close = all(abs(a - b) <= 1e-6 * abs(b) for ta, tb in zip(trace_shard, trace_single) for a, b in zip(ta, tb))
        flag = torch.tensor([same_here], dtype=torch.float64, device=red_dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        traces = [None] * world
        dist.all_gather_object(traces, trace_shard)
        res["workloads"][name] = {
            "sites": st1["sites"], "X_nnz": int(st1["X_nnz"]), "cg_iters_X": st1["cg_iters_X_per_step"],
            "single_gpu_ms_per_step": round(t_single / nsteps * 1e3, 3), "sharded_ms_per_step": round(t_shard / nsteps * 1e3, 3),
            "speedup": round(t_single / t_shard, 3),
            "bit_identical_to_single_gpu": bool(flag.item() == 1.0), "equal_to_single_gpu_within_1e-6": bool(close),
            "symmetric_tiles": int(st2["spmv_tiles"]), "collective": "all-reduce" if st2["spmv_tiles"] > 0 else "all-gather",
            "ranks_agree": all(t == traces[0] for t in traces),
            "segments": int(st2["spmv_segments"]), "exchanged_doubles_per_rank": int(st2["comm_count_per_rank"]),
            "exchange_us": round(st2["comm_ms"] / max(st2["comm_launches"], 1) * 1e3, 2),
            "segment_kernel_us": round(st2["spmv_long_ms"] / max(st2["spmv_long_launches"], 1) * 1e3, 2),
            "single_gpu_segment_kernel_us": round(st1["spmv_long_ms"] / max(st1["spmv_long_launches"], 1) * 1e3, 2),
        }
    return res


if __name__ == "__main__":
    main()
