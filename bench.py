#!/usr/bin/env python3
"""KMC supersteps/s of the DeviceKMC hot path on MI355X.

One "step" = one KMC superstep (kmc_main.cpp:175-279): charge update, background potential (K assembly + Jacobi-CG) +
screened-Coulomb pair sum, event table + residence-time event loop, current solve (X assembly + Jacobi-CG + I_macro +
dissipated power) and the global temperature update, on a synthetic device whose site fields are resident in HBM before
the timed region.

Workloads (config.workload):
  tile:K  the 2.5 nm cell tiled K x K laterally: tile:3 = 84 591, tile:5 = 234 975, tile:10 = 939 900 sites
          (configs[2], the "~1e6" stack: the configuration north_star's target is stated on)      [default at every N]
          (tile:17 = 2 716 311 sites, 3.1e10 matrix entries, is the largest run so far on one GPU)
  7.5nm   the reference's own 85 071-site single device (structures/single_devices/timing_7.5nm; the configuration its only
          full-step timing log is quoted on = configs[1]), V = 5, rnd_seed = 5
  2.5nm   9 399 sites (configs[0], plumbing)
  crossbar_10nm_5pitch   the reference's 110 813-site crossbar (structures/crossbars/timing_10nm_5pitch), V = 1, current solve off

Contract: python bench.py --gpus N --steps K --warmup W ; prints ONE JSON line on rank 0.  `--gpus N` with N > 1 and no rank environment
starts the N ranks itself (devicekmc_amd/launch.py); a world that is not --gpus is refused.

N = 1: `value` = steady-state steps/s of tile:10 at the library's defaults: CG tolerance 1e-6 (the snapshot's), the block-CG of width 16
on the tiled X (csrc/xtb.hip), current solve started from the previous step's solution (dkmc_set_current_warm_start(1), the default since
round 5: tests/test_gpu_warm_start.py).  W >= 1 puts the cold step (coefficient cache, buffer sizing, zero start vector) into the warm-up; it is
reported as `cold_step`.  The same line carries
  * `roofline`: the dominant kernel (k_xtb_apply: tile x panel product on the matrix cores) -- HIP-event time of sampled launches,
    algorithmic bytes AND fp64 flops per launch, HBM `traffic` from two rocprofv3 --pmc child runs;
  * `reference_order_cg`: the same workload with dkmc_set_x_block(1), the reference's single-vector CG (its iterate sequence) from the reference's start vector: cold + one
    steady step, sweeps per step, its own kernel roofline, and the strong-scaling model of the sharded solve built on it;
  * `reference_start_vector`: the same simulation with dkmc_set_current_warm_start(0), the reference code's start vector (the buffer it
    scaled by G0 in place, i.e. practically zero): what the default's warm start from the previous solution saves;
  * `cpu_baseline`: the oracle's CG iteration timed at this size x the REFERENCE algorithm's iteration count (a lower bound on a CPU step);
  * `device_7p5nm`: the reference's own 85 071-site device: steps/s, split, kernel roofline, oracle superstep on the host cores, the run
    under KMCParameters.log_revision() checked against the reference's CUDA log (`at_log_tolerance`), the C++ host cross-check;
  * `scale_points`: tile:5; the reference's crossbar under log_revision() against ITS log (KMC time of the timed steps, per-phase split next to
    the log's medians); a crossbar-SIZED stack (tile:20, 3.8e6 sites) with the current solve off, as every shipped crossbar runs.

N > 1 (the reference has no multi-GPU path): STRONG scaling of ONE simulation of tile:10: every rank advances the same simulation in
lockstep, the tunnelling block of X is generated, stored and streamed in per-rank shares (csrc/xt.hip, csrc/xtb.hip, csrc/comm.hip);
`value` = steps/s of that simulation, "scaling": "strong".  A `replicas` block (independent replicas of the 85 k-site device, aggregate
steps/s, weak scaling) rides along.  A watchdog ends the run with a non-zero exit code if a rank hangs.
If the requested K steps of a seconds-per-step workload would not fit the time budget (--budget, default 420 s), fewer steps are
timed and `steps` says how many (`steps_requested` keeps K).
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak (MI355X_MICROARCH.md)
FP64_VECTOR_PEAK_TFLOPS = 78.6  # MI355X fp64 vector peak (half the 157.3 TF f32 vector rate of MI355X_MICROARCH.md)
FP64_MATRIX_PEAK_TFLOPS = 78.6  # fp64 matrix peak = the vector rate on MI355X; v_mfma_f64_4x4x4_4b_f64 measured at 68-69 TFLOP/s sustained (tools/bench_mfma_f64.hip)
VD = 5.0
X_BLOCK = 16                   # dkmc_set_x_block (set from --x-block; 16 = the library default)


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
    elif name == "crossbar_10nm_5pitch":
        s = structure.load_structure(os.path.join(g, "crossbar_10nm_5pitch.npz"))
        p = params.KMCParameters(rnd_seed=5, lattice=tuple(s.meta["lattice"]), num_atoms_first_layer=144, num_atoms_contact=11520)
        p.solve_current = False
        return s, p                          # (its parameters.txt: V = 1, heating off, current off)
    else:
        raise SystemExit("unknown workload " + name)
    p.solve_heating_global = True
    return s, p


class Sim:
    """One simulation resident on the GPU: the reference's host objects (Device / KMCProcess / GPUBuffers mirror) + one superstep."""

    def __init__(self, name, devname, kmc_seed=None, x_format=1, warm_start=None, log_revision=False, cg_tol=None, x_block=None, Vd=None, solve_current=None):
        from devicekmc_amd import host, lib
        self.host, self.L = host, lib.load()
        self.name = name
        t0 = time.perf_counter()
        self.s, self.p = make_workload(name)
        self.Vd = VD if Vd is None else Vd
        if solve_current is not None:
            self.p.solve_current = solve_current
            if not solve_current:
                self.p.solve_heating_global = False
        self.x_block = X_BLOCK if x_block is None else x_block
        if log_revision:
            self.p = self.p.log_revision()           # cg_tol 1e-12 + CB edge on atoms: the settings that reproduce the reference's own log
        if cg_tol is not None:
            self.p.cg_tol = cg_tol
        if kmc_seed is not None:
            self.p.rnd_seed_kmc = kmc_seed
        self.dev = host.Device(self.s, self.p, gpu_neighbors=devname)   # HIP cell-list neighbour index (setup, outside the timed region)
        self.kmc = host.KMCProcess(self.dev, self.p.freq)
        self.gb = self.dev.make_gpubuf(devname)
        self.warm_start = 1 if warm_start is None else int(warm_start)        # 1 = the library default (start from the previous solution)
        self.L.dkmc_set_current_warm_start(self.warm_start)
        self.L.dkmc_set_x_format(x_format)
        self.L.dkmc_set_x_block(self.x_block)
        self.dev.setLaplacePotential(self.gb, self.p, self.Vd)
        self.gb.sync_HostToGPU(self.dev)
        self.setup_s = time.perf_counter() - t0
        self.k = 0
        self.cold = None            # (seconds, CG iterations on X) of the very first step, when run() makes it as a warm-up step
        self.reset_counters()

    def reset_counters(self):
        self.phases = {"charge": 0.0, "potential": 0.0, "rates": 0.0, "current": 0.0, "heat": 0.0}
        self.cnt = {"events": 0, "cg_iters_K": 0, "cg_iters_X": 0, "steps": 0}
        self.prof = {"long_ms": 0.0, "long_n": 0, "short_ms": 0.0, "short_n": 0, "comm_ms": 0.0, "comm_n": 0,
                     "kcg_ms": 0.0, "kcg_iters": 0, "pair_ms": 0.0, "pair_n": 0}
        self.trace = []
        self.step_log = []          # (seconds, CG iterations on X) of every timed step

    def step(self, timed):
        import torch
        dev, gb, p = self.dev, self.gb, self.p
        sync = torch.cuda.synchronize
        t0 = time.perf_counter()
        dev.updateCharge(gb)
        if timed: sync()
        t1 = time.perf_counter()
        self.L.dkmc_set_x_block(self.x_block)
        self.L.dkmc_set_current_warm_start(self.warm_start)
        dev.updatePotential(gb, p, self.Vd, self.k)
        if timed: sync()
        t2 = time.perf_counter()
        _, dt = self.kmc.executeKMCStep(gb, dev)
        t3 = time.perf_counter()
        if p.solve_current:
            dev.updatePower(gb, p, self.Vd)
        t4 = time.perf_counter()
        if p.solve_current:
            dev.updateTemperature(gb, p, dt)
        sync()
        t5 = time.perf_counter()
        self.k += 1
        self.trace.append((dt, dev.imacro, dev.T_bg))
        if timed:
            st = self.host.get_stats()
            for key, v in zip(self.phases, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
                self.phases[key] += v
            self.cnt["events"] += self.kmc.last_n_events
            self.cnt["cg_iters_K"] += st["cg_iters_K"]; self.cnt["cg_iters_X"] += st["cg_iters_X"]; self.cnt["steps"] += 1
            self.step_log.append((t5 - t0, int(st["cg_iters_X"])))
            pr = self.prof
            pr["long_ms"] += st["spmv_long_ms"]; pr["long_n"] += st["spmv_long_launches"]
            pr["short_ms"] += st["spmv_short_ms"]; pr["short_n"] += st["spmv_short_launches"]
            pr["comm_ms"] += st["comm_ms"]; pr["comm_n"] += st["comm_launches"]
            pr["kcg_ms"] += st["kcg_ms"]; pr["kcg_iters"] += st["kcg_iters_timed"]
            pr["pair_ms"] += st["pair_ms"]; pr["pair_n"] += 1 if st["pair_ms"] > 0 else 0
        return t5 - t0

    def run(self, steps, warmup, budget_s=None, barrier=None, agree=None):
        """warmup untimed steps, then `steps` timed ones (fewer if they would not fit budget_s; at least one).  Returns (seconds, steps).
        agree: maps a local number to the one every rank uses (rank 0's) -- the budget decisions of ranks in lockstep."""
        import torch
        agree = agree or (lambda v: v)
        tw = 0.0
        for w in range(warmup):
            tw = self.step(False)
            if w == 0 and self.k == 1:
                self.cold = (tw, int(self.host.get_stats()["cg_iters_X"]))
        tw = agree(tw)
        if budget_s is not None and warmup > 0 and tw * steps > budget_s:
            steps = max(1, int(budget_s / tw))
        self.L.dkmc_set_profiling(1)
        self.reset_counters()
        torch.cuda.synchronize()
        if barrier: barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        done = 0
        for _ in range(steps):
            self.step(True); done += 1
            if budget_s is not None and warmup == 0 and done < steps:
                if agree(1.0 if (time.perf_counter() - t0) / done * (done + 1) > budget_s else 0.0) != 0.0:
                    break
        torch.cuda.synchronize()
        if barrier: barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        self.L.dkmc_set_profiling(0)
        return el, done

    def close(self):
        import torch
        del self.gb, self.kmc, self.dev
        torch.cuda.empty_cache()


def rooflines(sim, local_share=1.0):
    """Roofline entries from the HIP-event profile of the timed steps: the dominant kernel (A*p of the CG on X), one CG iteration on
    K, the pair sum.  Algorithmic bytes / flops: SURVEY 8(d) and DESIGN.md section 4."""
    st, pr = sim.host.get_stats(), sim.prof
    out = {}
    if pr["long_n"] > 0:
        avg_ms = pr["long_ms"] / pr["long_n"]
        nnz_all = st["spmv_long_nnz"] + st["spmv_short_nnz"]
        rows_all = st["spmv_long_rows"] + st["spmv_short_rows"]
        flops_per_launch = None
        if st["xt_subblocks"] > 0 and st["xb_width"] > 1 and not st["xb_fallback"]:
            # tiled X, block-CG (default): k_xtb_apply -- 8 KiB per stored 32 x 32 sub-block, read ONCE for both triangles and all `so` vectors
            # (so = the block width rounded up to the matrix instruction's four); per tile a 16 B descriptor, its 32 panel rows read
            # (32 x so x 8 B) and its 32 x so row sums written; per workgroup (four runs of one strip) the strip's 256 panel rows read and one
            # record of 256 x so column sums written.  Flops: 32 x 32 entries x 2 products (row and column sums) x 2 x so.
            so = 4 * ((int(st["xb_width"]) + 3) // 4)
            kname = "k_xtb_apply"
            nrec = st["xt_records"] or max(st["xt_items"] // 4, 1)
            bytes_per_launch = (8192.0 * st["xt_local_subblocks"] + (16.0 + 2 * 32.0 * so * 8.0) * st["spmv_tiles"] * local_share
                                + 32.0 * st["xt_items"] + (256.0 * 16 * 8.0 + 256.0 * so * 8.0) * nrec)
            flops_per_launch = 4096.0 * so * st["xt_local_subblocks"]
        elif st["xt_subblocks"] > 0:
            # tiled X, single-vector loop: k_xt_apply -- one wave per run of tiles of the tunnelling block (8 KiB per stored 32 x 32 sub-block,
            # read once for both triangles; 16 B descriptor and 32 row sums written per tile; 32 B descriptor and 256 column sums
            # per run) plus, in the same launch, the neighbour part Xs in CSR form (12 B per non-zero; per row 8 B row pointer,
            # 8 B result, 8 B scale, 4 B class).  In a sharded solve these are THIS rank's tiles.
            kname = "k_xt_apply"
            bytes_per_launch = (8192.0 * st["xt_local_subblocks"] + (16.0 + 256.0) * st["spmv_tiles"] * local_share
                                + 32.0 * (st["comm_local_segments"] if st["comm_ranks"] else st["xt_items"])
                                + 2048.0 * (st["comm_local_segments"] if st["comm_ranks"] else (st["xt_records"] or st["xt_items"]))
                                + (0.0 if st["xt_split_launch"] else 12.0 * st["xt_sparse_nnz"] + 28.0 * rows_all))
            # (sharded solve: the timed launch is the tile pass alone; the neighbour part runs on a second stream beside the exchange)
        elif st["spmv_segments"] > 0:
            # CSR X, dense-run mode: k_spmv_segs (8 B per entry of a long run, 24 B per segment; short rows in CSR form)
            kname = "k_spmv_segs"
            bytes_per_launch = (8.0 * st["spmv_segment_entries"] + 24.0 * st["spmv_segments"]
                                + 12.0 * st["spmv_short_nnz"] + 24.0 * st["spmv_short_rows"]) * local_share
        else:
            kname = "k_spmv_ap"
            bytes_per_launch = 12.0 * nnz_all + 24.0 * rows_all
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
        out["roofline"] = {
            "bound": "hbm", "kernel": kname, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
            "avg_launch_us": round(avg_ms * 1e3, 2), "launches": pr["long_n"], "algorithmic_bytes_per_launch": bytes_per_launch,
            "csr_equivalent_GBps": round((12.0 * nnz_all + 24.0 * rows_all) * local_share / (avg_ms * 1e-3) / 1e9, 1),
            "tiles": int(st["spmv_tiles"]), "tile_entries": int(st["spmv_tile_entries"]), "subblocks": int(st["xt_subblocks"]),
            "local_subblocks": int(st["xt_local_subblocks"]), "tile_runs": int(st["xt_items"]),
            "row_kernel_us": round(pr["short_ms"] / max(pr["short_n"], 1) * 1e3, 2)}
        if flops_per_launch is not None:
            tf = flops_per_launch / (avg_ms * 1e-3) / 1e12
            out["roofline"]["block_width"] = int(st["xb_width"])
            out["roofline"]["mfma_f64"] = {"instruction": "v_mfma_f64_4x4x4_4b_f64", "flops_per_launch": flops_per_launch, "achieved": round(tf, 2), "peak": FP64_MATRIX_PEAK_TFLOPS,
                                           "unit": "TFLOP/s", "frac": round(tf / FP64_MATRIX_PEAK_TFLOPS, 4),
                                           "measured_instruction_rate_TFLOPs": 68.5,
                                           "note": "flops = 4096 x so per stored sub-block (so = block width rounded up to 4); the instruction's issue rate alone (tools/bench_mfma_f64.hip, "
                                                   "profiles/r04_mfma_f64_rates.txt) is 68-69 TFLOP/s sustained: at width 16 the kernel is bound by the matrix pipe, at width <= 8 by HBM"}
        if bytes_per_launch >= 256.0 * 2 ** 20:
            out["roofline"]["note"] = ("frac is against the 8 TB/s spec peak (the contract's ceiling); MI355X_MICROARCH.md measures 6.3 TB/s for a float4 copy and "
                                       "6.5-6.8 TB/s for a non-temporal read stream: against 6.8 TB/s this launch is at %.2f" % (achieved / 6800.0))
        if bytes_per_launch < 256.0 * 2 ** 20:
            # the contract's ceiling is the HBM peak; a sweep this small is served by the 256 MiB Infinity Cache (MI355X_MICROARCH.md: 7.4-7.9 TB/s
            # gather rate measured), and the launch is one wave generation long: ramp and drain, not bandwidth, set its time
            out["roofline"]["note"] = ("sweep fits the 256 MiB Infinity Cache (PMC FETCH_SIZE still counts the bytes); against the guide's "
                                       "measured cache gather rate of 7.4 TB/s the fraction is %.2f" % (achieved / 7400.0))
    if pr["kcg_iters"] >= 32 * max(sim.cnt["steps"], 1):
        # (warm-started K solves of 1-4 iterations are a host poll, not a kernel measurement: no entry below 32 iterations per solve)
        # one Jacobi-CG iteration on K (SpMV + update + direction): 12 nnz + 4 (m + 1) + 96 m bytes
        m, nnz = sim.s.N - 2 * sim.p.num_atoms_first_layer, int(sim.gb.c.Device_nnz)
        if sim.p.cb_edge_domain == "atoms":
            pass        # (the potential system is over every site in both domains; only the CB-edge solve changes)
        # bytes the two kernels of an iteration move (the library's own count, dkmc_stats.kcg_bytes): CSR positions: 4 B per stored entry
        # (column | class bit, no value array) + row pointers + 15 vector touches of 8 B; blocked form (systems up to 262 144 rows):
        # 4 B per padded entry + 8 B per column of every block's LDS window + 14 vector touches
        blocked = bool(st["kcg_blocked"])
        b = float(st["kcg_bytes"]) if st["kcg_bytes"] > 0 else 4.0 * nnz + 4.0 * (m + 1) + 120.0 * m
        b_csr = 12.0 * nnz + 4.0 * (m + 1) + 96.0 * m
        us = pr["kcg_ms"] / pr["kcg_iters"] * 1e3
        out["roofline_K_cg"] = {"bound": "hbm", "kernel": "one CG iteration on K (%s + k_kc_step)" % ("k_kb_apply, blocked form" if blocked else "k_kc_apply"),
                                "achieved": round(b / us / 1e3, 1),
                                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(b / us / 1e3 / HBM_PEAK_GBS, 4), "traffic": None,
                                "us_per_iteration": round(us, 2), "iterations_timed": pr["kcg_iters"], "algorithmic_bytes_per_iteration": b,
                                "blocked_form": blocked,
                                "note": "achieved / frac count the bytes this formulation moves; csr_equivalent_* prices the same iteration at the bytes of "
                                        "the reference's CSR formulation (SURVEY 8d: 12 nnz + 4 (m + 1) + 96 m). Below ~2e5 rows the iteration is two "
                                        "dependent launches of a few microseconds each, bound by launch + memory latency, not by bytes (DESIGN 4)",
                                "csr_equivalent_bytes_per_iteration": b_csr, "csr_equivalent_GBps": round(b_csr / us / 1e3, 1)}
    if pr["pair_n"] > 0 and st["n_charged"] > 0:
        # pair sum: 64 fp64 flops per evaluated (site, charged site) pair; pairs beyond the screening cut-off
        # (erfc < 3.8e-20) pay the distance only, 12 flops
        pairs = int(sim.s.N) * int(st["n_charged"]) // max(int(st["comm_ranks"]), 1)      # a sharded run: this rank's slab of sites
        ev = int(st["pair_evaluated"]) if st["pair_evaluated"] > 0 else pairs
        tested = int(st["pair_tested"]) if st["pair_tested"] > 0 else pairs            # with the cell list: the 3 x 3 columns around a site
        fl = 64.0 * ev + 12.0 * max(tested - ev, 0)
        ms = pr["pair_ms"] / pr["pair_n"]
        out["roofline_pair_sum"] = {"bound": "fp64-valu", "kernel": "k_pairwise_cells" if tested < 0.9 * pairs else "k_pairwise",
                                    "achieved": round(fl / (ms * 1e-3) / 1e12, 2),
                                    "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(fl / (ms * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TFLOPS, 4),
                                    "traffic": None, "ms_per_launch": round(ms, 4), "pairs": pairs, "pairs_tested": tested, "pairs_evaluated": ev,
                                    "note": "ms_per_launch spans the whole call (compaction, binning, the sum); flops = 64 per evaluated pair + 12 per "
                                            "distance test that rejects (SURVEY 8d)",
                                    "all_pairs_equivalent_TFLOPs": round(64.0 * pairs / (ms * 1e-3) / 1e12, 2)}
    return out


def summary(sim, elapsed, steps):
    st = sim.host.get_stats()
    n = max(sim.cnt["steps"], 1)
    return {
        "sites": int(sim.s.N), "nn": int(sim.dev.max_num_neighbors), "atoms": int(sim.dev.N_atom),
        "ms_per_step": round(elapsed / steps * 1e3, 3), "steps_per_s": round(steps / elapsed, 5),
        "split_ms": {k: round(v / n * 1e3, 3) for k, v in sim.phases.items()},
        "per_step": {"events": sim.cnt["events"] / n, "cg_iters_K": sim.cnt["cg_iters_K"] / n, "cg_iters_X": sim.cnt["cg_iters_X"] / n,
                     "X_nnz": int(st["X_nnz"]), "n_charged": int(st["n_charged"]), "tunnelling_set": int(st["xt_ns"]),
                     "K_rows": int(sim.s.N - 2 * sim.p.num_atoms_first_layer), "K_nnz": int(sim.gb.c.Device_nnz)},
        "setup_s": round(sim.setup_s, 2),
    }


XGMI_LINK_GBPS_ASSUMED = 50.0      # achieved rate per direction of ONE xGMI link (MI355X: 7 links x ~153 GB/s bidirectional per GPU, one per peer): ASSUMED, never measured here
XCHG_LATENCY_US_ASSUMED = 15.0     # per all-to-all-v (grouped ncclSend / ncclRecv); the 12 KB Gram all-gather: 25 us.  ASSUMED.
SLAB_CLASSES = ("tile_x_panel", "neighbour_part", "fold_and_pack", "rows_and_gram", "gram_reduction", "sxs_algebra", "panel_step", "pack_unpack_exchange3")


def strong_scaling_model_slabs(sim, res, sweeps=28):
    """Strong-scaling model of the SLAB-DISTRIBUTED block-CG (csrc/xtb_slab.inc), rebuilt from per-rank kernel times measured on ONE GPU: for N = 1, 2,
    4, 8 the distributed loop runs with N virtual ranks inside this process on the resident X (dkmc_xtb_emulate_slabs: shares of the tiles, row
    slabs, lists and exchange buffers exactly as N processes build them, exchanges as device copies) and every kernel of the middle rank is timed
    over sweeps 2 ... 25.  What cannot be measured on one GPU are the three exchanges of a sweep: their PAYLOADS are the solver's own counts, their
    durations are ASSUMED (payload per pair / 50 GB/s + 15 us; Gram all-gather 25 us).  step_N = step_1 - sweeps x (sweep_1 - sweep_N): everything
    outside the sweeps (K-CG, pair sum, events, assembly) counted as replicated."""
    import ctypes as C
    iters = res["per_step"]["cg_iters_X"]
    step_ms = res["ms_per_step"]
    rows = {}
    for n in (1, 2, 4, 8):
        rd, it_s, it_r = C.c_double(0), C.c_int(0), C.c_int(0)
        us, xd, mm = (C.c_double * 8)(), (C.c_longlong * 3)(), (C.c_int * 2)()
        rc = sim.L.dkmc_xtb_emulate_slabs(n, sim.x_block, sim.p.cg_tol, n // 2, sweeps, C.byref(rd), C.byref(it_s), C.byref(it_r), us, xd, mm)
        if rc != 0:
            err = sim.L.dkmc_last_error().decode(); sim.L.dkmc_clear_error()
            return {"error": "dkmc_xtb_emulate_slabs(%d) failed: %s" % (n, err[:200])}
        k = {name: round(us[i], 1) for i, name in enumerate(SLAB_CLASSES)}
        pair = lambda doubles: 0.0 if n == 1 else doubles * 8.0 / (n - 1) / (XGMI_LINK_GBPS_ASSUMED * 1e3) + XCHG_LATENCY_US_ASSUMED      # us
        x1, x2, x3 = pair(xd[0]), (0.0 if n == 1 else 25.0), pair(xd[2])
        sweep = sum(us[i] for i in range(8)) + x1 + x2 + x3
        rows[n] = {"kernel_us": k, "kernels_us_per_sweep": round(sum(us[i] for i in range(8)), 1), "rows_per_slab_min_max": [mm[0], mm[1]],
                   "doubles_received_per_sweep": {"exchange1_tile_sums_to_owners": xd[0], "exchange2_gram": xd[1], "exchange3_QS_and_halo": xd[2]},
                   "exchange_us_ASSUMED": {"exchange1": round(x1, 1), "exchange2": x2, "exchange3": round(x3, 1)}, "sweep_us": round(sweep, 1)}
    s1 = rows[1]["sweep_us"]
    for n in rows:
        tn = step_ms - iters * (s1 - rows[n]["sweep_us"]) * 1e-3
        rows[n]["modelled_ms_per_step"] = round(tn, 1)
        rows[n]["modelled_speedup"] = round(step_ms / tn, 2)
        rows[n]["speedup_of_the_sweep"] = round(s1 / rows[n]["sweep_us"], 2)
    return {"what": "slab-distributed block-CG: per-rank kernel times of the middle rank of N virtual ranks measured on ONE GPU (each kernel bracketed by events, "
                    "host-synchronised: small kernels include their launch latency); exchange payloads counted by the solver, exchange durations ASSUMED "
                    "(%g GB/s per pair and direction + %g us; Gram all-gather 25 us); nothing here ran on more than one GPU" % (XGMI_LINK_GBPS_ASSUMED, XCHG_LATENCY_US_ASSUMED),
            "cg_sweeps_X_per_step": iters, "single_gpu_ms_per_step": step_ms, "outside_the_sweeps_ms": round(step_ms - iters * s1 * 1e-3, 1), "by_n_gpus": rows,
            "note": "outside_the_sweeps_ms (K-CG, pair sum, events, assembly of X) is replicated on every rank: with the warm start it is what caps the step's speed-up"}


ALLREDUCE_US_ASSUMED = {1: 0.0, 2: 25.0, 4: 40.0, 8: 60.0}     # in-place all-reduce of |S| + 1 doubles (0.74 MB at 9.4e5 sites) over xGMI: ASSUMED, not measured


def strong_scaling_model(sim, ms_per_step, reps=10):
    """What can be measured of an N-GPU strong-scaling run on ONE GPU (dkmc_xt_time_share): the time per CG iteration of the kernels
    one rank of an N-way sharded solve runs on its share of the tiles (a middle rank's share, work items sized as an N-rank run
    sizes them), each kernel timed on its own.  The model adds an ASSUMED all-reduce latency:
    T_N = T_1 - iters * (iteration_1 - iteration_N), iteration_N = tile pass + max(partial row sums + allreduce_N, neighbour part)
    + finish + vector step."""
    import ctypes as C
    # the LAST timed step stands for the steady state (the first one of a fresh simulation also fills the coefficient cache, 0.4 s)
    if sim.step_log:
        ms_per_step, iters = sim.step_log[-1][0] * 1e3, float(sim.step_log[-1][1])
    else:
        iters = sim.cnt["cg_iters_X"] / max(sim.cnt["steps"], 1)
    rows = {}
    for n in (1, 2, 4, 8):
        a, r, it, sb = C.c_double(0), (C.c_double * 4)(), C.c_int(0), C.c_longlong(0)
        rc = sim.L.dkmc_xt_time_share(n, n // 2, reps, C.byref(a), r, C.byref(it), C.byref(sb))
        if rc != 0:
            sim.L.dkmc_clear_error()
            return {"error": "dkmc_xt_time_share failed"}
        rows[n] = {"apply_us": round(a.value, 2), "partial_row_sums_us": round(r[0], 2), "finish_us": round(r[1], 2), "vector_step_us": round(r[2], 2),
                   "neighbour_part_us": round(r[3], 2), "work_items": it.value, "subblocks": sb.value,
                   "share_GBps": round(8192.0 * sb.value / a.value / 1e3, 1)}
    # one GPU: apply (tiles + neighbour part in one launch) + row sums (finish fused in) + vector step.
    # N > 1: tile pass, then [partial row sums + all-reduce] beside [neighbour part on the second stream], then finish + vector step
    t1 = rows[1]["apply_us"] + rows[1]["neighbour_part_us"] + rows[1]["partial_row_sums_us"] + rows[1]["vector_step_us"]
    # (neighbour_part_us of N = 1 is non-zero when the one-GPU solve runs it as its own kernel behind the tile pass: multi-GB sweeps)
    for n in rows:
        if n == 1:
            tk = t1
        else:
            tk = (rows[n]["apply_us"] + max(rows[n]["partial_row_sums_us"] + ALLREDUCE_US_ASSUMED[n], rows[n]["neighbour_part_us"])
                  + rows[n]["finish_us"] + rows[n]["vector_step_us"])
        tn = ms_per_step - iters * (t1 - tk) * 1e-3
        rows[n]["allreduce_us_assumed"] = ALLREDUCE_US_ASSUMED[n]
        rows[n]["modelled_ms_per_step"] = round(tn, 1)
        rows[n]["modelled_speedup"] = round(ms_per_step / tn, 2)
    return {"what": "per-rank kernel times of an N-way sharded solve measured on one GPU (share of a middle rank), each kernel on its own; all-reduce "
                    "latency ASSUMED; everything outside the CG iterations on X (K-CG, pair sum, events, assembly) counted as replicated",
            "cg_iters_X_per_step": iters, "single_gpu_ms_per_step": round(ms_per_step, 3), "basis": "last timed step of this scale point",
            "not_in_the_cg_iterations_ms": round(ms_per_step - iters * t1 * 1e-3, 1), "by_n_gpus": rows}


def k_slab_model(sim, iters=28):
    """configs[4]'s domain-decomposed potential: the CG on K distributed by lateral row slabs (csrc/kcg.hip), N virtual ranks in this process on the
    system of the resident state (dkmc_kcg_emulate_slabs, a measurement run of `iters` iterations): the middle rank's kernel times per iteration, the
    halo payload, and an iteration time with ASSUMED exchanges (two all-gathers of block partials at 25 us each + the halo at 50 GB/s per pair + 15 us)."""
    import ctypes as C
    n1 = sim.p.num_atoms_first_layer
    rows = {}
    for n in (1, 2, 4, 8):
        md, it_s, it_r = C.c_double(0), C.c_int(0), C.c_int(0)
        us, hr = (C.c_double * 4)(), (C.c_longlong * 2)()
        rc = sim.L.dkmc_kcg_emulate_slabs(C.byref(sim.gb.c), sim.dev.N, n1, n1, sim.Vd, sim.p.high_G, sim.p.low_G, len(sim.p.metals), n, n // 2, iters,
                                          C.byref(md), C.byref(it_s), C.byref(it_r), us, hr)
        if rc != 0:
            err = sim.L.dkmc_last_error().decode(); sim.L.dkmc_clear_error()
            return {"error": "dkmc_kcg_emulate_slabs(%d) failed: %s" % (n, err[:200])}
        kern = sum(us[i] for i in range(4))
        x = 0.0 if n == 1 else 2 * 25.0 + hr[0] * 8.0 / 2 / (XGMI_LINK_GBPS_ASSUMED * 1e3) + XCHG_LATENCY_US_ASSUMED      # (a slab has two neighbours)
        rows[n] = {"kernel_us": {"product": round(us[0], 1), "update": round(us[1], 1), "direction": round(us[2], 1), "halo_pack_unpack": round(us[3], 1)},
                   "rows_of_the_largest_slab": hr[1], "halo_doubles_received_per_iteration": hr[0], "exchanges_us_ASSUMED": round(x, 1), "iteration_us": round(kern + x, 1)}
    for n in rows:
        rows[n]["speedup_of_the_iteration"] = round(rows[1]["iteration_us"] / rows[n]["iteration_us"], 2)
    return {"what": "slab-distributed CG on K: per-rank kernel times of the middle rank of N virtual ranks measured on ONE GPU (event-bracketed, host-synchronised: "
                    "each includes its launch latency); exchange durations ASSUMED; nothing here ran on more than one GPU", "by_n_gpus": rows}


def cpu_model():
    """Model string of the host CPU the cpu_baseline legs ran on (BASELINE.md section 3 asks for it)."""
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.lower().startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine() or "unknown"


def cpu_cg_baseline(sim, ncores, iters=None, iters_source=None):
    """CPU leg of a scale point: the oracle's CG iteration (okmc_cg_iter_bench: the loop body of okmc_cg_jacobi, OpenMP) timed on a
    CSR of X's shape at this size, x the GPU run's iteration count.  Everything else of a CPU step (assembly, K solve, events) is
    left out, so the CPU time is a lower bound."""
    os.environ["OMP_NUM_THREADS"] = str(ncores)
    from oracle import oracle as oc
    st = sim.host.get_stats()
    m = int(st["N_atom"]) + 1
    n_timed = 16                                  # ~12 s of CPU at 9.4e5 sites (0.7 s per iteration): the bounded sample of the contract
    t_it = oc.cg_iter_bench(m, int(st["xt_ns"]), 2 * int(st["spmv_tile_entries"]), int(st["xt_sparse_nnz"]), n_timed)
    if iters is None:
        iters, iters_source = sim.cnt["cg_iters_X"] / max(sim.cnt["steps"], 1), "the GPU run's iteration count"
    if t_it <= 0:
        return None
    sec = t_it * iters
    return {"value": round(1.0 / sec, 6), "unit": "KMC steps/s", "cores": ncores, "cpu_model": cpu_model(), "host_cpus": os.cpu_count(), "kind": "port",
            "sample": "oracle CG iteration (loop body of okmc_cg_jacobi: CSR SpMV + 3 dots + 3 vector updates, OpenMP; oracle/kmc_oracle.c "
                      "okmc_cg_iter_bench) timed over 16 iterations on a CSR of X's row and non-zero counts at this size, x %.0f CG iterations per step "
                      "(%s: the CPU port runs the reference's single-vector CG); assembly, K solve, pair sum and events of a CPU step are NOT "
                      "included (lower bound on the CPU time)" % (iters, iters_source),
            "s_per_cg_iteration": round(t_it, 4), "ms_per_step": round(sec * 1e3, 1)}


def cpu_superstep_baseline(s, p, ncores):
    os.environ["OMP_NUM_THREADS"] = str(ncores)   # read by libgomp when the oracle library is loaded
    from oracle import oracle as oc
    o = oc.OracleKMC(s.element, s.x, s.y, s.z, p)
    o.set_laplace_potential(VD)
    o.superstep(VD)                              # untimed: cold-start CG of the first step
    t0 = time.perf_counter()
    nsamp = 1 if s.N > 30000 else 5
    for _ in range(nsamp):
        o.superstep(VD)
    tc = (time.perf_counter() - t0) / nsamp
    return {"value": round(1.0 / tc, 5), "unit": "KMC steps/s", "cores": ncores, "cpu_model": cpu_model(), "host_cpus": os.cpu_count(), "kind": "port",
            "sample": "%d superstep(s) of the same workload after one untimed step (oracle/kmc_oracle.c, OpenMP)" % nsamp,
            "ms_per_step": round(tc * 1e3, 1), "split_ms": {k: round(v * 1e3, 2) for k, v in o.timing.items()}}


def cpp_host_crosscheck(sim, steps, warmup):
    """The same timed loop driven by the C++ host (devicekmc_amd/host/kmc_superstep: the reference's call sequence and argument lists
    through include/gpu_solvers.h -> C ABI) instead of the Python mirror: steps/s of a child process on the same workload."""
    import subprocess
    import tempfile
    from devicekmc_amd import io as kio, structure
    exe = os.path.join(ROOT, "devicekmc_amd", "host", "kmc_superstep")
    if not os.path.exists(exe):
        return {"error": "kmc_superstep not built"}
    d = tempfile.mkdtemp(prefix="dkmc_cpp_", dir="/tmp")
    try:
        element, neigh, nn, layer = structure.prepare_device(sim.s, sim.p, (sim.dev.neigh_idx, sim.dev.max_num_neighbors))
        fin, fout = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
        kio.write_host_bundle(fin, sim.s, sim.p, VD, element, neigh, nn, layer)
        r = subprocess.run([exe, fin, fout, str(steps), str(warmup)], capture_output=True, text=True, timeout=300)
        line = [l for l in r.stdout.splitlines() if l.startswith("TIMING")]
        if r.returncode != 0 or not line:
            return {"error": (r.stderr or r.stdout)[-300:]}
        sec = float(line[0].split("seconds=")[1])
        return {"host": "C++ driver over the drop-in shim (child process)", "steps": steps, "warmup": warmup,
                "value": round(steps / sec, 4), "ms_per_step": round(sec / steps * 1e3, 3)}
    except Exception as exc:
        return {"error": repr(exc)[:300]}
    finally:
        import shutil
        shutil.rmtree(d, ignore_errors=True)


def pmc_traffic(workload, kernel_prefix, x_format, cg_tol=None, x_block=None):
    """HBM bytes per launch of the dominant kernel from the PMC counters, measured in this run: two rocprofv3 child runs of this
    script on the same workload (one step), `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` in separate passes with the kernel trace only
    (MI355X_MICROARCH.md, HBM / rocprofv3 sections); median over the kernel's working launches (the no-op launches behind the
    last iteration of a batch fetch next to nothing); gfx950 correction: FETCH_SIZE counts half of a wide coalesced streaming
    read, so bytes = (2 FETCH_SIZE + WRITE_SIZE) x 1024.  Returns (bytes or None, detail)."""
    import csv
    import glob
    import shutil
    import statistics
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, {"error": "rocprofv3 not found"}
    med = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="dkmc_pmc_", dir="/tmp")
        cmd = [exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "--",
               sys.executable, os.path.abspath(__file__), "--workload", workload, "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
               "--scale-points", "none", "--no-alt", "--no-pmc", "--no-cpp-host", "--no-log-tolerance", "--no-device", "--no-reference-order",
               "--x-format", str(x_format), "--x-block", str(x_block if x_block is not None else X_BLOCK)]
        if cg_tol is not None:      # bytes per launch do not depend on how many iterations the solve takes: a loose tolerance shortens the (serialised) counter run
            cmd += ["--cg-tol", repr(cg_tol)]
        try:
            subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=240, check=True,
                           cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
            vals = []
            for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for r in csv.DictReader(fh):
                        if r["Counter_Name"] == counter and kernel_prefix in r["Kernel_Name"]:
                            vals.append(float(r["Counter_Value"]))
            if not vals:
                return None, {"error": "no %s samples of %s" % (counter, kernel_prefix)}
            # working launches only, in both passes: the no-op launches behind the last iteration of a batch move next to nothing
            top = max(vals)
            work = [v for v in vals if v > 0.5 * top]
            med[counter] = (statistics.median(work), len(work))
        except Exception as exc:
            return None, {"error": repr(exc)[:200]}
        finally:
            shutil.rmtree(d, ignore_errors=True)
    b = (2.0 * med["FETCH_SIZE"][0] + med["WRITE_SIZE"][0]) * 1024.0
    return b, {"FETCH_SIZE_KB_median": med["FETCH_SIZE"][0], "WRITE_SIZE_KB_median": med["WRITE_SIZE"][0], "launches": med["FETCH_SIZE"][1], "launches_write_pass": med["WRITE_SIZE"][1],
               "formula": "(2 * FETCH_SIZE + WRITE_SIZE) * 1024, separate --pmc passes"}


def log_tolerance_block(devname, x_format, steps, warmup):
    """The 85 071-site workload under KMCParameters.log_revision() (CG tolerance 1e-12, "used to be 1e-12", iterative_solvers_gpu.cu:322;
    CB edge solved on atoms): the configuration in which KMC time AND current of all 19 supersteps of the reference's own CUDA-path log
    are reproduced to the printed digits (tests/test_gpu_parity.py::test_reference_log_7p5_currents).  Same timing rule as `value`; the
    steps run here are checked against the log on the way (fixture tests/golden/reference_logs.json; no oracle involved)."""
    sim = Sim("7.5nm", devname, x_format=x_format, log_revision=True)
    sim.p.solve_heating_global = False          # the log was written with heating off (its parameters.txt); the timed phases are the log's
    elapsed, n = sim.run(steps, warmup)
    res = summary(sim, elapsed, n)
    blk = {"workload": "7.5nm", "cg_tol": sim.p.cg_tol, "cb_edge_domain": sim.p.cb_edge_domain, "steps": n, "warmup": warmup,
           "value": round(n / elapsed, 4), "unit": "KMC steps/s", "ms_per_step": res["ms_per_step"], "split_ms": res["split_ms"],
           "per_step": {k: res["per_step"][k] for k in ("events", "cg_iters_K", "cg_iters_X", "X_nnz")}}
    try:
        with open(os.path.join(ROOT, "tests", "golden", "reference_logs.json")) as f:
            gold = json.load(f)["timing_7.5nm/output_noguess.txt"]["steps"]
        # sim.trace holds the timed steps = logged steps warmup .. warmup + n - 1 (the warm-up steps were steps 0 .. warmup - 1)
        t_log0 = gold[warmup - 1]["KMC time"] if warmup > 0 else 0.0
        t, worst_t, worst_i, k = t_log0, 0.0, 0.0, 0
        for (dt, im, _), g in zip(sim.trace, gold[warmup:]):
            t += dt; k += 1
            worst_t = max(worst_t, abs(t / g["KMC time"] - 1)); worst_i = max(worst_i, abs(im * 1e6 - g["Current [uA]"]))
        blk["vs_reference_log"] = {"steps_compared": k, "max_abs_current_diff_uA": worst_i, "max_rel_kmc_time_diff": worst_t,
                                   "log": "structures/single_devices/timing_7.5nm/output_noguess.txt (6 printed digits)",
                                   "agrees_to_printed_digits": bool(worst_i <= 1e-4 and worst_t < 1e-5)}
    except Exception as exc:
        blk["vs_reference_log"] = {"error": repr(exc)[:200]}
    sim.close()
    return blk


def crossbar_block(devname):
    """The crossbar the reference timed (structures/crossbars/timing_10nm_5pitch/output_initial.txt: 110 813 sites, V = 1, solve_current = 0,
    13 supersteps, 2.04 s median per superstep on its unnamed GPU) under KMCParameters.log_revision(): the KMC time of every timed step is
    checked against that log (fixture tests/golden/reference_logs.json; no oracle involved).  With the current solve off, a step is the charge
    update, the potential (K-CG + pair sum) and the event loop."""
    sim = Sim("crossbar_10nm_5pitch", devname, log_revision=True, Vd=1.0)
    nsteps, warm = 12, 1
    elapsed, n = sim.run(nsteps, warm)
    res = summary(sim, elapsed, n)
    blk = {"workload": "crossbar_10nm_5pitch", "sites": res["sites"], "Vd": 1.0, "solve_current": False, "cg_tol": sim.p.cg_tol, "cb_edge_domain": sim.p.cb_edge_domain,
           "steps": n, "warmup": warm, "value": round(n / elapsed, 4), "unit": "KMC steps/s", "ms_per_step": res["ms_per_step"], "split_ms": res["split_ms"],
           "cold_step_ms": round(sim.cold[0] * 1e3, 1) if sim.cold else None,
           "per_step": {k: res["per_step"][k] for k in ("events", "cg_iters_K", "n_charged", "K_rows", "K_nnz")},
           "us_per_executed_event": round(res["split_ms"]["rates"] * 1e3 / max(res["per_step"]["events"], 1), 1),
           "reference_cuda_log": {"s_per_superstep_median": 2.04, "rates_s_median": 1.97, "potential_boundaries_s_median": 0.012,
                                  "source": "structures/crossbars/timing_10nm_5pitch/output_initial.txt:11-14 (BASELINE.md; unnamed GPU)"}}
    blk["speedup_over_reference_log"] = round(2.04 / (elapsed / n), 1)
    blk.update(rooflines(sim))
    try:
        with open(os.path.join(ROOT, "tests", "golden", "reference_logs.json")) as f:
            gold = json.load(f)["crossbars/timing_10nm_5pitch/output_initial.txt"]["steps"]
        # sim.trace holds the timed steps = logged steps warm .. warm + n - 1; the KMC time is cumulative from the log's value before them
        t, worst, k = (gold[warm - 1]["KMC time"] if warm > 0 else 0.0), 0.0, 0
        for (dt, _, _), g in zip(sim.trace, gold[warm:]):
            t += dt; k += 1
            worst = max(worst, abs(t / g["KMC time"] - 1))
        blk["vs_reference_log"] = {"steps_compared": k, "max_rel_kmc_time_diff": worst, "agrees_to_printed_digits": bool(worst < 1e-5),
                                   "log": "structures/crossbars/timing_10nm_5pitch/output_initial.txt (6 printed digits)"}
    except Exception as exc:
        blk["vs_reference_log"] = {"error": repr(exc)[:200]}
    sim.close()
    return blk


def arm_watchdog(seconds, rank, what):
    """A rank stuck in a collective cannot be unwound: report and leave with a non-zero exit code."""
    def fire():
        sys.stderr.write("bench.py: rank %d: %s did not finish within %.0f s -- aborting (exit 3)\n" % (rank, what, seconds))
        sys.stderr.flush()
        if rank == 0:
            print(json.dumps({"metric": "KMC steps/sec", "value": None, "error": "%s did not finish within %.0f s" % (what, seconds)}), flush=True)
        os._exit(3)
    t = threading.Timer(seconds, fire); t.daemon = True; t.start()
    return t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=None, help="default: tile:10 (configs[2])")
    ap.add_argument("--no-device", action="store_true", help="N = 1: skip the block on the reference's own 85 071-site device")
    ap.add_argument("--no-reference-order", action="store_true", help="N = 1: skip the runs with dkmc_set_x_block(1), the reference's single-vector CG")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--warm-start", type=int, default=1, help="dkmc_set_current_warm_start: 1 = library default (previous solution), 0 = the reference code's G0-scaled buffer")
    ap.add_argument("--x-format", type=int, default=1, help="1: tiled X (default); 0: CSR X as the reference stores it")
    ap.add_argument("--mode", choices=["sharded", "replicas"], default="sharded",
                    help="N > 1: ONE simulation, X sharded over the ranks (strong scaling, default) or independent replicas (weak scaling)")
    ap.add_argument("--scale-points", default=None, help="N = 1: comma list of extra workloads measured in the same run (default tile:5,crossbar_10nm_5pitch,tile:20:nocurrent; 'none')")
    ap.add_argument("--no-replicas", action="store_true", help="N > 1: skip the replicas block")
    ap.add_argument("--no-single-ref", action="store_true", help="N > 1: skip the single-GPU run of the same simulation (reference point of the speed-up)")
    ap.add_argument("--no-alt", action="store_true", help="skip the reference_start_vector block")
    ap.add_argument("--no-scaling-model", action="store_true", help="N = 1: skip strong_scaling_model / k_cg_slab_model (the virtual-rank emulations launch the solver's kernels on per-rank shares: "
                                                                    "a kernel-trace profile of the run would average them in)")
    ap.add_argument("--no-cpp-host", action="store_true", help="N = 1: skip the cross-check line through the C++ host driver")
    ap.add_argument("--no-pmc", action="store_true", help="N = 1: do not measure roofline.traffic (two rocprofv3 --pmc child runs)")
    ap.add_argument("--no-log-tolerance", action="store_true", help="N = 1: skip the at_log_tolerance block")
    ap.add_argument("--peer-exchange", action="store_true", help="N > 1, sharded: exchange the block loop's slots by peer writes over hipIpc-mapped buffers instead of the transport's all-gather (also DKMC_PEER_EXCHANGE=1)")
    ap.add_argument("--k-blocked", type=int, default=1, help="dkmc_set_k_blocked: 1 = library default (K-CG on the blocked form up to 262 144 rows), 0 = CSR positions")
    ap.add_argument("--x-block", type=int, default=16, help="dkmc_set_x_block: 16 = library default (block-CG), 1 = the reference's single-vector CG on X")
    ap.add_argument("--cg-tol", type=float, default=None, help="override the CG tolerance (default: the reference's 1e-6)")
    ap.add_argument("--budget", type=float, default=420.0, help="time budget [s] for the timed steps of seconds-per-step workloads")
    ap.add_argument("--timeout", type=float, default=570.0, help="watchdog [s]: exit 3 if the run has not finished (a rank stuck in a collective cannot be unwound)")
    args = ap.parse_args()

    # ---- `python bench.py --gpus N` with N > 1 and no rank environment: start the N ranks ourselves (one process per GPU), BEFORE anything
    # touches the GPU in this process; relay rank 0's JSON line as the last line of stdout and leave with the child's exit code ----
    if args.gpus > 1 and "RANK" not in os.environ:
        from devicekmc_amd import launch
        sys.exit(launch.run_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env != args.gpus:
        # never a silent N = 1 line under --gpus N (or the reverse): the line would be read as an N-GPU measurement
        if int(os.environ.get("RANK", "0")) == 0:
            print(json.dumps({"metric": "KMC steps/sec", "value": None, "n_gpus": world_env,
                              "error": "--gpus %d but the process group has WORLD_SIZE=%d ranks" % (args.gpus, world_env)}), flush=True)
        sys.exit(5)

    import torch
    from devicekmc_amd import parallel
    # rehearsal on a one-GPU box: DKMC_BENCH_BACKEND=gloo DKMC_BENCH_SINGLE_DEVICE=1 lets several ranks share cuda:0
    backend = os.environ.get("DKMC_BENCH_BACKEND", "nccl")
    if os.environ.get("DKMC_BENCH_SINGLE_DEVICE"):
        os.environ["LOCAL_RANK"] = "0"
    rank, world, local_rank = parallel.init(backend)
    torch.cuda.set_device(local_rank)
    devname = "cuda:%d" % local_rank
    red_dev = devname if backend == "nccl" else "cpu"
    wd = arm_watchdog(args.timeout, rank, "bench.py")
    from devicekmc_amd import lib as _dlib
    global X_BLOCK
    X_BLOCK = args.x_block
    _dlib.load().dkmc_set_x_block(args.x_block)
    _dlib.load().dkmc_set_k_blocked(args.k_blocked)
    ncores = min(16, os.cpu_count() or 1)         # the box's CPU share for one GPU; more threads only add contention
    out = None

    if world == 1:
        name = args.workload or "tile:10"
        big = name.startswith("tile:") and int(name.split(":")[1]) >= 8
        ncpu = ncores
        # ================= the headline: steady-state steps of the default workload at the library's defaults =================
        sim = Sim(name, devname, x_format=args.x_format, warm_start=args.warm_start, cg_tol=args.cg_tol)
        elapsed, n = sim.run(args.steps, args.warmup, budget_s=args.budget if sim.s.N > 150000 else None)
        res = summary(sim, elapsed, n)
        roofs = rooflines(sim)
        out = {
            "metric": "KMC steps/sec", "value": round(n / elapsed, 5), "unit": "KMC steps/s", "n_gpus": 1, "steps": n, "warmup": args.warmup,
            "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": "none", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": name, "sites": res["sites"], "nn": res["nn"], "atoms": res["atoms"], "Vd": sim.Vd,
                       "phases": "charge+potential+rates+current+heat", "parallelism": "single GPU", "x_format": "tiled" if args.x_format else "csr",
                       "current_warm_start": sim.warm_start, "cg_tol": sim.p.cg_tol, "x_block": sim.x_block,
                       "start_vector_of_X": "previous step's solution (library default)" if sim.warm_start else "the reference code's G0-scaled buffer",
                       "cg_on_X": ("block-CG of width %d, tile x panel product on the matrix cores (csrc/xtb.hip)" % sim.x_block) if sim.x_block > 1
                                  else "single-vector CG in the reference's iterate order (csrc/xt.hip)",
                       "x_aux_columns": ("smooth (lowest Laplacian modes of the bounding box / s)" if sim.host.get_stats()["xb_aux"] else "fixed-seed hash") if sim.x_block > 1 else None,
                       "x_poly": (("split polynomial preconditioner of degree %d on the neighbour part of X (library default; dkmc_set_x_poly): the block loop runs on L A L, "
                                   "2 d sparse panel products per sweep, stop test checked on the TRUE residual" % sim.L.dkmc_get_x_poly())
                                  if (sim.x_block > 1 and world == 1 and sim.L.dkmc_get_x_poly() > 0) else None)},
            "split_ms": res["split_ms"], "per_step": res["per_step"],
            "steady": {"steps": n, "ms_each": [round(t * 1e3, 1) for t, _ in sim.step_log], "cg_sweeps_X_each": [i for _, i in sim.step_log]},
            "cold_step": ({"ms": round(sim.cold[0] * 1e3, 1), "cg_sweeps_X": sim.cold[1],
                           "note": "first step of a fresh simulation (in the warm-up): fills the tunnelling-coefficient cache, sizes every buffer, zero start vector"}
                          if sim.cold else None),
            "cpu_baseline": None,
        }
        if n != args.steps:
            out["steps_requested"] = args.steps
        out.update(roofs)
        # ---- the same simulation without the split polynomial preconditioner (dkmc_set_x_poly(0)): what it saves, and the baseline of the strong-scaling
        # model below (the slab-distributed loop carries no preconditioner yet: the model compares plain loop with plain loop) ----
        res_plain = None
        pd_default = sim.L.dkmc_get_x_poly()
        if sim.x_block > 1 and pd_default > 0 and not args.no_alt:
            sim.L.dkmc_set_x_poly(0)
            t0 = time.perf_counter(); itp = 0; npl = 2 if big else min(n, 5)
            for _ in range(npl):
                sim.step(False); itp += sim.host.get_stats()["cg_iters_X"]
            tp = time.perf_counter() - t0
            sim.L.dkmc_set_x_poly(pd_default)
            res_plain = {"per_step": {"cg_iters_X": itp / npl}, "ms_per_step": round(tp / npl * 1e3, 3)}
            out["plain_block_loop"] = {"x_poly": 0, "steps": npl, "value": round(npl / tp, 5), "ms_per_step": res_plain["ms_per_step"], "cg_sweeps_X": itp / npl,
                                       "note": "the next %d steps of the same simulation with dkmc_set_x_poly(0): the block-CG on the Jacobi-scaled X as in round 4" % npl}
            out["x_poly_gain"] = {"sweeps": round(itp / npl / max(res["per_step"]["cg_iters_X"], 1), 2), "steps_per_s": round(out["value"] / (npl / tp), 2),
                                  "note": "ratio of different steps of one trajectory (the sweeps of a step follow its events): tools/ab_x_poly.py compares the same steps"}
        if big and sim.x_block > 1 and "roofline" in out and not args.no_scaling_model:
            out["strong_scaling_model"] = strong_scaling_model_slabs(sim, res_plain or res)
            if res_plain:
                out["strong_scaling_model"]["baseline"] = ("the PLAIN block loop (plain_block_loop above): the slab-distributed loop has no split polynomial preconditioner "
                                                           "yet, so the model compares plain with plain -- the preconditioned single-GPU step of this line is faster than its N = 1 row")
        # ---- same simulation from the reference code's start vector (dkmc_set_current_warm_start(0)): what the default's warm start saves ----
        if args.warm_start == 1 and not args.no_alt:
            sim.warm_start = 0
            t0 = time.perf_counter(); it0 = 0; na = 2 if big else min(n, 5)
            for _ in range(na):
                sim.step(False); it0 += sim.host.get_stats()["cg_iters_X"]
            ta = time.perf_counter() - t0
            out["reference_start_vector"] = {"current_warm_start": 0, "steps": na, "value": round(na / ta, 5), "ms_per_step": round(ta / na * 1e3, 3), "cg_sweeps_X": it0 / na,
                                             "note": "same block-CG, started from gpubuf.atom_virtual_potentials as the reference code does (the buffer holds G0 x the previous "
                                                     "solution, current_solver_gpu.cu:1013-1016: practically a zero start)"}
            out["warm_start_gain"] = {"sweeps": round(it0 / na / max(res["per_step"]["cg_iters_X"], 1), 2), "steps_per_s": round(out["value"] / (na / ta), 2)}
            sim.warm_start = 1
        sites_main = sim.s.N
        sim.close()
        # ---- roofline.traffic: HBM bytes per launch of the dominant kernel from the PMC counters, measured now (the counter runs stop their
        # solves early on seconds-per-step workloads: the bytes one launch moves do not depend on the iteration count) ----
        if not args.no_pmc and "roofline" in out:
            tb, detail = pmc_traffic(name, out["roofline"]["kernel"], args.x_format, cg_tol=1e-3 if sites_main > 150000 else None, x_block=X_BLOCK)
            out["roofline"]["traffic"] = tb
            out["roofline"]["traffic_detail"] = detail
            if tb:
                out["roofline"]["traffic_over_algorithmic"] = round(tb / out["roofline"]["algorithmic_bytes_per_launch"], 3)
        # ================= the same workload in the reference's iterate order (single-vector CG) =================
        ref_iters = None
        if not args.no_reference_order and X_BLOCK > 1 and args.x_format:
            try:
                sr = Sim(name, devname, x_format=args.x_format, x_block=1, warm_start=0)
                elr, nr_ = sr.run(1 if big else min(n, 5), 1, budget_s=60.0)
                rr = summary(sr, elr, nr_)
                ref_iters = rr["per_step"]["cg_iters_X"]
                blk = {"x_block": 1, "current_warm_start": 0, "what": "dkmc_set_x_block(1) + dkmc_set_current_warm_start(0): solve_sparse_CG_Jacobi's iterate sequence from the reference code's start vector (iterative_solvers_gpu.cu:309-480)",
                       "steps": nr_, "warmup": 1, "value": round(nr_ / elr, 5), "ms_per_step": rr["ms_per_step"], "split_ms": rr["split_ms"],
                       "cg_sweeps_X": ref_iters, "cold_step": {"ms": round(sr.cold[0] * 1e3, 1), "cg_sweeps_X": sr.cold[1]} if sr.cold else None}
                blk.update({k: v for k, v in rooflines(sr).items() if k == "roofline"})
                if big:
                    blk["strong_scaling_model"] = strong_scaling_model(sr, rr["ms_per_step"])
                out["reference_order_cg"] = blk
                out["block_cg_gain"] = {"sweeps": round(ref_iters / max(res["per_step"]["cg_iters_X"], 1), 2),
                                        "steps_per_s": round(out["value"] / blk["value"], 2)}
                sr.close()
            except Exception as exc:
                out["reference_order_cg"] = {"error": repr(exc)[:300]}
        # ---- CPU leg: the oracle's CG iteration at this size x the reference algorithm's iteration count ----
        if not args.no_cpu_baseline:
            if sites_main > 150000:
                # (needs the shape of X: a short-lived simulation, one step at a loose tolerance)
                sc_ = Sim(name, devname, x_format=args.x_format, cg_tol=1e-2)
                sc_.run(1, 0)
                out["cpu_baseline"] = cpu_cg_baseline(sc_, ncpu, iters=ref_iters if ref_iters else res["per_step"]["cg_iters_X"],
                                                      iters_source=("the sweeps of the single-vector CG on the GPU (reference_order_cg)" if ref_iters
                                                                    else "the block-CG's sweep count (the single-vector count was not measured: an underestimate)"))
                sc_.close()
            else:
                s_, p_ = make_workload(name)
                out["cpu_baseline"] = cpu_superstep_baseline(s_, p_, ncpu)
            if out["cpu_baseline"]:
                out["gpu_over_cpu"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
        # ================= the reference's own 85 071-site device =================
        if not args.no_device and name != "7.5nm":
            try:
                d = Sim("7.5nm", devname, x_format=args.x_format)
                eld, nd = d.run(20, 3)
                rd = summary(d, eld, nd)
                blk = {"workload": "7.5nm", "sites": rd["sites"], "steps": nd, "warmup": 3, "value": round(nd / eld, 4), "unit": "KMC steps/s", "ms_per_step": rd["ms_per_step"],
                       "split_ms": rd["split_ms"], "per_step": rd["per_step"], "x_block": d.x_block,
                       "reference_cuda_log": {"s_per_superstep_median": 4.78, "source": "structures/single_devices/timing_7.5nm/output_noguess.txt (BASELINE.md; unnamed GPU)"}}
                blk.update(rooflines(d))
                if not args.no_alt:
                    d.warm_start = 0
                    t0 = time.perf_counter(); it0 = 0
                    for _ in range(5):
                        d.step(False); it0 += d.host.get_stats()["cg_iters_X"]
                    ta = time.perf_counter() - t0
                    blk["reference_start_vector"] = {"current_warm_start": 0, "value": round(5 / ta, 4), "ms_per_step": round(ta / 5 * 1e3, 3), "cg_sweeps_X": it0 / 5}
                    d.warm_start = 1
                if not args.no_cpp_host:
                    blk["cpp_host_crosscheck"] = cpp_host_crosscheck(d, 10, 2)
                if not args.no_cpu_baseline:
                    blk["cpu_baseline"] = cpu_superstep_baseline(d.s, d.p, ncpu)
                d.close()
                if not args.no_reference_order:
                    d1 = Sim("7.5nm", devname, x_format=args.x_format, x_block=1, warm_start=0)
                    el1, n1 = d1.run(10, 2)
                    r1 = summary(d1, el1, n1)
                    blk["reference_order_cg"] = {"x_block": 1, "value": round(n1 / el1, 4), "ms_per_step": r1["ms_per_step"], "cg_sweeps_X": r1["per_step"]["cg_iters_X"]}
                    blk["reference_order_cg"].update({k: v for k, v in rooflines(d1).items() if k == "roofline"})
                    d1.close()
                if not args.no_log_tolerance and args.cg_tol is None:
                    blk["at_log_tolerance"] = log_tolerance_block(devname, args.x_format, 10, 2)
                out["device_7p5nm"] = blk
            except Exception as exc:
                out["device_7p5nm"] = {"error": repr(exc)[:300]}
        # ================= scale points =================
        sp_names = (args.scale_points if args.scale_points is not None else ("tile:5,crossbar_10nm_5pitch,tile:20:nocurrent" if args.workload is None else "none"))
        points = {}
        for spn in [x for x in sp_names.split(",") if x and x != "none"]:
            try:
                if spn == "crossbar_10nm_5pitch":
                    points[spn] = crossbar_block(devname)
                    continue
                nocur = spn.endswith(":nocurrent")
                wname = spn[:-len(":nocurrent")] if nocur else spn
                sp = Sim(wname, devname, x_format=args.x_format, solve_current=False if nocur else None)
                el, ns_ = sp.run(3, 1, budget_s=60.0)
                r = summary(sp, el, ns_)
                r["steps"] = ns_; r["warmup"] = 1
                if sp.cold:
                    r["cold_step"] = {"ms": round(sp.cold[0] * 1e3, 1), "cg_sweeps_X": sp.cold[1]}
                if nocur:
                    for kx in ("cg_iters_X", "X_nnz", "tunnelling_set"):          # no current solve in these steps
                        r["per_step"].pop(kx, None)
                    r["what"] = ("a crossbar-SIZED stack with the current solve off, as every shipped crossbar parameter set runs "
                                 "(structures/crossbars/*/parameters.txt: solve_current = 0): charge + potential (K-CG + pair sum) + event loop")
                    r["us_per_executed_event"] = round(r["split_ms"]["rates"] * 1e3 / max(r["per_step"]["events"], 1), 1)
                    if not args.no_scaling_model:
                        r["k_cg_slab_model"] = k_slab_model(sp)
                rf = rooflines(sp)
                if nocur:                                # the library's X statistics are those of an earlier simulation of this process
                    rf.pop("roofline", None)
                    if "cold_step" in r: r["cold_step"].pop("cg_sweeps_X", None)
                r.update(rf)
                points[spn] = r
                sp.close()
            except Exception as exc:                 # a failed scale point must not void the main measurement
                points[spn] = {"error": repr(exc)[:300]}
        if points:
            out["scale_points"] = points
    else:
        name = args.workload or "tile:10"
        replicas = None
        if args.mode == "sharded" and not args.no_replicas:
            # weak scaling on the side: independent replicas of the 85 k-site device (own KMC random streams), aggregate steps/s
            rs = Sim("7.5nm", devname, kmc_seed=parallel.replica_kmc_seed(1, rank))
            el, n = rs.run(5, 1, barrier=parallel.barrier)
            el = parallel.max_over_ranks(el, red_dev)
            replicas = {"workload": "7.5nm", "steps_per_rank": n, "value": round(parallel.aggregate_rate(n, world, el), 4),
                        "unit": "KMC steps/s (aggregate over %d independent replicas)" % world, "scaling": "weak", "ms_per_step": round(el / n * 1e3, 3)}
            rs.close()
        single = None
        if args.mode == "sharded" and not args.no_single_ref:
            # the same simulation on ONE GPU of this node (rank 0; the other ranks wait): the reference point of the strong scaling
            if rank == 0:
                s1 = Sim(name, devname, x_format=1)
                big1 = s1.s.N > 150000
                el1, n1 = s1.run(2 if big1 else 5, 0 if big1 else 1, budget_s=90.0)
                single = {"workload": name, "n_gpus": 1, "steps": n1, "value": round(n1 / el1, 5), "ms_per_step": round(el1 / n1 * 1e3, 3),
                          "cg_iters_X": s1.cnt["cg_iters_X"] / max(s1.cnt["steps"], 1), "trace": [list(t) for t in s1.trace[:n1]]}
                s1.close()
            parallel.barrier()
        transport_note = None
        if args.mode == "sharded":
            # every rank advances the same simulation (same seeds); X is sharded.  RCCL over xGMI when the process group is nccl.  Should
            # the in-library communicator fail to come up on ANY rank (no multi-GPU node was available to try it on), all ranks fall back
            # together to the host-callback transport over a gloo group: slower exchanges, same results -- and the line says so.
            # (Symmetric failures only: ncclCommInitRank is a collective, a single rank that throws before entering it leaves the others
            # inside it -- the watchdog then ends the run with exit code 3.)
            import torch.distributed as dist
            ok, why = 1.0, ""
            try:
                if os.environ.get("DKMC_BENCH_FORCE_COMM_FAIL"):
                    raise RuntimeError("DKMC_BENCH_FORCE_COMM_FAIL")
                transport = parallel.attach_solver_comm()
            except Exception as exc:
                ok, why = 0.0, repr(exc)[:200]
            tok = torch.tensor([ok], dtype=torch.float64, device=red_dev)
            dist.all_reduce(tok, op=dist.ReduceOp.MIN)
            if tok.item() == 0.0:
                try:
                    parallel.detach_solver_comm()
                except Exception:
                    pass
                transport = parallel.attach_solver_comm("host", group=dist.new_group(backend="gloo"))
                transport_note = "in-library RCCL communicator failed to attach (%s); exchanges go through pinned host memory + gloo" % (why or "on a peer rank")
            sim = Sim(name, devname, x_format=1)
            # opt-in: the one-shot peer-write exchange for the block loop's sweeps (hipIpc-mapped buffers, push + signal + bounded wait;
            # validated with two processes on ONE GPU only, hence not the default beside RCCL).  All ranks decide together.
            peer_on = False
            if args.peer_exchange or os.environ.get("DKMC_PEER_EXCHANGE") == "1":
                sim.L.dkmc_set_x_slab(0)          # the peer-write exchange carries the slots of the all-gather variant of the sharded block loop
                peer_on = parallel.attach_peer_exchange(16 * (int(sim.dev.N_atom) + 2) + 2)
        else:
            transport = "none"; peer_on = False
            sim = Sim(name, devname, kmc_seed=parallel.replica_kmc_seed(1, rank))
        big = sim.s.N > 150000
        warm = args.warmup if not big else min(args.warmup, 1)
        elapsed, n = sim.run(args.steps, warm, budget_s=args.budget if big else None, barrier=parallel.barrier,
                             agree=(lambda v: parallel.from_rank0(v, red_dev)) if args.mode == "sharded" else None)
        # every rank must have timed the same number of steps (the budget rule uses local clocks)
        import torch.distributed as dist
        tn = torch.tensor([float(n)], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tn, op=dist.ReduceOp.MIN)
        if int(tn.item()) != n:
            raise SystemExit("ranks timed different step counts (%d vs %d): rerun with a smaller --steps" % (n, int(tn.item())))
        elapsed = parallel.max_over_ranks(elapsed, red_dev)
        res = summary(sim, elapsed, n)
        st = sim.host.get_stats()
        share = st["xt_local_subblocks"] / max(st["xt_subblocks"], 1)
        roofs = rooflines(sim, local_share=share if args.mode == "sharded" else 1.0)
        traces = [None] * world
        dist.all_gather_object(traces, sim.trace)
        # how many ranks the in-library communicator of the sharded solve actually holds (dkmc_comm_info), from every rank
        import ctypes as _C
        _n, _r, _t = _C.c_int(0), _C.c_int(0), _C.c_int(0)
        sim.L.dkmc_comm_info(_C.byref(_n), _C.byref(_r), _C.byref(_t))
        infos = [None] * world
        dist.all_gather_object(infos, (_n.value, _r.value, _t.value))
        agree = all(t == traces[0] for t in traces)
        if rank == 0:
            sharded = args.mode == "sharded"
            out = {
                "metric": "KMC steps/sec", "value": round(n / elapsed if sharded else parallel.aggregate_rate(n, world, elapsed), 5), "unit": "KMC steps/s",
                "n_gpus": world, "steps": n, "warmup": warm, "ms_per_step": res["ms_per_step"],
                "higher_is_better": True, "scaling": "strong" if sharded else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "config": {"workload": name, "sites": res["sites"], "nn": res["nn"], "atoms": res["atoms"], "Vd": VD,
                           "phases": "charge+potential+rates+current+heat",
                           "parallelism": (("one simulation; X generated/stored/streamed in %d per-rank shares; block-CG state distributed by lateral row slabs (csrc/xtb_slab.inc): per sweep the "
                                            "tile sums of the S rows go to their owners (all-to-all-v), 6 x 256 Gram entries are all-gathered and added in rank order, own rows of QS + halo rows of P go out (all-to-all-v)")
                                           if sharded and not peer_on else
                                           "one simulation; X generated/stored/streamed in %d per-rank shares, one exchange of 16 |S| + 2 doubles per block-CG sweep, slots added in rank order (all-gather variant)"
                                           if sharded else "replicas x%d") % world,
                           "x_slab": int(sim.L.dkmc_get_x_slab()) if sharded else None,
                           "x_poly": ("none: the sharded / slab-distributed block loops do not carry the split polynomial preconditioner of the one-GPU loop yet (dkmc_set_x_poly, "
                                      "degree 8 there, ~3.5x fewer seconds per step at 9.4e5 sites) -- an N-rank run of this round is compared with the PLAIN one-GPU loop, "
                                      "not with the N = 1 line of bench.py") if sharded else None,
                           "exchange": ("peer-write over hipIpc-mapped buffers (opt-in, all-gather variant)" if peer_on else "all-to-all-v + all-gather of the transport (slab-distributed block-CG)") if sharded else None,
                           "transport": transport, "transport_note": transport_note, "x_format": "tiled", "current_warm_start": sim.warm_start, "cg_tol": sim.p.cg_tol,
                           "comm_ranks": min(i[0] for i in infos), "comm_rank_ids": sorted(i[1] for i in infos),
                           "comm_transport_code": sorted(set(i[2] for i in infos)), "process_group_world": world, "process_group_backend": backend},
                "split_ms": res["split_ms"], "per_step": res["per_step"], "cpu_baseline": None,
                "sharding": {"ranks_agree_bitwise": bool(agree) if sharded else None, "subblocks_total": int(st["xt_subblocks"]),
                             "subblocks_rank0": int(st["xt_local_subblocks"]), "rank0_share": round(share, 4),
                             "exchange_us": round(sim.prof["comm_ms"] / max(sim.prof["comm_n"], 1) * 1e3, 2),
                             "exchanged_doubles": int(st["comm_count_per_rank"]),
                             "peer_exchange": parallel.peer_exchange_info() if peer_on else None,
                             "replicated_phases": "charge, K-CG, event loop, assembly of the neighbour part of X; pair sum: site slabs + all-gather"},
                "replicas": replicas, "single_gpu_reference": None,
            }
            if single is not None:
                # same seeds, same steps: the sharded run must reproduce the single-GPU trajectory (to rounding)
                ref_tr = single.pop("trace")           # steps 0, 1, ... of the single-GPU run; sim.trace holds steps warm, warm + 1, ... of the sharded one
                k = min(len(ref_tr) - warm, len(sim.trace))
                dev_rel = None
                if k > 0:
                    dev_rel = max(abs(a - b) / max(abs(b), 1e-300) for jj in range(k) for a, b in zip(sim.trace[jj], ref_tr[warm + jj]))
                single["common_steps"] = max(k, 0)
                single["max_rel_deviation_of_dt_I_T_over_common_steps"] = dev_rel
                out["single_gpu_reference"] = single
                out["strong_scaling_speedup"] = round(out["value"] / single["value"], 3)
            if n != args.steps:
                out["steps_requested"] = args.steps
            out.update(roofs)
            if sharded and not agree:
                out["error"] = "ranks disagree"
            if sharded and min(i[0] for i in infos) != world:
                out["error"] = "the solver communicator holds %d ranks, the process group %d" % (min(i[0] for i in infos), world)
        if args.mode == "sharded":
            parallel.detach_solver_comm()
        sim.close()

    wd.cancel()
    if rank == 0:
        print(json.dumps(out), flush=True)
    parallel.finalize()
    if out is not None and out.get("error"):
        sys.exit(4)


if __name__ == "__main__":
    main()
