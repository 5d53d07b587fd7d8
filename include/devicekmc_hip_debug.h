/*
 * devicekmc_hip_debug.h -- test and measurement aids exported by libdevicekmc_hip.so.  NOT part of the drop-in boundary
 * (include/devicekmc_hip.h): nothing here replaces a reference symbol; tests/ and bench.py bind them through devicekmc_amd/lib.py.
 */
#ifndef DEVICEKMC_HIP_DEBUG_H
#define DEVICEKMC_HIP_DEBUG_H
#include "devicekmc_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* measurement aid (bench.py's strong-scaling model): on the tiled X left resident by the last single-GPU update_power, the time per CG
 * iteration of what ONE rank of an nranks-way sharded solve runs -- apply_us: the apply kernel over that rank's share of the tiles
 * (work items sized as an nranks run sizes them) + the neighbour part; side_us[4]: partial row sums, finish, vector step, and -- nranks > 1 or a multi-GB sweep, where apply_us is the tile pass alone -- the
 * neighbour part that a sharded solve runs on a second stream beside the exchange (each timed on its own).  The
 * all-reduce between them cannot be measured on one GPU.  Scratch vectors are overwritten; results of the last solve already
 * delivered (potentials, I_macro, power) are not. */
int dkmc_xt_time_share(int nranks, int rank, int reps, double *apply_us, double *side_us /* [4] */, int *items_out, long long *subblocks_out);
/* Test aid: emulates on ONE GPU the tile pass of an nranks-way sharded matrix-vector product over the X of the last single-GPU solve
 * (every rank's work items built as a sharded assembly builds them, partial arrays zeroed per rank, partial row sums restricted to
 * the rank's windows) and compares the sum of the ranks' results with the one-GPU pass.  subblocks_sum / items_sum: totals over the
 * shares (must equal the stored sub-blocks / items_total: every tile in exactly one share). */
int dkmc_xt_check_shares(int nranks, double *max_abs_diff, double *max_abs, long long *subblocks_sum, long long *items_sum, int *items_total);
/* Test aid: on the X left resident by the last single-GPU solve, the MFMA tile x panel product of the block-CG (16 test vectors, one
 * sweep) against 16 passes of the single-vector tile kernel; largest absolute deviation and largest sum over the S rows. */
int dkmc_xtb_check_product(int width, double *max_abs_diff, double *max_abs);
/* Measurement aid: average duration [us] of the tile x panel kernel of the block-CG over the X left resident by the last single-GPU solve
 * (`reps` launches).  variant 0: as a solve runs it; on the round-4 form of the loop: 1: without its matrix instructions (tile stream + LDS
 * traffic); 2: without re-reading the tile stream (matrix instructions + LDS traffic); 3: operand stages of one k-pair; 4: without LDS
 * traffic; 7: the matrix instructions alone; 10: the product form without re-reading the tile stream; 12: the product form with the partial
 * tiles skipped.  Variants other than 0 exist only in a library built with DKMC_MEASURE_VARIANTS=1 (python __graft_entry__.py); the shipped one
 * returns error 13 for them. */
int dkmc_xtb_time_apply(int width, int variant, int reps, double *us);
/* Measurement aid: how the resident X fills its tiles -- hist[c] = tiles with c of their 8 sub-blocks present (c = 0 .. 8), hist[9] = tiles inside a
 * chain (>= 2) of full tiles of a run, hist[10] = runs.  hist: 11 entries. */
int dkmc_xt_tile_census(long long *hist);
/* Measurement aid for the run list of the tile kernels (takes effect at the next assembly of X): kc > 0 overrides the nominal run length in tiles
 * (default min(32, tiles / ranks / 4096)); 0 restores it. */
void dkmc_set_x_items(int kc);
/* Same-box comparison aid: 1 = the solves (and variant 0 above) run the round-4 form of the tile x panel loop (stages issued in bursts,
 * conditional loads at the tile end, panel rows loaded directly) instead of the product form; same results bit for bit.  Default 0. */
void dkmc_set_x_apply_form(int form);
int dkmc_get_x_apply_form(void);
/* Test aid for the error path of a sharded current solve (no counterpart in the reference): the calling rank fails ONCE, in the
 * assembly of X (phase 1) or on the host side of CG iteration `iteration` (phase 2).  Every rank's dkmc_update_power_gpu_sparse then
 * returns non-zero (the failing rank its own code, the others 46) instead of blocking in a collective: the ranks agree on the
 * outcome of the local set-up before the first collective, and inside the loop an abort word travels with every all-reduce. */
void dkmc_debug_inject_fault(int phase, int iteration);
/* test aid: one launch of the CG step kernel of iteration `it` over m elements with the stop word preset to done_word; *updated = elements of y it
 * changed (0 / it + 2: all; 1 ... it + 1: none), *done_after = the stop word afterwards (csrc/xt.hip: k_xt_step's iteration-stamped stop word) */
int dkmc_debug_step_stop_word(int m, int it, int done_word, int *updated, int *done_after);

/* Test / measurement aid: the slab-distributed block-CG (csrc/xtb_slab.inc) with nranks VIRTUAL ranks inside this process, on the X left resident
 * by the last single-GPU solve: the same system solved by the one-GPU block-CG and by the distributed loop (shares of the tiles as a sharded
 * assembly builds them, rows owned by lateral slabs, exchanges as device copies), both from a zero start to `tol`.  rel_diff: largest deviation of
 * the two solutions / largest entry; times_us[8]: mean kernel times of virtual rank time_rank (apply, neighbour part, fold, rows, Gram reduction,
 * s x s algebra, step, pack + unpack); xdoubles[3]: doubles a rank receives per sweep in the three exchanges; rows_min_max[2]: rows of the smallest
 * and largest slab.  sweep_cap > 0: measurement run -- the distributed loop stops after that many sweeps, the one-GPU solve is skipped, rel_diff = -1.  Fails if the virtual ranks leave the loop at different sweeps or end with different bits. */
int dkmc_xtb_emulate_slabs(int nranks, int width, double tol, int time_rank, int sweep_cap, double *rel_diff, int *iters_slab, int *iters_ref,
                           double *times_us, long long *xdoubles, int *rows_min_max);

/* Test / measurement aid: the slab-distributed CG on K (csrc/kcg.hip) with nranks VIRTUAL ranks inside this process: the background-potential
 * system of the buffer's current state solved from the buffer's current potential by the one-GPU reference-order loop and by the distributed loop
 * (exchanges as device copies), into scratch copies.  max_abs_diff [V]; times_us[4]: product, update, direction, halo pack + unpack of virtual
 * rank time_rank; halo_rows[2]: doubles received per iteration in the halo exchange (largest over the ranks), rows of the largest slab.
 * iter_cap > 0: measurement run (the distributed loop stops after that many iterations, max_abs_diff = -1). */
int dkmc_kcg_emulate_slabs(dkmc_gpubuf *buf, int N, int N_left, int N_right, double Vd, double high_G, double low_G, int num_metals, int nranks, int time_rank,
                           int iter_cap, double *max_abs_diff, int *iters_slab, int *iters_ref, double *times_us, long long *halo_rows);

#ifdef __cplusplus
}
#endif
#endif
