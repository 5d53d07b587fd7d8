/*
 * devicekmc_hip.h -- C ABI of the MI355X (gfx950) KMC-superstep engine.
 *
 * Drop-in boundary for DeviceKMC's GPU path: every entry point below replaces one symbol of the
 * reference's gpu_solvers.h / gpu_buffers.h surface (cited per function, paths relative to
 * /root/reference/src).  Plain C types only: device pointers, sizes and scalars.  The C++ shim
 * include/gpu_solvers.h + include/gpu_buffers.h re-exposes the reference's own names and
 * signatures (GPUBuffers&, std::vector, RandomNumberGenerator&) on top of this header.
 *
 * Conventions
 *  - all pointers named d_* are DEVICE pointers; h_* are host pointers;
 *  - ELEMENT / EVENTTYPE are 4-byte ints with the reference's enum values (utils.h:37-60);
 *  - one process drives one GPU; all work is enqueued on one HIP stream (dkmc_set_stream);
 *  - functions return 0 on success, non-zero HIP/engine error code otherwise; the message is
 *    available from dkmc_last_error().  (The reference prints CUDA errors and carries on,
 *    utils.h:145-153; the shim does the same with the message.)
 */
#ifndef DEVICEKMC_HIP_H
#define DEVICEKMC_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

/* Mirror of the public members of class GPUBuffers (gpu_buffers.h:16-52). */
/* Element arrays hold the 4-byte ELEMENT enum (utils.h:37-44).  A C++ host that has the enum type defines DKMC_ELEMENT_T to it before
 * including this header (include/gpu_buffers.h does): the members then have the reference's own type, `ELEMENT *`, and the
 * reference's call sites (potential_solver.cpp:152, KMCProcess.cpp:269-274) compile unchanged.  Same layout either way. */
#ifndef DKMC_ELEMENT_T
#define DKMC_ELEMENT_T int
#endif
typedef struct dkmc_gpubuf {
    int *site_charge;
    double *site_power, *site_potential_boundary, *site_potential_charge, *site_temperature;
    double *site_CB_edge;
    double *T_bg;
    double *atom_power;
    double *atom_CB_edge;
    double *atom_virtual_potentials;
    int *atom_charge;
    DKMC_ELEMENT_T *site_element;
    DKMC_ELEMENT_T *atom_element;
    double *site_x, *site_y, *site_z;
    double *atom_x, *atom_y, *atom_z;
    DKMC_ELEMENT_T *metal_types;
    double *sigma, *k, *lattice, *freq;
    int *neigh_idx, *site_layer;
    int *Device_row_ptr_d, *Device_col_indices_d;
    int *contact_left_row_ptr, *contact_left_col_indices;
    int *contact_right_row_ptr, *contact_right_col_indices;
    int Device_nnz, contact_left_nnz, contact_right_nnz;
    int num_metal_types_, N_, nn_, N_atom_;
} dkmc_gpubuf;

/* Per-call statistics of the last solver invocations (no reference twin; the reference prints
 * "# CG steps" to stdout, iterative_solvers_gpu.cu:456). */
typedef struct dkmc_stats {
    int cg_iters_K, cg_iters_CB, cg_iters_X;
    double cg_rr_K, cg_rr_CB, cg_rr_X;     /* final ||r||^2 of the scaled system */
    int n_events;                          /* events executed by the last execute_kmc_step */
    int n_charged;                         /* charged sites seen by the last poisson_gridless */
    int N_atom;                            /* atoms found by the last update_power */
    long long X_nnz;
    double psum_last;                      /* sum of rates before the last executed event */
    /* kernel profile (dkmc_set_profiling(1)): HIP-event time of the SpMV launches (A*p) of the last CG solve
     * on X, split into the wave-per-row kernel (long rows) and the 16-lanes-per-row kernel (short rows) */
    double spmv_long_ms, spmv_short_ms;
    int spmv_long_launches, spmv_short_launches;
    long long spmv_long_nnz, spmv_short_nnz;
    int spmv_long_rows, spmv_short_rows;
    /* dense-run mode: k_spmv_segs work items and the matrix entries they cover (spmv_long_ms then times that kernel alone,
     * spmv_short_ms the row kernel that follows it) */
    int spmv_segments, spmv_pad;
    long long spmv_segment_entries;
    /* sharded current solve (dkmc_comm_*): ranks, the segments this rank multiplied in the last solve, doubles per rank in
     * the all-gather (row sums of the owned long rows, padded to the largest share), and (profiling on) the HIP-event time
     * of the sampled exchange steps (row-sum kernel + all-gather) */
    int comm_ranks, comm_local_segments;
    long long comm_count_per_rank;
    double comm_ms;
    int comm_launches, comm_pad;
    /* tiled X: 32 x 256 tiles of the tunnelling block (read once per iteration for both triangles) and the entries of the
     * upper triangle they hold */
    int spmv_tiles, spmv_pad2;
    long long spmv_tile_entries;
    /* tiled X (dkmc_set_x_format(1), the default): stored 32 x 32 sub-blocks of the tunnelling block (8 KiB each; all ranks /
     * this rank), work items of the apply kernel and tiles per item, entries of the neighbour part Xs, size of the tunnelling set;
     * spmv_tiles / spmv_tile_entries then count the 32 x 256 tiles and the entries of the upper triangle they hold */
    long long xt_subblocks, xt_local_subblocks;
    int xt_items, xt_kc;
    long long xt_sparse_nnz;
    int xt_ns, xt_split_launch;           /* 1: sharded solve, the neighbour part of A*p runs on a second stream beside the exchange */
    /* profiling on: HIP-event time of the whole iteration loop of the last CG solve on K (and the iterations it covers), and
     * of the last pair-sum kernel */
    double kcg_ms;
    int kcg_iters_timed, kcg_blocked;      /* kcg_blocked: 1 if the last K solve ran on the blocked form of the pattern (dkmc_set_k_blocked) */
    double pair_ms;
    long long pair_evaluated;              /* (site, charged site) pairs inside the screening cut-off of the last pair sum (profiling on) */
    long long pair_tested;                 /* pairs whose distance was tested (all N x N_charged without the cell list; the 3 x 3 columns with it) */
    long long xt_records;                  /* records of column partial sums one matrix-vector product writes (= runs; runs / 4 on one GPU, where the four waves of a workgroup share one) */
    long long tcache_bytes;                /* bytes of tunnelling-coefficient cache THIS rank holds (sharded solve on the tiled X: what its tiles read) */
    long long kcg_bytes;                   /* bytes one iteration of the last K solve moves: column/class words + the q windows (or one read of q per row) + 14 (15) vector touches */
    int xb_aux, xb_pad;                    /* 1: the last block solve used the smooth auxiliary columns (dkmc_set_x_aux) */
    int xb_width, xb_fallback;             /* block-CG width of the last current solve (1 = single-vector loop); 1 if the block loop lost definiteness and the single-vector loop finished the solve */
} dkmc_stats;

const char *dkmc_last_error(void);
void dkmc_clear_error(void);
const dkmc_stats *dkmc_get_stats(void);

/* kmc_events.cu:15-32 */
int dkmc_get_gpu_info(char *gpu_string, int capacity, int dev);
int dkmc_set_gpu(int dev);
/* stream all kernels are enqueued on (hipStream_t); NULL = the null stream (reference behaviour) */
int dkmc_set_stream(void *hip_stream);
int dkmc_synchronize(void);
/* tolerance of solve_sparse_CG_Jacobi; the reference hard-codes 1e-6 (iterative_solvers_gpu.cu:322) */
void dkmc_set_cg_tolerance(double tol);
/* Domain of the system update_CB_edge_gpu_sparse solves.  0 (default): every site, as potential_solver_gpu.cu:595-694 of the reference
 * snapshot does.  1: atoms only -- links to interstitial sites (DEFECT, OXYGEN_DEFECT) are left out and those sites get CB edge 0.  The
 * reference's own artefacts (X pattern dump timing_2.5nm/fullmatrix_assembly, the 19 currents of timing_7.5nm/output_noguess.txt) were
 * written by a revision that solved on atoms (its host twin still shows `gesv(.., &N_atom, ..)` as a comment, potential_solver.cpp:98):
 * with 1 they are reproduced to every entry / printed digit, with 0 the current is 0.83 % lower. */
void dkmc_set_cb_edge_domain(int atoms_only);
/* Memory the tunnelling-coefficient cache of the current solve may take (one row per vacancy site x one column per inner-contact
 * metal, 8 B each: 27 GB at 9.4e5 sites).  -1 (default): a third of the device memory that is free when the cache is (re)built, between
 * 8 and 128 GiB.  When one row per vacancy does not fit, the contact->trap integrals are evaluated directly while the tiles are filled,
 * every step -- same function, same bits, slower assembly.  Takes effect at the next rebuild of the cache (new bias point / structure). */
void dkmc_set_tcache_budget(long long bytes);
/* Screening cut-off of poisson_gridless_gpu.  A term of the pair sum is q erfc(x) k Q / r with x = r / (sigma sqrt 2); terms beyond
 * x_cut are not evaluated.  Default 6.5 (erfc < 3.8e-20: all omitted terms of a 1e6-site stack together stay below 2e-17 V, under the
 * rounding of the sum and under the last-bit noise of the reference's atomicAdd order).  0: every pair, exactly the terms the reference
 * sums (potential_solver_gpu.cu:908-958) -- for a parity run. */
void dkmc_set_pair_cutoff(double x_cut);
/* The pair sum keeps the grouping of the sites by (y, z) column between calls, keyed on the position / lattice arrays, N, pbc and the
 * cut-off; freeing or re-initialising a GPUBuffers drops it.  A host that rewrites site positions IN PLACE behind the same device pointers
 * (only possible through the raw-pointer entry dkmc_poisson_gridless_gpu; the reference never moves a site) calls this afterwards. */
void dkmc_reset_pair_sum_cache(void);
/* Width s of the block-CG of the current solve on the tiled X (csrc/xtb.hip).  1: the reference's single-vector loop
 * (solve_sparse_CG_Jacobi, iterative_solvers_gpu.cu:309-480: same iterate sequence, same start vector).  2 ... 16: block-CG over s
 * columns -- column 0 carries the physical right-hand side and start vector, columns 1 ... s - 1 fixed-seed auxiliary right-hand sides
 * that only enlarge the Krylov space; every 32 x 32 sub-block of the tunnelling block is streamed ONCE per sweep and multiplied (it and
 * its transpose) into 32 x 16 panels by v_mfma_f64_16x16x4_f64 -- the contraction the reference's unfinished split path applies as a
 * GEMV (add_submatrix_product, iterative_solvers_gpu.cu:634-652).  Same stop test on column 0 (||r||^2 <= tol^2); the solution agrees
 * with the single-vector loop to the stop tolerance, NOT iterate by iterate.  85 071 sites: 666 -> 208 / 133 / 95 sweeps at s = 4 / 8 / 16. */
void dkmc_set_x_block(int s);
int dkmc_get_x_block(void);
/* More than one rank (dkmc_comm_*): 1 (default) distributes the STATE of the block-CG by spatial row slabs -- a rank owns the rows of one lateral
 * slab of the device: neighbour part, Gram pass and panel updates run on its rows only; per sweep the tile sums of the S rows go to their owners
 * (all-to-all-v), 6 x 256 Gram entries are all-gathered and added in rank order, the own rows of Q = S P and the halo rows of P go out
 * (all-to-all-v) (csrc/xtb_slab.inc).  0: the all-gather variant -- only the tile stream is sharded, everything else replicated. */
/* Split polynomial preconditioner of the block-CG of the current solve (one GPU; csrc/xtb.hip): degree d > 0 runs the block loop on L A L with
 * L = the degree-d truncation of the series of (I - N)^(-1/2), N = I - (neighbour part + diagonal of the Jacobi-scaled X).  A sweep then costs 2 d
 * more sparse panel products and the loop needs 2-3x fewer sweeps (tools/precond_block_proto.py); the start vector enters through the right-hand
 * side L (b - A y0), the result meets the reference's stop test in the TRUE residual (checked, the loop is re-entered if it does not).  0 = off (default). */
void dkmc_set_x_poly(int degree);
int dkmc_get_x_poly(void);
void dkmc_set_x_slab(int on);
int dkmc_get_x_slab(void);
/* More than one rank, K system above the size of the blocked form (262 144 rows): 1 (default) distributes the CG on K (background potential, CB
 * edge) by the same lateral row slabs -- product, update and direction on the own rows, the two dot products of an iteration completed by
 * all-gathers of block partials added in one fixed order on every rank, the halo of the scaled direction by an all-to-all-v (csrc/kcg.hip;
 * SURVEY 8e row "K-CG").  0: the solve is replicated on every rank. */
void dkmc_set_k_slab(int on);
int dkmc_get_k_slab(void);
/* Auxiliary right-hand sides of the block-CG (a free choice: only their block Krylov space matters; column 0 is always the physical system).
 * 0: fixed-seed hash of (row, column), uniform in [-1, 1).  1: smooth set, column v = cos(v pi xi) / s with xi the atom's x coordinate scaled to
 * [0, 1] -- rich in the low modes of the neighbour part of X: 20-35 % fewer sweeps at the default tolerance.  2 (default): the smooth set at
 * tolerances of 1e-8 and looser, the hash set below (smooth systems converge before the physical column does and the s x s systems then lose
 * rank close to a converged tolerance).  No counterpart in the reference. */
void dkmc_set_x_aux(int mode);
int dkmc_get_x_aux(void);
/* With dkmc_set_current_warm_start(1).  1: the auxiliary columns of the hash set start from the solutions the previous solve left for them (their
 * right-hand sides never change), and at tolerances of 1e-8 and looser the set is half smooth (zero start) + half hash (warm): the residuals the
 * warm columns start with carry what the previous solve had not resolved yet.  0 (default): every auxiliary column starts from zero.  On the
 * oracle's X of the 2.5 nm device the warm auxiliary start halves the sweeps (33 -> 16-19, tools/warm_aux_proto.py); measured on the GPU it gains
 * nothing from 85 k sites up (profiles/r05_ab_aux_warm.json: 9.4e5 sites 435 -> 425-445 sweeps), hence off.  Column 0 and the contract
 * (solution within the stop test) are unaffected either way. */
void dkmc_set_x_aux_warm(int on);
int dkmc_get_x_aux_warm(void);
/* 1 (default): initialize_sparsity also builds the blocked form of the K pattern (csrc/kcg.hip) for systems of up to 262 144 device rows:
 * the CG on K (background potential, CB edge: solve_sparse_CG_Jacobi on K, iterative_solvers_gpu.cu:309-480) then runs in an internal
 * x-sorted row order, one block of rows per CU with its window of the direction vector in LDS; site order outside the solve is untouched.
 * 0: the solve uses the CSR positions of the pattern (the only form above that size).  Read when a pattern is built. */
void dkmc_set_k_blocked(int on);
int dkmc_get_k_blocked(void);
/* Start vector of the current solve.  1 (default): the previous step's solution, from a private unscaled copy kept per GPUBuffers -- what
 * the reference's own comment asks for ("use the previous solution as the initial guess", current_solver_gpu.cu:976-977).  0: whatever
 * gpubuf.atom_virtual_potentials holds, exactly as the reference code does -- and that buffer was scaled by G0 in place after the previous
 * solve (current_solver_gpu.cu:1013-1016), so the reference starts from G0*m, i.e. practically from zero: the reference-order switch
 * (with dkmc_set_x_block(1): the reference's iterate sequence).  Either way the contract is the solution within the stop test; the first
 * solve of a run starts from the buffer in both modes.  dkmc_get/set_current_warm_vector export and restore the private copy (restart). */
void dkmc_set_current_warm_start(int mode);
int dkmc_get_current_warm_start(void);
int dkmc_get_current_warm_vector(const dkmc_gpubuf *buf, double *h_out, int capacity, int *n_out);
int dkmc_set_current_warm_vector(const dkmc_gpubuf *buf, const double *h_in, int n);
/* the same for the solutions of the block-CG's auxiliary columns (dkmc_set_x_aux_warm): [rows][16] doubles */
int dkmc_get_current_warm_aux(const dkmc_gpubuf *buf, double *h_out, long long capacity, long long *n_out);
int dkmc_set_current_warm_aux(const dkmc_gpubuf *buf, const double *h_in, long long n);
/* 1: bracket every SpMV launch of the CG solves with HIP events on the engine's stream and accumulate
 * their durations into dkmc_stats (measurement aid for bench.py; off by default) */
void dkmc_set_profiling(int on);
/* layout of the current-solve matrix X (update_power_gpu_sparse).  1 (default): tiled X -- the neighbour part as a small CSR,
 * the tunnelling block generated straight into symmetric 32 x 256 tiles (upper triangle only, no column indices, one copy);
 * per-rank storage and assembly when a communicator is attached.  0: CSR X exactly as the reference stores it
 * (Assemble_X_sparsity / Assemble_X2), solved by reading every stored entry; both agree to rounding.  dkmc_get_last_X
 * returns the same column-sorted CSR in both modes. */
void dkmc_set_x_format(int tiled);
int dkmc_get_x_format(void);

/* ---- GPUBuffers (gpu_buffers.h:73-158, gpu_buffers.cpp:10-118) ---------------------------- */
/* allocates every array of the struct with hipMalloc and uploads the constant ones */
int dkmc_gpubuf_create(dkmc_gpubuf *buf, int N, int N_atom, int nn, int num_metal_types,
                       const int *h_site_layer, const double *h_site_x, const double *h_site_y,
                       const double *h_site_z, const int *h_neigh_idx, const int *h_metals,
                       double freq, double sigma, double k, const double *h_lattice);
int dkmc_gpubuf_free(dkmc_gpubuf *buf);
/* sync_HostToGPU / sync_GPUToHost: the nine arrays of gpu_buffers.cpp:10-55 */
int dkmc_gpubuf_sync_host_to_gpu(dkmc_gpubuf *buf, const int *h_site_element, const int *h_site_charge,
                                 const double *h_site_power, const double *h_site_CB_edge,
                                 const double *h_site_potential_boundary, const double *h_site_potential_charge,
                                 const double *h_site_temperature, const double *h_atom_CB_edge, double T_bg);
int dkmc_gpubuf_sync_gpu_to_host(const dkmc_gpubuf *buf, int *h_site_element, int *h_site_charge,
                                 double *h_site_power, double *h_site_CB_edge,
                                 double *h_site_potential_boundary, double *h_site_potential_charge,
                                 double *h_site_temperature, double *h_atom_CB_edge, double *T_bg);
int dkmc_copy_power_from_gpu(const dkmc_gpubuf *buf, double *h_power);          /* gpu_buffers.cpp:58-72 */
int dkmc_copy_charge_to_gpu(dkmc_gpubuf *buf, const int *h_charge);             /* :88-93 */
int dkmc_copy_Tbg_to_gpu(dkmc_gpubuf *buf, double T_bg);                        /* :75-80 */

/* copytoConstMemory (kmc_events.cu:369-375): per-layer zero-field energies, at most 5 layers */
int dkmc_copy_to_const_memory(const double *h_E_gen, const double *h_E_rec, const double *h_E_Vdiff,
                              const double *h_E_Odiff, int num_layers);

/* ---- neighbour index: Device::constructSiteNeighborList + padding (Device.cpp:98-136, :69-80) ------- */
/* Row i of d_neigh_out[N * nn] holds every j != i with site_dist(i, j) < nn_dist, ascending, -1 padded; nn = global maximum.
 * Two calls: d_neigh_out == NULL computes nn into *nn_out; the second call fills d_neigh_out using nn = *nn_out. */
int dkmc_build_neighbor_index(int N, const double *d_x, const double *d_y, const double *d_z, const double *h_lattice,
                              int pbc, double nn_dist, int *nn_out, int *d_neigh_out);

/* ---- sparsity of K: initialize_sparsity (iterative_solvers_gpu.cu:96-109) ------------------- */
/* fills Device_*, contact_left_*, contact_right_* of buf (allocated by the library) */
int dkmc_initialize_sparsity(dkmc_gpubuf *buf, int pbc, double nn_dist, int num_atoms_contact);
/* frees the six pattern arrays initialize_sparsity allocated and the solver state kept for this buffer (dkmc_gpubuf_free does both
 * for buffers created by dkmc_gpubuf_create) */
int dkmc_free_sparsity(dkmc_gpubuf *buf);

/* ---- potential ------------------------------------------------------------------------------ */
/* update_charge_gpu (potential_solver_gpu.cu:54-63) */
int dkmc_update_charge_gpu(const int *d_site_element, int *d_site_charge, const int *d_neigh_idx,
                           int N, int nn, const int *d_metals, int num_metals);
/* update_CB_edge_gpu_sparse (potential_solver_gpu.cu:595-694) */
int dkmc_update_CB_edge_gpu_sparse(dkmc_gpubuf *buf, int N, int N_left_tot, int N_right_tot, double Vd,
                                   int pbc, double high_G, double low_G, double nn_dist, int num_metals);
/* background_potential_gpu_sparse (potential_solver_gpu.cu:696-781) */
int dkmc_background_potential_gpu_sparse(dkmc_gpubuf *buf, int N, int N_left_tot, int N_right_tot, double Vd,
                                         int pbc, double high_G, double low_G, double nn_dist,
                                         int num_metals, int kmc_step_count);
/* poisson_gridless_gpu (potential_solver_gpu.cu:960-978) */
int dkmc_poisson_gridless_gpu(int num_atoms_contact, int pbc, int N, const double *d_lattice,
                              const double *d_sigma, const double *d_k,
                              const double *d_posx, const double *d_posy, const double *d_posz,
                              const int *d_site_charge, double *d_site_potential_charge);

/* solve_sparse_CG_Jacobi (iterative_solvers_gpu.cu:309-480): solves A y = x; overwrites A_data
 * (scaled) and x (scaled rhs); y holds the warm start on entry and the solution on exit. */
int dkmc_solve_sparse_CG_Jacobi(double *d_A_data, const int *d_A_row_ptr, const int *d_A_col_indices,
                                int A_nnz, int m, double *d_x, double *d_y, int *iters_out, double *rr_out);

/* ---- KMC events: execute_kmc_step_gpu (kmc_events.cu:146-365) ------------------------------- */
/* solve_sparse_CG_splitmatrix (iterative_solvers_gpu.cu:656-821; unfinished in the reference: it prints the solution and calls
 * exit(1)) with add_submatrix_product (:634-652): UNPRECONDITIONED CG on (A + P^T M P) y = x, A in CSR (m rows), M dense msub x msub
 * (row-major, device), (P v)_k = v[insertion_indices[k] + index_offset] (the reference hard-codes + 2: node = atom + 2).  Loop while
 * ||r||_2 > tol (1e-5 in the source).  A and M are read only; y: start vector in, solution out.  The product's own current solve
 * keeps the dense block as symmetric tiles instead (dkmc_set_x_format(1)); this entry point completes the reference's split API. */
int dkmc_solve_sparse_CG_splitmatrix(const double *d_M, int msub, const double *d_A_data, const int *d_A_row_ptr, const int *d_A_col_indices,
                                     int A_nnz, int m, const int *d_insertion_indices, int index_offset, const double *d_x, double *d_y,
                                     double tol, int *iters_out, double *rnorm_out);

/* The reference draws two numbers per executed event from the host RandomNumberGenerator
 * (kmc_events.cu:221,348).  Here the caller passes the next n_uniform numbers of that stream
 * (h_uniform[2e] selects event e, h_uniform[2e+1] draws its waiting time); *n_events_out tells how
 * many events ran, i.e. 2 * n_events numbers were consumed.  If the stream runs out before the
 * stop criterion (event_time >= 1/freq) is met, *exhausted_out is set and the call can be repeated
 * with more numbers (state is kept in the site arrays).  h_event_log (optional, 4 ints per event):
 * slot index, i, j, event type.  Returns the last event time through *event_time_out. */
int dkmc_execute_kmc_step_gpu(int N, int nn, const int *d_neigh_idx, const int *d_site_layer,
                              const double *d_lattice, int pbc, const double *d_T_bg, const double *d_freq,
                              const double *d_sigma, const double *d_k,
                              const double *d_posx, const double *d_posy, const double *d_posz,
                              const double *d_site_potential_boundary, const double *d_site_potential_charge,
                              const double *d_site_temperature, int *d_site_element, int *d_site_charge,
                              const double *h_uniform, int n_uniform, int resume,
                              int *n_events_out, int *exhausted_out, int *h_event_log, double *event_time_out);
/* build_event_list only (kmc_events.cu:34-126); d_event_type may be NULL */
int dkmc_build_event_list(int N, int nn, const int *d_neigh_idx, const int *d_site_layer,
                          const double *d_lattice, int pbc, const double *d_T_bg, const double *d_freq,
                          const double *d_sigma, const double *d_k,
                          const double *d_posx, const double *d_posy, const double *d_posz,
                          const double *d_site_potential_boundary, const double *d_site_potential_charge,
                          const int *d_site_element, const int *d_site_charge,
                          int *d_event_type, double *d_event_prob);

/* ---- current / power: update_power_gpu_sparse (current_solver_gpu.cu:854-1147) --------------- */
int dkmc_update_power_gpu_sparse(dkmc_gpubuf *buf, int num_source_inj, int num_ground_ext, int num_layers_contact,
                                 double Vd, int pbc, double high_G, double low_G, double loop_G, double G0,
                                 double tol, double nn_dist, double m_e, double V0, int num_metals,
                                 double *h_imacro, int solve_heating_local, int solve_heating_global, double alpha_disp);
/* Assemble_X_sparsity + Assemble_X2 of the last update_power call, copied to host for inspection
 * (dump_csr_matrix_txt twin, iterative_solvers_gpu.cu:142-169).  Pass NULL pointers to query sizes. */
int dkmc_get_last_X(int *rows_out, long long *nnz_out, int *h_row_ptr, int *h_col, double *h_data);

/* ---- heat ------------------------------------------------------------------------------------ */
/* update_temperatureglobal_gpu (heat_solver_gpu.cu:52-69), the reference's (uncalled) device form */
int dkmc_update_temperatureglobal_gpu(const double *d_site_power, double *d_T_bg, int N, double a_coeff,
                                      double b_coeff, double number_steps, double C_thermal, double small_step);
/* the form the reference actually runs, on the host (heat_solver.cpp:316-350), done on the device */
int dkmc_update_temperature_global_analytic(const double *d_site_power, double *d_T_bg, int N, double event_time,
                                            double dissipation_constant, double t_ox, double A, double c_p,
                                            double *h_P_tot);

/* ---- local temperature model (host-only dense code in the reference: heat_solver.cpp:40-246, 286-308, 354-513) ----
 * Sparse form of the same linear systems, solved with the Jacobi-CG.
 * dkmc_construct_laplacian = Device::constructLaplacian: N_left_tot / N_right_tot are get_num_in_contacts (:5-37) evaluated
 * by the caller on the host copy of site_element, gamma = 1/(delta*(k_th_interface/k_th_metal + 1)) (:86).
 * dkmc_update_temperature_local = the local branch of Device::updateTemperature (:286-308): one steady-state solve when
 * step_time > 1e3*delta_t, else int(step_time/delta_t)+1 transient solves; reads gpubuf.site_power, updates
 * gpubuf.site_temperature and gpubuf.T_bg.  Tolerance of these solves: dkmc_set_heat_cg_tolerance (default 1e-10; the
 * reference applies an explicit inverse). */
void dkmc_set_heat_cg_tolerance(double tol);
int dkmc_construct_laplacian(const dkmc_gpubuf *buf, int N_left_tot, int N_right_tot, double gamma);
int dkmc_update_temperature_local(dkmc_gpubuf *buf, double step_time, double delta_t, double tau, double background_temp,
                                  double k_th_interface, double k_th_vacancies, double nn_dist, int num_atoms_contact,
                                  int *n_solves_out, int *steady_out, int *cg_iters_out, double *T_bg_out);

/* (test and measurement aids of the library -- dkmc_xt_time_share, dkmc_xt_check_shares, dkmc_xtb_check_product, dkmc_xtb_time_apply,
 * dkmc_debug_* -- are declared in devicekmc_hip_debug.h: they are exported by the .so but are not part of the drop-in surface) */

/* ---- multi-GPU: one simulation advanced in lockstep by N processes, one GPU each (no reference counterpart) ----
 * While a communicator is attached, update_power_gpu_sparse generates, stores and streams the tunnelling block of X in per-rank
 * shares (tiled X, the default) and completes the S-rows' sums with ONE in-place all-reduce of |S| + 1 doubles per matrix-vector
 * product; all ranks receive the same bits -- including rank 0's stop decision, from which every rank takes its control flow --
 * and the result equals the single-GPU one to rounding.  With dkmc_set_x_format(0) (CSR X) the long rows are dealt to the ranks
 * and an all-gather of row sums keeps the result bit-identical to the single-GPU one.  poisson_gridless_gpu sums a slab of sites
 * per rank and all-gathers the potentials (bit-identical).  Every other phase is computed redundantly and identically on every
 * rank, so all ranks hold the same state after every call.  Every rank must make the same sequence of
 * calls with the same inputs.
 * Transports: RCCL over xGMI (unique id from rank 0, distributed by the caller), or a host callback that all-gathers a
 * pinned host buffer in place (rehearsal on machines where the ranks share a GPU, which RCCL refuses). */
enum { DKMC_COMM_NONE = 0, DKMC_COMM_RCCL = 1, DKMC_COMM_HOST = 2 };
/* host_buf holds nranks chunks of bytes_per_rank bytes, the caller's own chunk (index rank) filled in; on return all are */
typedef int (*dkmc_allgather_fn)(void *host_buf, size_t bytes_per_rank, int rank, int nranks, void *user);
int dkmc_comm_unique_id(char *id128);                                   /* ncclGetUniqueId; 128 bytes out */
int dkmc_comm_init_rccl(int nranks, int rank, const char *id128);       /* after dkmc_set_gpu / dkmc_set_stream */
int dkmc_comm_init_host(int nranks, int rank, dkmc_allgather_fn fn, void *user);
int dkmc_comm_allgather_host(double *host_buf, size_t count_per_rank);  /* the callback transport's host half (no GPU needed) */
int dkmc_comm_info(int *nranks, int *rank, int *transport);
int dkmc_comm_destroy(void);
/* One-shot peer-write exchange of the sharded block-CG (csrc/comm.hip): beside an attached communicator, every rank
 * exports an exchange buffer of 2 x nranks slots of slot_doubles doubles and a row of sequence words (hipIpc; fine-grained device memory, so
 * that a peer DEVICE's stores are visible without a kernel boundary); prepare returns this rank's two handles + its device identity (192
 * bytes), the host all-gathers them through its process group and hands all of them (nranks x 192 bytes, rank order) to attach.  attach
 * refuses (error 49) a group whose ranks sit on different devices unless DKMC_PEER_CROSS_DEVICE=1 -- that case has never been run.  A sweep's exchange is then push + signal + bounded wait on the engine's stream and the slots are added in rank order; a solve
 * whose slots do not fit uses the communicator's all-gather.  Tested with two processes on one GPU; opt-in.  The reference has no counterpart. */
int dkmc_comm_peer_prepare(size_t slot_doubles, char *handles192);
int dkmc_comm_peer_attach(const char *all_handles);
int dkmc_comm_peer_detach(void);
int dkmc_comm_peer_info(int *ready, long long *slot_doubles, long long *exchanges, double *mean_us);

#ifdef __cplusplus
}
#endif
#endif
