// potential.hip -- charge rule, K sparsity + assembly, background potential / CB edge solves,
// screened-Coulomb pair sum.  Replaces potential_solver_gpu.cu (live parts) and the K-pattern
// builders of iterative_solvers_gpu.cu.
#include "common.h"
#include <stdint.h>
#include <algorithm>
#include <vector>

int kcg_assemble_and_solve(int cb, int m, int N_left, const int *element, const int *charge, MetalSet ms, double high_G, double low_G,
                           const int *rp, const int *ci, int nnz, const int *lrp, const int *lci, const int *rrp, const int *rci,
                           double VL, double VR, double *y, int *iters_out, double *rr_out, const KBlocked *kb,
                           const double *row_y, const double *row_z, int emu_nr, int emu_time_rank);
void kcg_slab_report(double *times_us, long long *halo_rows);
void kcg_slab_iter_cap(int cap);
KBlocked *kblocked_build(const int *rp_d, const int *ci_d, int m, int nnz, const double *x_d, hipStream_t st);
void kblocked_free(KBlocked *kb);
void tcache_invalidate(const void *key);
void xstate_reset(const void *key);

// ------------------------------------------------------------------------------------------------
// update_charge (potential_solver_gpu.cu:10-52).  One thread per site; only vacancies and oxygen
// ions (a few % of the sites) touch their neighbour row.  Padded slots (-1) are skipped: the
// reference reads element[-1] there.
__global__ __launch_bounds__(256) void k_update_charge(int N, int nn, const int *__restrict__ element, int *__restrict__ charge,
                                                       const int *__restrict__ neigh, MetalSet ms)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const int el = element[i];
    if (el != VACANCY && el != OXYGEN_DEFECT) return;
    const int *row = neigh + (size_t)i * nn;
    int vnn = 0; bool metal = false;
    for (int s = 0; s < nn; ++s) {
        const int j = row[s];
        if (j < 0) continue;
        const int ej = element[j];
        vnn += (ej == VACANCY);
        metal |= is_metal(ej, ms);
    }
    int c;
    if (el == VACANCY) c = (metal || vnn >= 2) ? 0 : 2;
    else c = metal ? 0 : -2;
    charge[i] = c;
}

extern "C" int dkmc_update_charge_gpu(const int *el, int *q, const int *neigh, int N, int nn, const int *d_metals, int nm)
{
    MetalSet ms = load_metals(d_metals, nm);
    hipLaunchKernelGGL(k_update_charge, dim3((N + 255) / 256), dim3(256), 0, eng().stream, N, nn, el, q, neigh, ms);
    KCHK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// K sparsity (Assemble_K_sparsity, iterative_solvers_gpu.cu:2158-2208).  The reference scans all
// m columns per row (O(m^2) site_dist calls).  The padded neighbour index already holds exactly
// the pairs with dist < nn_dist (same cutoff, kmc_main.cpp:121), ascending in j, so the three CSR
// blocks are a filtered copy of it: O(N * nn).
__global__ void k_kpat_count(int m, int N_left, int nn, const int *__restrict__ neigh, int *cd, int *cl, int *cr)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= m) return;
    const int *row = neigh + (size_t)(N_left + r) * nn;
    int d = 1, l = 0, rr = 0;       // diagonal is part of the device block (dist = 0 < cutoff)
    for (int s = 0; s < nn; ++s) {
        const int j = row[s];
        if (j < 0) continue;
        if (j < N_left) ++l; else if (j >= N_left + m) ++rr; else ++d;
    }
    cd[r] = d; cl[r] = l; cr[r] = rr;
}

__global__ void k_kpat_fill(int m, int N_left, int nn, const int *__restrict__ neigh,
                            const int *rpd, const int *rpl, const int *rpr, int *cold, int *coll, int *colr)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= m) return;
    const int i = N_left + r;
    const int *row = neigh + (size_t)i * nn;
    int pd = rpd[r], pl = rpl[r], pr = rpr[r];
    bool diag_done = false;
    for (int s = 0; s < nn; ++s) {
        const int j = row[s];
        if (j < 0) continue;
        if (j < N_left) coll[pl++] = j;
        else if (j >= N_left + m) colr[pr++] = j - (N_left + m);
        else {
            if (!diag_done && j > i) { cold[pd++] = r; diag_done = true; }
            cold[pd++] = j - N_left;
        }
    }
    if (!diag_done) cold[pd++] = r;
}

// shape of every K pattern built by initialize_sparsity, keyed by its row-pointer array: a solve with other contact sizes
// than the pattern was built for would index out of bounds
struct KPatInfo { const int *rp; int m, N_left; KBlocked *kb; };      // kb: the pattern's blocked form for the K solve (kcg.hip), or nullptr
static KPatInfo g_kpat[64]; static int g_kpat_next = 0;          // ring: the 64 most recently built patterns
static void kpat_forget(KPatInfo &k) { kblocked_free(k.kb); k = KPatInfo{nullptr, 0, 0, nullptr}; }
static void kpat_register(const int *rp, int m, int N_left, KBlocked *kb)
{
    for (auto &k : g_kpat) if (k.rp == rp) kpat_forget(k);
    KPatInfo &slot = g_kpat[g_kpat_next++ % 64];
    kpat_forget(slot);
    slot = KPatInfo{rp, m, N_left, kb};
}
static const KBlocked *kpat_blocked(const int *rp)
{
    for (const auto &k : g_kpat) if (k.rp == rp) return k.kb;
    return nullptr;
}
static bool kpat_matches(const int *rp, int m, int N_left)
{
    for (const auto &k : g_kpat) if (k.rp == rp) return k.m == m && k.N_left == N_left;
    return false;
}

__global__ void k_set_last(int *rp, int m, const int *total) { if (threadIdx.x == 0 && blockIdx.x == 0) rp[m] = *total; }


// releases what initialize_sparsity allocated (for hosts that own the other arrays themselves, e.g. the Python mirror) and the solver
// state kept for this buffer
extern "C" int dkmc_free_sparsity(dkmc_gpubuf *buf)
{
    xstate_reset(buf->site_x);
    HIPCHK(hipStreamSynchronize(eng().stream));
    int **ps[6] = { &buf->Device_row_ptr_d, &buf->Device_col_indices_d, &buf->contact_left_row_ptr, &buf->contact_left_col_indices,
                    &buf->contact_right_row_ptr, &buf->contact_right_col_indices };
    for (auto pp : ps) {
        if (!*pp) continue;
        for (auto &k : g_kpat) if (k.rp == *pp) kpat_forget(k);
        (void)hipFree(*pp); *pp = nullptr;
    }
    buf->Device_nnz = buf->contact_left_nnz = buf->contact_right_nnz = 0;
    return 0;
}

extern "C" int dkmc_initialize_sparsity(dkmc_gpubuf *buf, int pbc, double nn_dist, int num_atoms_contact)
{
    (void)pbc; (void)nn_dist;
    xstate_reset(buf->site_x);
    Engine &e = eng(); hipStream_t st = e.stream;
    const int N = buf->N_, nn = buf->nn_, N_left = num_atoms_contact, m = N - 2 * num_atoms_contact;
    if (m <= 0) return dkmc_fail(5, "initialize_sparsity: no device rows", __FILE__, __LINE__);
    int *cnt = (int *)scratch(S_MISC0, (size_t)3 * m * sizeof(int));
    int *tot = (int *)scratch(S_MISC1, 4 * sizeof(int));
    if (!cnt || !tot) return e.err_code;
    int **rps[3] = { &buf->Device_row_ptr_d, &buf->contact_left_row_ptr, &buf->contact_right_row_ptr };
    int **cis[3] = { &buf->Device_col_indices_d, &buf->contact_left_col_indices, &buf->contact_right_col_indices };
    for (int b = 0; b < 3; ++b) {
        if (*rps[b]) { for (auto &k : g_kpat) if (k.rp == *rps[b]) kpat_forget(k); (void)hipFree(*rps[b]); }
        if (*cis[b]) { (void)hipFree(*cis[b]); *cis[b] = nullptr; }
        HIPCHK(hipMalloc((void **)rps[b], (size_t)(m + 1) * sizeof(int)));
    }
    const int blocks = (m + 255) / 256;
    hipLaunchKernelGGL(k_kpat_count, dim3(blocks), dim3(256), 0, st, m, N_left, nn, buf->neigh_idx, cnt, cnt + m, cnt + 2 * m);
    int h_tot[3];
    for (int b = 0; b < 3; ++b) {
        int rc = dkmc_exclusive_scan_i32(cnt + (size_t)b * m, *rps[b], m, tot + b); if (rc) return rc;
        hipLaunchKernelGGL(k_set_last, dim3(1), dim3(1), 0, st, *rps[b], m, tot + b);
    }
    HIPCHK(hipMemcpyAsync(h_tot, tot, 3 * sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    buf->Device_nnz = h_tot[0]; buf->contact_left_nnz = h_tot[1]; buf->contact_right_nnz = h_tot[2];
    for (int b = 0; b < 3; ++b) HIPCHK(hipMalloc((void **)cis[b], (size_t)(h_tot[b] > 0 ? h_tot[b] : 1) * sizeof(int)));
    hipLaunchKernelGGL(k_kpat_fill, dim3(blocks), dim3(256), 0, st, m, N_left, nn, buf->neigh_idx,
                       buf->Device_row_ptr_d, buf->contact_left_row_ptr, buf->contact_right_row_ptr,
                       buf->Device_col_indices_d, buf->contact_left_col_indices, buf->contact_right_col_indices);
    KCHK();
    HIPCHK(hipStreamSynchronize(st));
    kpat_register(buf->Device_row_ptr_d, m, N_left, kblocked_build(buf->Device_row_ptr_d, buf->Device_col_indices_d, m, h_tot[0], buf->site_x + N_left, st));
    return 0;
}

// K values, rhs and the solve: kcg.hip (the off-diagonals of K take two values: the matrix is kept as class bits)
__global__ void k_fill_contacts(double *field, int N, int N_left, int N_right, double vl, double vr, double scale_all)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    double v = field[i];
    if (i < N_left) v = vl; else if (i >= N - N_right) v = vr;
    field[i] = v * scale_all;
}

static int solve_K(dkmc_gpubuf *buf, int N, int N_left, int N_right, double VL, double VR, int cb,
                   double high_G, double low_G, int num_metals, double *field, int *iters, double *rr)
{
    Engine &e = eng(); hipStream_t st = e.stream;
    const int m = N - N_left - N_right;
    if (!buf->Device_row_ptr_d || !buf->Device_col_indices_d || !buf->contact_left_row_ptr || !buf->contact_right_row_ptr)
        return dkmc_fail(6, "K sparsity not initialised (call initialize_sparsity)", __FILE__, __LINE__);
    if (N_left != N_right || !kpat_matches(buf->Device_row_ptr_d, m, N_left))
        return dkmc_fail(6, "K sparsity was built for other contact sizes than this solve asks for", __FILE__, __LINE__);
    MetalSet ms = load_metals(buf->metal_types, num_metals);
    (void)e; (void)st;
    return kcg_assemble_and_solve(cb, m, N_left, buf->site_element, buf->site_charge, ms, high_G, low_G, buf->Device_row_ptr_d, buf->Device_col_indices_d,
                                  buf->Device_nnz, buf->contact_left_row_ptr, buf->contact_left_col_indices, buf->contact_right_row_ptr,
                                  buf->contact_right_col_indices, VL, VR, field + N_left, iters, rr, kpat_blocked(buf->Device_row_ptr_d),
                                  buf->site_y + N_left, buf->site_z + N_left, 0, -1);
}

// ---- test / measurement aid: the slab-distributed CG on K (kcg.hip) with nranks VIRTUAL ranks on ONE GPU ---------------------------------------
// The background-potential system of the buffer's current state (elements, charges), solved twice from the buffer's current potential -- by the
// one-GPU reference-order loop on the CSR positions and by the slab-distributed loop with nranks virtual ranks (exchanges as device copies) -- into
// scratch copies: the buffer is not changed.  The emulation itself fails unless all virtual ranks stop at the same iteration with the same r.r
// and end with the same bits.  max_abs_diff: largest deviation of the two potentials [V]; times_us[4]: mean kernel times of virtual rank
// time_rank (product, update, direction, halo pack + unpack); halo_rows[2]: doubles a rank receives per iteration in the halo exchange (largest),
// rows of the largest slab.  iter_cap > 0: measurement run, the distributed loop stops after that many iterations (max_abs_diff = -1).
extern "C" int dkmc_kcg_emulate_slabs(dkmc_gpubuf *buf, int N, int N_left, int N_right, double Vd, double high_G, double low_G, int num_metals, int nranks, int time_rank,
                                      int iter_cap, double *max_abs_diff, int *iters_slab, int *iters_ref, double *times_us, long long *halo_rows)
{
    Engine &e = eng(); hipStream_t st = e.stream;
    const int m = N - N_left - N_right;
    if (!buf || m <= 0 || nranks < 1 || nranks > 32 || comm_attached()) return dkmc_fail(13, "kcg_emulate_slabs: bad arguments (or a communicator is attached)", __FILE__, __LINE__);
    if (!buf->Device_row_ptr_d || N_left != N_right || !kpat_matches(buf->Device_row_ptr_d, m, N_left)) return dkmc_fail(6, "K sparsity not initialised for these contact sizes", __FILE__, __LINE__);
    MetalSet ms = load_metals(buf->metal_types, num_metals);
    double *tmp = (double *)scratch(S_KS_EMU, (size_t)m * 8 * 2);
    if (!tmp) return e.err_code;
    double *yref = tmp, *yslab = tmp + m;
    HIPCHK(hipMemcpyAsync(yref, buf->site_potential_boundary + N_left, (size_t)m * 8, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemcpyAsync(yslab, buf->site_potential_boundary + N_left, (size_t)m * 8, hipMemcpyDeviceToDevice, st));
    if (iter_cap > 0) HIPCHK(hipMemsetAsync(yslab, 0, (size_t)m * 8, st));       // measurement run: from zero, so that the capped iterations are working ones (the buffer usually holds the solution)
    int it_ref = 0, it_slab = 0; double rr = 0.0;
    int rc = 0;
#define KS_SOLVE(Y, EMU, IT) kcg_assemble_and_solve(0, m, N_left, buf->site_element, buf->site_charge, ms, high_G, low_G, buf->Device_row_ptr_d, buf->Device_col_indices_d, \
        buf->Device_nnz, buf->contact_left_row_ptr, buf->contact_left_col_indices, buf->contact_right_row_ptr, buf->contact_right_col_indices, -Vd / 2, Vd / 2, Y, IT, &rr, \
        (const KBlocked *)nullptr, buf->site_y + N_left, buf->site_z + N_left, EMU, time_rank)
    if (iter_cap <= 0) { rc = KS_SOLVE(yref, 0, &it_ref); if (rc) return rc; }
    kcg_slab_iter_cap(iter_cap);
    rc = KS_SOLVE(yslab, nranks, &it_slab);
    kcg_slab_iter_cap(0);
#undef KS_SOLVE
    if (rc) return rc;
    kcg_slab_report(times_us, halo_rows);
    double md = -1.0;
    if (iter_cap <= 0) {
        std::vector<double> a((size_t)m), b((size_t)m);
        HIPCHK(hipMemcpyAsync(a.data(), yref, (size_t)m * 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(b.data(), yslab, (size_t)m * 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        md = 0.0;
        for (int i = 0; i < m; ++i) md = std::max(md, fabs(a[i] - b[i]));
    }
    if (max_abs_diff) *max_abs_diff = md;
    if (iters_slab) *iters_slab = it_slab;
    if (iters_ref) *iters_ref = it_ref;
    return e.err_code;
}

// background_potential_gpu_sparse (potential_solver_gpu.cu:696-781)
extern "C" int dkmc_background_potential_gpu_sparse(dkmc_gpubuf *buf, int N, int N_left, int N_right, double Vd, int pbc,
                                                    double high_G, double low_G, double nn_dist, int num_metals, int kmc_step_count)
{
    (void)pbc; (void)nn_dist; (void)kmc_step_count;
    Engine &e = eng();
    int rc = solve_K(buf, N, N_left, N_right, -Vd / 2, Vd / 2, 0, high_G, low_G, num_metals, buf->site_potential_boundary,
                     &e.stats.cg_iters_K, &e.stats.cg_rr_K);
    if (rc) return rc;
    hipLaunchKernelGGL(k_fill_contacts, dim3((N + 255) / 256), dim3(256), 0, e.stream, buf->site_potential_boundary, N, N_left, N_right,
                       -Vd / 2, Vd / 2, 1.0);
    KCHK();
    return 0;
}

// update_CB_edge_gpu_sparse (potential_solver_gpu.cu:595-694); result scaled by eV_to_J (:674)
extern "C" int dkmc_update_CB_edge_gpu_sparse(dkmc_gpubuf *buf, int N, int N_left, int N_right, double Vd, int pbc,
                                              double high_G, double low_G, double nn_dist, int num_metals)
{
    (void)pbc; (void)nn_dist;
    tcache_invalidate(buf->site_x);
    Engine &e = eng();
    int rc = solve_K(buf, N, N_left, N_right, Vd / 2, -Vd / 2, e.cb_edge_domain ? 2 : 1, high_G, low_G, num_metals, buf->site_CB_edge,
                     &e.stats.cg_iters_CB, &e.stats.cg_rr_CB);
    if (rc) return rc;
    hipLaunchKernelGGL(k_fill_contacts, dim3((N + 255) / 256), dim3(256), 0, e.stream, buf->site_CB_edge, N, N_left, N_right,
                       Vd / 2, -Vd / 2, DKMC_Q);
    KCHK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// poisson_gridless (potential_solver_gpu.cu:908-978).  The reference launches N x ceil(N/512)
// blocks (N^2 threads), though only columns with a non-zero charge contribute, and combines block
// sums with atomicAdd in arrival order.  Here: (1) the charged sites are compacted, in ascending
// site order, into a packed {x,y,z,q,idx} list; (2) a workgroup owns 64 sites and sweeps the list
// through LDS tiles, its four waves taking every fourth list entry; fixed summation order (eight
// interleaved partial sums per site, combined in a fixed tree), no atomics, no memset.
struct __attribute__((aligned(32))) ChargedSite { double x, y, z; int q, idx; };

__global__ void k_charge_flags(int N, const int *charge, int *flag)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) flag[i] = charge[i] != 0;
}
__global__ void k_charge_scatter(int N, const int *__restrict__ charge, const int *__restrict__ off,
                                 const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ z,
                                 ChargedSite *list)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N && charge[i] != 0) { ChargedSite c; c.x = x[i]; c.y = y[i]; c.z = z[i]; c.q = charge[i]; c.idx = i; list[off[i]] = c; }
}

#define PW_NT 256
#define PW_SITES 64             // sites per workgroup: one per lane; the four waves split the charged list
#define PW_CUT 6.5              // default of dkmc_set_pair_cutoff: pairs with r / (sigma sqrt 2) beyond this are not evaluated, erfc(6.5) = 3.8e-20
// Each workgroup owns 64 sites; wave w adds the charged sites c = w, w + 4, ... of every LDS tile (all lanes of a wave read the
// same list entry: an LDS broadcast), two independent accumulators per lane; the four partial sums of a site are combined in a
// fixed order.  Four times the waves of a thread-per-site launch (85 k sites: 21 waves per CU instead of 5) and twice the
// independent erfc / sqrt / divide chains per lane.
// Screening cut-off: a term is q erfc(x) k Q / r with x = r / (sigma sqrt 2).  Beyond x = 6.5 (r > 32 A at sigma = 3.5 A) it is below
// 3.8e-20 k Q / r, i.e. < 1e-19 of a nearest-neighbour term; all such terms of a 1e6-site stack together stay under 2e-17 V, below
// the rounding of the sum itself (the reference's atomicAdd order already moves the last bits).  Those pairs pay the
// distance (12 flops) but not erfc / sqrt / divide: 95 % of the pairs at 9.4e5 sites.  The count of evaluated pairs is reported.
__device__ __forceinline__ double pw_term(double xi, double yi, double zi, const ChargedSite &c, int i, double laty, double latz, int pbc,
                                          double sigma, double kk, double cut2, int &neval)
{
    double dx, dy, dz;
    if (pbc) {
        dx = xi - c.x;
        double fy = (yi - c.y) / laty; fy -= round(fy);
        double fz = (zi - c.z) / latz; fz -= round(fz);
        dy = fy * laty; dz = fz * latz;
    } else { dx = c.x - xi; dy = c.y - yi; dz = c.z - zi; }
    const double d2 = dx * dx + dy * dy + dz * dz;                 // the argument of site_dist's sqrt (gpu_solvers.h:225-257), same order
    if (c.idx == i || d2 > cut2) return 0.0;
    ++neval;
    return v_solve(1e-10 * sqrt(d2), c.q, sigma, kk);
}
__global__ __launch_bounds__(PW_NT) void k_pairwise(int N, const double *__restrict__ x, const double *__restrict__ y,
                                                    const double *__restrict__ z, const double *__restrict__ lattice, int pbc,
                                                    const double *__restrict__ sigma_p, const double *__restrict__ k_p,
                                                    const ChargedSite *__restrict__ list, const int *__restrict__ ncharged,
                                                    double *__restrict__ out, unsigned long long *__restrict__ nevaluated, int i0, double xcut,
                                                    const int *__restrict__ cells_in_use)
{
    if (*cells_in_use) return;                                     // the cell-list kernel (k_pairwise_cells, below) does this call
    __shared__ ChargedSite tile[PW_NT];
    __shared__ double partial[PW_NT / 64][PW_SITES];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i = i0 + blockIdx.x * PW_SITES + lane;               // sites [i0, N): all of them on one GPU, this rank's slab in a sharded run
    const int nc = *ncharged;
    const double sigma = *sigma_p, kk = *k_p, laty = lattice[1], latz = lattice[2];
    const double rc = xcut * sigma * sqrt(2.0) * 1e10;             // [A]; xcut = 0: every pair, as the reference sums (dkmc_set_pair_cutoff)
    const double cut2 = xcut > 0.0 ? rc * rc : 1.0e300;
    const double xi = i < N ? x[i] : 0.0, yi = i < N ? y[i] : 0.0, zi = i < N ? z[i] : 0.0;
    double v0 = 0.0, v1 = 0.0;
    int neval = 0, ntest = 0;
    for (int base = 0; base < nc; base += PW_NT) {
        const int n = min(PW_NT, nc - base);
        __syncthreads();
        if (threadIdx.x < n) tile[threadIdx.x] = list[base + threadIdx.x];
        __syncthreads();
        if (i < N) {
            int c = w;
            ntest += (n - w + 3) / 4;                                // entries w, w + 4, ... of this tile
            for (; c + 4 < n; c += 8) {
                v0 += pw_term(xi, yi, zi, tile[c], i, laty, latz, pbc, sigma, kk, cut2, neval);
                v1 += pw_term(xi, yi, zi, tile[c + 4], i, laty, latz, pbc, sigma, kk, cut2, neval);
            }
            if (c < n) v0 += pw_term(xi, yi, zi, tile[c], i, laty, latz, pbc, sigma, kk, cut2, neval);
        }
    }
    partial[w][lane] = v0 + v1;
    neval = wave_sum_all_i(neval); ntest = wave_sum_all_i(ntest);
    if (lane == 0 && neval) atomicAdd(nevaluated, (unsigned long long)neval);
    if (lane == 0 && ntest) atomicAdd(nevaluated + 1, (unsigned long long)ntest);
    __syncthreads();
    if (w == 0 && i < N) out[i] = (partial[0][lane] + partial[1][lane]) + (partial[2][lane] + partial[3][lane]);
}

// ---- pair sum over a cell list of the charged sites ---------------------------------------------------------------------------
// With the screening cut-off on, a site only needs the charged sites within rc = x_cut sigma sqrt 2 (32 A).  The charged sites are binned
// into (y, z) columns of edge >= rc / 2 -- a site sums over the (2 r + 1)^2 columns around its own, r = ceil(rc / edge) <= 2: 6.25 rc^2 of
// lateral area instead of the 9 rc^2 of round 3's columns of edge rc -- and, inside a column, into x bins of ~rc / 4: the sites of a column are
// grouped by x bin too, so the 64 sites of a workgroup span one or two bins and only the charged sites whose bin lies within rc of that
// span are swept (a contact site 30 A from the oxide reads almost nothing; x was unbinned before).  Measured at 3.76e6 sites: 7.5 -> 3.x
// distance tests per evaluated pair.  Workgroups own 64 sites OF ONE COLUMN (sites grouped by column and x bin through a permutation; their
// order inside a bin does not matter, every site is summed independently); the four waves sweep LDS tiles of the gathered segments with
// broadcast reads.  Determinism: the charged sites of a column are kept in (x bin, ascending site) order -- a stable, atomic-free partition by
// column (one workgroup per column walks the compacted list) and a stable ballot sort by x bin inside it --, columns and bins are visited in
// a fixed order, the four partial sums are combined as before.  Everything that depends on the lattice is decided on the device (no host
// read of the box per call): the kernels of the path not taken find `use` cleared and return.
#define PW_MAXDIM 32
#define PW_MAXCELL (PW_MAXDIM * PW_MAXDIM)
#define PW_MAXX 16                  // x bins per column
#define PW_MIN_CHARGED 512
struct PwGrid { int use, ny, nz, pbc; double hy, hz, ly, lz; int nchunks, rebuild; int ry, rz, nx, rx; double hx, x0; };
__device__ __forceinline__ int pw_axis_cell(double v, double L, double h, int n, int pbc)
{
    if (pbc) { double f = v / L; f -= floor(f); return min((int)(f * n), n - 1); }
    return min(max((int)floor(v / h), 0), n - 1);              // clamping is monotone: points within h of each other stay in adjacent cells
}
__device__ __forceinline__ int pw_xbin(double x, const PwGrid &G) { return min(max((int)floor((x - G.x0) / G.hx), 0), G.nx - 1); }      // (x is never periodic)
// The grid of this call.  The grouping of the SITES by column depends only on positions, box and cut-off, which do not change between
// calls: it is rebuilt when the host sees other position arrays / N / pbc / cut-off (host_rebuild) or when the box or sigma read here
// differ from those of the cached grouping; otherwise only the charged sites are binned per call.
__global__ void k_pw_grid(const double *__restrict__ lattice, const double *__restrict__ sigma_p, double xcut, int pbc, const int *__restrict__ ncharged,
                          PwGrid *g, int *__restrict__ tcount, int *__restrict__ ccount, int *__restrict__ cursor, int host_rebuild)
{
    __shared__ int rebuild_s;
    const int t = threadIdx.x;
    if (t == 0) {
        const double rc = xcut * *sigma_p * sqrt(2.0) * 1e10;
        const PwGrid old = *g;
        PwGrid G{};
        G.pbc = pbc; G.ly = lattice[1]; G.lz = lattice[2];
        G.ny = xcut > 0.0 ? max(1, min(PW_MAXDIM, (int)floor(2.0 * G.ly / rc))) : 1;
        G.nz = xcut > 0.0 ? max(1, min(PW_MAXDIM, (int)floor(2.0 * G.lz / rc))) : 1;
        G.hy = G.ly / G.ny; G.hz = G.lz / G.nz;
        G.ry = xcut > 0.0 ? (int)ceil(rc / G.hy) : 0; G.rz = xcut > 0.0 ? (int)ceil(rc / G.hz) : 0;       // columns to either side (1 or 2; more only on a box narrower than rc / 2)
        // x: the device spans [0, lattice[0]]; bins of ~rc / 4 (clamped: a site outside the box lands in the first / last bin, whose reach then
        // only grows -- the cut below is conservative by construction: it drops a bin only if EVERY point of it is farther than rc from the span)
        G.x0 = 0.0;
        G.nx = xcut > 0.0 ? max(1, min(PW_MAXX, (int)floor(4.0 * lattice[0] / rc))) : 1;
        G.hx = lattice[0] / G.nx; G.rx = 0;
        G.use = xcut > 0.0 && (G.ny >= 2 * G.ry + 1 || G.nz >= 2 * G.rz + 1) && min(2 * G.ry + 1, G.ny) * min(2 * G.rz + 1, G.nz) <= 32 && *ncharged >= PW_MIN_CHARGED;
        G.rebuild = host_rebuild || old.ny != G.ny || old.nz != G.nz || old.pbc != G.pbc || old.ly != G.ly || old.lz != G.lz || old.nx != G.nx || old.hx != G.hx;
        G.nchunks = G.rebuild ? 0 : old.nchunks;
        *g = G;
        rebuild_s = G.rebuild;
    }
    __syncthreads();
    for (int i = t; i <= PW_MAXCELL; i += blockDim.x) { ccount[i] = 0; if (rebuild_s) { tcount[i] = 0; cursor[i] = 0; } }
}
// column of every charged site (every call) and of every site (rebuild only); counts through an LDS histogram per workgroup, then one
// global integer atomic per (workgroup, occupied column): the counts do not depend on any order
__global__ __launch_bounds__(256) void k_pw_bin(int N, const double *__restrict__ y, const double *__restrict__ z, const PwGrid *__restrict__ g,
                                                const ChargedSite *__restrict__ list, const int *__restrict__ ncharged, int *__restrict__ site_cell,
                                                int *__restrict__ ccell, int *__restrict__ tcount, int *__restrict__ ccount)
{
    const PwGrid G = *g;
    const int nc = *ncharged;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nc) {                                                 // charged sites: few (1 % of the sites), plain atomics
        const ChargedSite cs = list[i];
        const int c = pw_axis_cell(cs.z, G.lz, G.hz, G.nz, G.pbc) * G.ny + pw_axis_cell(cs.y, G.ly, G.hy, G.ny, G.pbc);
        ccell[i] = c; atomicAdd(&ccount[c], 1);
    }
    if (!G.rebuild) return;
    __shared__ int hist[PW_MAXCELL];
    const int ncell = G.ny * G.nz;
    for (int k = threadIdx.x; k < ncell; k += 256) hist[k] = 0;
    __syncthreads();
    if (i < N) {
        const int c = pw_axis_cell(z[i], G.lz, G.hz, G.nz, G.pbc) * G.ny + pw_axis_cell(y[i], G.ly, G.hy, G.ny, G.pbc);
        site_cell[i] = c; atomicAdd(&hist[c], 1);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < ncell; k += 256) if (hist[k]) atomicAdd(&tcount[k], hist[k]);
}
// prefix sums over the (at most 1024) columns: cstart every call; tstart and the first chunk of 64 sites of every column on a rebuild
__global__ __launch_bounds__(PW_MAXCELL) void k_pw_offsets(PwGrid *g, const int *__restrict__ tcount, const int *__restrict__ ccount, int *__restrict__ tstart,
                                                          int *__restrict__ cstart, int *__restrict__ chunk0)
{
    __shared__ int a[PW_MAXCELL], b[PW_MAXCELL], c[PW_MAXCELL];
    const int t = threadIdx.x, n = g->ny * g->nz, rebuild = g->rebuild;
    a[t] = t < n ? tcount[t] : 0; b[t] = t < n ? ccount[t] : 0; c[t] = t < n ? (tcount[t] + PW_SITES - 1) / PW_SITES : 0;
    __syncthreads();
    for (int off = 1; off < PW_MAXCELL; off <<= 1) {                 // inclusive scans (Hillis-Steele; 1024 entries)
        const int va = t >= off ? a[t - off] : 0, vb = t >= off ? b[t - off] : 0, vc = t >= off ? c[t - off] : 0;
        __syncthreads();
        a[t] += va; b[t] += vb; c[t] += vc;
        __syncthreads();
    }
    if (t < n) { cstart[t + 1] = b[t]; if (rebuild) { tstart[t + 1] = a[t]; chunk0[t + 1] = c[t]; } }
    if (t == 0) { cstart[0] = 0; if (rebuild) { tstart[0] = 0; chunk0[0] = 0; g->nchunks = c[n - 1]; } }
}
// sites grouped by column (any order inside a column): a workgroup reserves one range per occupied column, ranks inside it from LDS
__global__ __launch_bounds__(256) void k_pw_perm(int N, const PwGrid *__restrict__ g, const int *__restrict__ site_cell, const int *__restrict__ tstart,
                                                 int *__restrict__ cursor, int *__restrict__ perm)
{
    if (!g->rebuild) return;
    __shared__ int hist[PW_MAXCELL], base[PW_MAXCELL];
    const int ncell = g->ny * g->nz;
    for (int k = threadIdx.x; k < ncell; k += 256) hist[k] = 0;
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int c = -1, r = 0;
    if (i < N) { c = site_cell[i]; r = atomicAdd(&hist[c], 1); }
    __syncthreads();
    for (int k = threadIdx.x; k < ncell; k += 256) if (hist[k]) base[k] = tstart[k] + atomicAdd(&cursor[k], hist[k]);
    __syncthreads();
    if (c >= 0) perm[base[c] + r] = i;
}
// ... then by x bin inside its column (rebuild only): workgroup = column, a counting sort of the column's part of the permutation (any order
// inside a bin).  perm -> perm2
__global__ __launch_bounds__(256) void k_pw_perm_x(const PwGrid *__restrict__ g, const double *__restrict__ x, const int *__restrict__ tstart,
                                                   const int *__restrict__ perm, int *__restrict__ perm2)
{
    if (!g->rebuild) return;
    const PwGrid G = *g;
    const int c = blockIdx.x;
    if (c >= G.ny * G.nz) return;
    __shared__ int cnt[PW_MAXX], off[PW_MAXX];
    if (threadIdx.x < PW_MAXX) cnt[threadIdx.x] = 0;
    __syncthreads();
    const int t0 = tstart[c], t1 = tstart[c + 1];
    for (int t = t0 + threadIdx.x; t < t1; t += 256) atomicAdd(&cnt[pw_xbin(x[perm[t]], G)], 1);
    __syncthreads();
    if (threadIdx.x == 0) { int a = 0; for (int b = 0; b < PW_MAXX; ++b) { off[b] = a; a += cnt[b]; } }
    __syncthreads();
    for (int t = t0 + threadIdx.x; t < t1; t += 256) { const int i = perm[t]; perm2[t0 + atomicAdd(&off[pw_xbin(x[i], G)], 1)] = i; }
}
// ... and every (column, x bin) segment into ascending site order (rebuild only): which site sits in which chunk of 64 decides the chunk's x span,
// hence which charged sites it sweeps and how its sums are grouped -- the two groupings above use atomics, so without this pass a result would
// be reproducible only to rounding.  Workgroup = column; segments of up to 8 192 sites are sorted in LDS (bitonic); a larger one (> 40 x the
// sites such a cell can hold) is sorted in pieces.
#define PW_SORTCAP 8192
__global__ __launch_bounds__(256) void k_pw_perm_sort(const PwGrid *__restrict__ g, const double *__restrict__ x, const int *__restrict__ tstart, int *__restrict__ perm2)
{
    if (!g->rebuild) return;
    const PwGrid G = *g;
    const int c = blockIdx.x;
    if (c >= G.ny * G.nz) return;
    __shared__ int key[PW_SORTCAP];
    __shared__ int cnt[PW_MAXX];
    if (threadIdx.x < PW_MAXX) cnt[threadIdx.x] = 0;
    __syncthreads();
    const int t0 = tstart[c], t1 = tstart[c + 1];
    for (int t = t0 + threadIdx.x; t < t1; t += 256) atomicAdd(&cnt[pw_xbin(x[perm2[t]], G)], 1);
    __syncthreads();
    int s0 = t0;
    for (int b = 0; b < G.nx; ++b) {
        const int nb = cnt[b];
        for (int p0 = 0; p0 < nb; p0 += PW_SORTCAP) {
            const int n = min(PW_SORTCAP, nb - p0);
            int np2 = 1; while (np2 < n) np2 <<= 1;
            for (int i = threadIdx.x; i < np2; i += 256) key[i] = i < n ? perm2[s0 + p0 + i] : 0x7fffffff;
            __syncthreads();
            for (int k = 2; k <= np2; k <<= 1)
                for (int j = k >> 1; j > 0; j >>= 1) {
                    for (int i = threadIdx.x; i < np2; i += 256) {
                        const int l = i ^ j;
                        if (l > i) {
                            const int a = key[i], bb = key[l];
                            const bool up = (i & k) == 0;
                            if ((a > bb) == up) { key[i] = bb; key[l] = a; }
                        }
                    }
                    __syncthreads();
                }
            for (int i = threadIdx.x; i < n; i += 256) perm2[s0 + p0 + i] = key[i];
            __syncthreads();
        }
        s0 += nb;
    }
}
// charged sites grouped by column, ascending site order kept: workgroup = column, a stable partition of the compacted list
__global__ __launch_bounds__(256) void k_pw_partition(const PwGrid *__restrict__ g, const ChargedSite *__restrict__ list, const int *__restrict__ ncharged,
                                                      const int *__restrict__ ccell, const int *__restrict__ cstart, ChargedSite *__restrict__ out)
{
    if (!g->use) return;
    const int c = blockIdx.x;
    if (c >= g->ny * g->nz) return;
    __shared__ int wcnt[4];
    __shared__ int base_s;
    const int nc = *ncharged, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x == 0) base_s = cstart[c];
    __syncthreads();
    for (int b0 = 0; b0 < nc; b0 += 256) {
        const int i = b0 + threadIdx.x;
        const bool mine = i < nc && ccell[i] == c;
        const unsigned long long bal = __ballot(mine);
        if (lane == 0) wcnt[w] = __popcll(bal);
        __syncthreads();
        int off = base_s;
        for (int u = 0; u < w; ++u) off += wcnt[u];
        if (mine) out[off + __popcll(bal & ((1ull << lane) - 1ull))] = list[i];
        __syncthreads();
        if (threadIdx.x == 0) base_s += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        __syncthreads();
    }
}
// ... then by x bin inside its column, site order kept inside a bin: one WAVE per column (a column holds tens of charged sites), bin by bin, the
// members of a bin extracted with ballots in list order.  in -> out; cxoff[c * (PW_MAXX + 1) + b] = first entry of bin b relative to cstart[c]
__global__ __launch_bounds__(64) void k_pw_partition_x(const PwGrid *__restrict__ g, const int *__restrict__ cstart, const ChargedSite *__restrict__ in,
                                                       ChargedSite *__restrict__ out, int *__restrict__ cxoff)
{
    if (!g->use) return;
    const PwGrid G = *g;
    const int c = blockIdx.x, lane = threadIdx.x;
    if (c >= G.ny * G.nz) return;
    const int s0 = cstart[c], s1 = cstart[c + 1];
    int pos = 0;
    for (int b = 0; b < G.nx; ++b) {
        if (lane == 0) cxoff[c * (PW_MAXX + 1) + b] = pos;
        for (int i0 = s0; i0 < s1; i0 += 64) {
            const int i = i0 + lane;
            const bool mine = i < s1 && pw_xbin(in[i].x, G) == b;
            const unsigned long long bal = __ballot(mine);
            if (mine) out[s0 + pos + __popcll(bal & ((1ull << lane) - 1ull))] = in[i];
            pos += __popcll(bal);
        }
    }
    if (lane == 0) for (int b = G.nx; b <= PW_MAXX; ++b) cxoff[c * (PW_MAXX + 1) + b] = pos;
}
// the sum itself: block -> (column, chunk of 64 of its sites, x-ordered); the columns within reach in a fixed order, of each the x bins within
// rc of the chunk's span; the segments are gathered into LDS tiles of 256 entries
#define PW_MAXSEG 32
__global__ __launch_bounds__(PW_NT) void k_pairwise_cells(int N, const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ z,
                                                          int pbc, const double *__restrict__ sigma_p, const double *__restrict__ k_p,
                                                          const PwGrid *__restrict__ g, const int *__restrict__ tstart, const int *__restrict__ cstart,
                                                          const int *__restrict__ chunk0, const int *__restrict__ perm, const ChargedSite *__restrict__ clist,
                                                          const int *__restrict__ cxoff,
                                                          double *__restrict__ out, unsigned long long *__restrict__ nevaluated, int i_lo, int i_hi, double xcut)
{
    if (!g->use) return;
    const PwGrid G = *g;
    if ((int)blockIdx.x >= G.nchunks) return;
    __shared__ ChargedSite tile[PW_NT];
    __shared__ double partial[PW_NT / 64][PW_SITES];
    __shared__ int seg_lo[PW_MAXSEG], seg_pre[PW_MAXSEG + 1], nseg_s;
    __shared__ double xspan[2];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int ncell = G.ny * G.nz;
    int lo = 0, hi = ncell;                                           // column of this chunk: last c with chunk0[c] <= blockIdx.x
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (chunk0[mid] <= (int)blockIdx.x) lo = mid; else hi = mid; }
    const int c = lo, cy = c % G.ny, cz = c / G.ny;
    const int t = tstart[c] + ((int)blockIdx.x - chunk0[c]) * PW_SITES + lane;
    const int isite = t < tstart[c + 1] ? perm[t] : -1;                // (the span below is over ALL sites of the chunk, also those of other ranks' slabs)
    int i = isite;
    if (i >= 0 && (i < i_lo || i >= i_hi)) i = -1;                     // sharded run: this rank's slab of sites only
    const double sigma = *sigma_p, kk = *k_p;
    const double rc = xcut * sigma * sqrt(2.0) * 1e10, cut2 = rc * rc;
    const double xi = isite >= 0 ? x[isite] : 0.0, yi = i >= 0 ? y[i] : 0.0, zi = i >= 0 ? z[i] : 0.0;
    // x span of the chunk's sites (every wave holds the same 64 sites)
    if (w == 0) {
        double xmn = isite >= 0 ? xi : 1.0e300, xmx = isite >= 0 ? xi : -1.0e300;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { xmn = fmin(xmn, __shfl_xor(xmn, o, 64)); xmx = fmax(xmx, __shfl_xor(xmx, o, 64)); }
        if (lane == 0) { xspan[0] = xmn; xspan[1] = xmx; }
    }
    __syncthreads();
    // segments: thread u < (2 ry + 1)(2 rz + 1) takes neighbour column u.  A bin b is needed unless all of it lies farther than rc from the span:
    // bins are [x0 + b hx, x0 + (b + 1) hx), the first / last one open-ended (out-of-box sites are clamped into them)
    if (threadIdx.x == 0) nseg_s = 0;
    const int wy = min(2 * G.ry + 1, G.ny), wz = min(2 * G.rz + 1, G.nz);      // distinct columns along an axis (a narrow / periodic axis: all of them, once)
    const int nnb = wy * wz;
    int my_lo = 0, my_n = 0;
    if ((int)threadIdx.x < nnb && nnb <= PW_MAXSEG) {
        const int uy = threadIdx.x % wy, uz = threadIdx.x / wy;
        int ny_ = wy == G.ny ? uy : cy - G.ry + uy, nz_ = wz == G.nz ? uz : cz - G.rz + uz;
        bool ok = true;
        if (wy != G.ny) { if (pbc) ny_ = (ny_ % G.ny + G.ny) % G.ny; else if (ny_ < 0 || ny_ >= G.ny) ok = false; }
        if (wz != G.nz) { if (pbc) nz_ = (nz_ % G.nz + G.nz) % G.nz; else if (nz_ < 0 || nz_ >= G.nz) ok = false; }
        if (ok) {
            const int cc = nz_ * G.ny + ny_;
            int blo = (int)floor((xspan[0] - rc - G.x0) / G.hx), bhi = (int)floor((xspan[1] + rc - G.x0) / G.hx);
            blo = min(max(blo, 0), G.nx - 1); bhi = min(max(bhi, 0), G.nx - 1);
            if (xspan[0] <= xspan[1]) { my_lo = cstart[cc] + cxoff[cc * (PW_MAXX + 1) + blo]; my_n = cstart[cc] + cxoff[cc * (PW_MAXX + 1) + bhi + 1] - my_lo; }
        }
    }
    __syncthreads();
    // fixed order: thread 0 walks the neighbour slots (wave 0 holds them all: nnb <= 32)
    if (w == 0) {
        int pre = 0;
        for (int u = 0; u < nnb && nnb <= PW_MAXSEG; ++u) {
            const int l_ = __shfl(my_lo, u, 64), n_ = __shfl(my_n, u, 64);
            if (lane == 0 && n_ > 0) { const int k_ = nseg_s; seg_lo[k_] = l_; seg_pre[k_] = pre; nseg_s = k_ + 1; }
            pre += n_ > 0 ? n_ : 0;
        }
        if (lane == 0) seg_pre[nseg_s] = pre;
    }
    __syncthreads();
    const int nseg = nseg_s, total = seg_pre[nseg];
    double v0 = 0.0, v1 = 0.0;
    int neval = 0, ntest = 0;
    for (int base = 0; base < total; base += PW_NT) {
        const int n = min(PW_NT, total - base);
        __syncthreads();
        if ((int)threadIdx.x < n) {
            const int vp = base + threadIdx.x;
            int k_ = 0;
            while (k_ + 1 < nseg && seg_pre[k_ + 1] <= vp) ++k_;
            tile[threadIdx.x] = clist[seg_lo[k_] + (vp - seg_pre[k_])];
        }
        __syncthreads();
        if (i >= 0) {
            int q = w;
            ntest += (n - w + 3) / 4;
            for (; q + 4 < n; q += 8) {
                v0 += pw_term(xi, yi, zi, tile[q], i, G.ly, G.lz, pbc, sigma, kk, cut2, neval);
                v1 += pw_term(xi, yi, zi, tile[q + 4], i, G.ly, G.lz, pbc, sigma, kk, cut2, neval);
            }
            if (q < n) v0 += pw_term(xi, yi, zi, tile[q], i, G.ly, G.lz, pbc, sigma, kk, cut2, neval);
        }
    }
    partial[w][lane] = v0 + v1;
    neval = wave_sum_all_i(neval); ntest = wave_sum_all_i(ntest);
    if (lane == 0 && neval) atomicAdd(nevaluated, (unsigned long long)neval);
    if (lane == 0 && ntest) atomicAdd(nevaluated + 1, (unsigned long long)ntest);
    __syncthreads();
    if (w == 0 && i >= 0) out[i] = (partial[0][lane] + partial[1][lane]) + (partial[2][lane] + partial[3][lane]);
}

// what the host knows of the cached grouping of the sites by column (see dkmc_poisson_gridless_gpu)
static struct PwKey { const void *x, *y, *z, *lattice, *perm, *cells; int N, pbc; double cut; int possible, ncell; } g_pw_key = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, -1.0, 0, 0};
void pairsum_invalidate() { g_pw_key.x = nullptr; g_pw_key.N = 0; }
extern "C" void dkmc_reset_pair_sum_cache(void) { pairsum_invalidate(); }

extern "C" int dkmc_poisson_gridless_gpu(int num_atoms_contact, int pbc, int N, const double *lattice, const double *sigma,
                                         const double *k, const double *x, const double *y, const double *z,
                                         const int *charge, double *out)
{
    (void)num_atoms_contact;
    Engine &e = eng(); hipStream_t st = e.stream;
    int *flag = (int *)scratch(S_MISC0, (size_t)N * 4), *off = (int *)scratch(S_MISC1, (size_t)N * 4);
    int *cnt = (int *)scratch(S_PW_CNT, 32);
    ChargedSite *list = (ChargedSite *)scratch(S_PW_LIST, (size_t)N * sizeof(ChargedSite));
    if (!flag || !off || !cnt || !list) return e.err_code;
    const int blocks = (N + 255) / 256;
    hipLaunchKernelGGL(k_charge_flags, dim3(blocks), dim3(256), 0, st, N, charge, flag);
    int rc = dkmc_exclusive_scan_i32(flag, off, N, cnt); if (rc) return rc;
    hipLaunchKernelGGL(k_charge_scatter, dim3(blocks), dim3(256), 0, st, N, charge, off, x, y, z, list);
    static hipEvent_t evp[2]; static bool evp_ready = false;
    if (e.profiling) {
        if (!evp_ready) { HIPCHK(hipEventCreate(&evp[0])); HIPCHK(hipEventCreate(&evp[1])); evp_ready = true; }
        HIPCHK(hipEventRecord(evp[0], st));
    }
    unsigned long long *d_ne = (unsigned long long *)(cnt + 4);          // [0] pairs evaluated (inside the cut-off), [1] pairs tested
    HIPCHK(hipMemsetAsync(d_ne, 0, 16, st));
    // cell list over the charged sites (taken on the device when the cut-off is on, the box has >= 3 columns along y or z and
    // enough sites are charged; otherwise these launches return at once and k_pairwise sums over the whole list)
    int *cells = (int *)scratch(S_PW_CELLS, (size_t)(8 * (PW_MAXCELL + 1) + PW_MAXCELL * (PW_MAXX + 1)) * 4 + sizeof(PwGrid) + 16);
    int *pperm = (int *)scratch(S_PW_PERM, (size_t)N * 4 * 4);
    ChargedSite *clist2 = (ChargedSite *)scratch(S_PW_LIST2, (size_t)N * sizeof(ChargedSite) * 2);
    if (!cells || !pperm || !clist2) return e.err_code;
    ChargedSite *clist3 = clist2 + N;                                  // the charged list grouped by column, then by x bin inside a column
    int *cxoff = cells + 8 * (PW_MAXCELL + 1);                         // [column][PW_MAXX + 1]: first entry of every x bin, relative to the column's start
    int *pperm2 = pperm + 3 * (size_t)N;                               // the sites grouped by column, x-ordered inside a column (what the sum kernel walks)
    int *tcount = cells, *ccount = cells + (PW_MAXCELL + 1), *cursor = cells + 2 * (PW_MAXCELL + 1);
    int *tstart = cells + 4 * (PW_MAXCELL + 1), *cstart = cells + 5 * (PW_MAXCELL + 1), *chunk0 = cells + 6 * (PW_MAXCELL + 1);
    PwGrid *grid = (PwGrid *)(((uintptr_t)(cells + 8 * (PW_MAXCELL + 1) + PW_MAXCELL * (PW_MAXX + 1)) + 15) & ~(uintptr_t)15);
    int *site_cell = pperm + N, *ccell = pperm + 2 * (size_t)N;
    // the grouping of the sites by column is kept between calls (positions, box and cut-off do not change); what the host can see of its key:
    // `possible`: the box has >= 3 columns along y or z, read back ONCE per key (one 64-byte copy); a box without (85 k sites: 2 x 2) skips
    // every launch of the cell path from then on
    // The key is dropped whenever a GPUBuffers is freed or (re-)initialised (pairsum_invalidate, called from xstate_reset): hipMalloc hands a
    // new structure of the same size the old addresses.  Positions rewritten IN PLACE behind the same pointers (only possible through this
    // raw-pointer entry; the reference never moves a site) need dkmc_reset_pair_sum_cache().
    auto &key = g_pw_key;
    const int host_rebuild = key.x != x || key.y != y || key.z != z || key.lattice != lattice || key.perm != pperm || key.cells != cells || key.N != N || key.pbc != pbc || key.cut != e.pair_cut;
    if (host_rebuild) { HIPCHK(hipMemsetAsync(grid, 0, sizeof(PwGrid), st)); key = {x, y, z, lattice, pperm, cells, N, pbc, e.pair_cut, 1, PW_MAXCELL}; }
    if (key.possible) hipLaunchKernelGGL(k_pw_grid, dim3(1), dim3(256), 0, st, lattice, sigma, e.pair_cut, pbc, (const int *)cnt, grid, tcount, ccount, cursor, host_rebuild);
    if (host_rebuild) {
        PwGrid hg{};
        HIPCHK(hipMemcpyAsync(&hg, grid, sizeof(PwGrid), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        key.possible = e.pair_cut > 0.0 && (hg.ny >= 2 * hg.ry + 1 || hg.nz >= 2 * hg.rz + 1) && std::min(2 * hg.ry + 1, hg.ny) * std::min(2 * hg.rz + 1, hg.nz) <= 32;
        key.ncell = std::max(1, hg.ny * hg.nz);
        if (!key.possible) HIPCHK(hipMemsetAsync(grid, 0, sizeof(PwGrid), st));          // use = 0 for good: k_pairwise does every call
    }
    // launch sizes do not rely on the host's copy of the column count (the device rebuilds its grid when the box changes): the kernels stop
    // at the device's own counts
    const int cell_blocks = (N + PW_SITES - 1) / PW_SITES + PW_MAXCELL;      // upper bound of the chunks of 64 sites of one column
    if (key.possible) {
        hipLaunchKernelGGL(k_pw_bin, dim3(blocks), dim3(256), 0, st, N, y, z, (const PwGrid *)grid, (const ChargedSite *)list, (const int *)cnt, site_cell, ccell, tcount, ccount);
        hipLaunchKernelGGL(k_pw_offsets, dim3(1), dim3(PW_MAXCELL), 0, st, grid, (const int *)tcount, (const int *)ccount, tstart, cstart, chunk0);
        hipLaunchKernelGGL(k_pw_perm, dim3(blocks), dim3(256), 0, st, N, (const PwGrid *)grid, (const int *)site_cell, (const int *)tstart, cursor, pperm);
        hipLaunchKernelGGL(k_pw_perm_x, dim3(PW_MAXCELL), dim3(256), 0, st, (const PwGrid *)grid, x, (const int *)tstart, (const int *)pperm, pperm2);
        hipLaunchKernelGGL(k_pw_perm_sort, dim3(PW_MAXCELL), dim3(256), 0, st, (const PwGrid *)grid, x, (const int *)tstart, pperm2);
        hipLaunchKernelGGL(k_pw_partition, dim3(PW_MAXCELL), dim3(256), 0, st, (const PwGrid *)grid, (const ChargedSite *)list, (const int *)cnt, (const int *)ccell,
                           (const int *)cstart, clist2);
        hipLaunchKernelGGL(k_pw_partition_x, dim3(PW_MAXCELL), dim3(64), 0, st, (const PwGrid *)grid, (const int *)cstart, (const ChargedSite *)clist2, clist3, cxoff);
    }
    const int *cells_in_use = &grid->use;
    if (comm_attached() && comm_nranks() > 1) {
        // SURVEY 8(e), "pairwise Poisson": the rows are independent -- every rank holds the (replicated) charged list and sums the
        // sites of its slab; one in-place all-gather hands every rank every potential.  Same arithmetic per site as on one GPU:
        // bit-identical.
        const int nr = comm_nranks(), me = comm_rank();
        const int chunk = ((N + nr - 1) / nr + PW_SITES - 1) / PW_SITES * PW_SITES;
        double *xbuf = (double *)scratch(S_CG_XCHG, (size_t)nr * chunk * 8);
        if (!xbuf) return e.err_code;
        const int lo = std::min(N, me * chunk), hi = std::min(N, lo + chunk);
        if (hi > lo)
            hipLaunchKernelGGL(k_pairwise, dim3((hi - lo + PW_SITES - 1) / PW_SITES), dim3(PW_NT), 0, st, hi, x, y, z, lattice, pbc, sigma, k, list, cnt, xbuf, d_ne, lo, e.pair_cut, cells_in_use);
        if (hi > lo && key.possible)
            hipLaunchKernelGGL(k_pairwise_cells, dim3(cell_blocks), dim3(PW_NT), 0, st, N, x, y, z, pbc, sigma, k, (const PwGrid *)grid, (const int *)tstart, (const int *)cstart,
                               (const int *)chunk0, (const int *)pperm2, (const ChargedSite *)clist3, (const int *)cxoff, xbuf, d_ne, lo, hi, e.pair_cut);
        KCHK();
        rc = comm_allgather_f64(xbuf, (size_t)chunk); if (rc) return rc;
        HIPCHK(hipMemcpyAsync(out, xbuf, (size_t)N * 8, hipMemcpyDeviceToDevice, st));
    } else {
        hipLaunchKernelGGL(k_pairwise, dim3((N + PW_SITES - 1) / PW_SITES), dim3(PW_NT), 0, st, N, x, y, z, lattice, pbc, sigma, k, list, cnt, out, d_ne, 0, e.pair_cut, cells_in_use);
        if (key.possible)
            hipLaunchKernelGGL(k_pairwise_cells, dim3(cell_blocks), dim3(PW_NT), 0, st, N, x, y, z, pbc, sigma, k, (const PwGrid *)grid, (const int *)tstart, (const int *)cstart,
                               (const int *)chunk0, (const int *)pperm2, (const ChargedSite *)clist3, (const int *)cxoff, out, d_ne, 0, N, e.pair_cut);
        KCHK();
    }
    e.stats.pair_ms = 0.0;
    if (e.profiling) {
        HIPCHK(hipEventRecord(evp[1], st));
        HIPCHK(hipEventSynchronize(evp[1]));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, evp[0], evp[1]));
        e.stats.pair_ms = ms;
        unsigned long long ne[2] = {0, 0};
        HIPCHK(hipMemcpy(ne, d_ne, 16, hipMemcpyDeviceToHost));
        e.stats.pair_evaluated = (long long)ne[0]; e.stats.pair_tested = (long long)ne[1];
    }
    return 0;
}
