// potential.hip -- charge rule, K sparsity + assembly, background potential / CB edge solves,
// screened-Coulomb pair sum.  Replaces potential_solver_gpu.cu (live parts) and the K-pattern
// builders of iterative_solvers_gpu.cu.
#include "common.h"

int kcg_assemble_and_solve(int cb, int m, int N_left, const int *element, const int *charge, MetalSet ms, double high_G, double low_G,
                           const int *rp, const int *ci, int nnz, const int *lrp, const int *lci, const int *rrp, const int *rci,
                           double VL, double VR, double *y, int *iters_out, double *rr_out);
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
struct KPatInfo { const int *rp; int m, N_left; };
static KPatInfo g_kpat[64]; static int g_kpat_next = 0;          // ring: the 64 most recently built patterns
static void kpat_register(const int *rp, int m, int N_left)
{
    for (auto &k : g_kpat) if (k.rp == rp) { k.m = m; k.N_left = N_left; return; }
    g_kpat[g_kpat_next++ % 64] = KPatInfo{rp, m, N_left};
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
        for (auto &k : g_kpat) if (k.rp == *pp) k = KPatInfo{nullptr, 0, 0};
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
        if (*rps[b]) (void)hipFree(*rps[b]);
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
    kpat_register(buf->Device_row_ptr_d, m, N_left);
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
                                  buf->contact_right_col_indices, VL, VR, field + N_left, iters, rr);
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
                                                    double *__restrict__ out, unsigned long long *__restrict__ nevaluated, int i0, double xcut)
{
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
    int neval = 0;
    for (int base = 0; base < nc; base += PW_NT) {
        const int n = min(PW_NT, nc - base);
        __syncthreads();
        if (threadIdx.x < n) tile[threadIdx.x] = list[base + threadIdx.x];
        __syncthreads();
        if (i < N) {
            int c = w;
            for (; c + 4 < n; c += 8) {
                v0 += pw_term(xi, yi, zi, tile[c], i, laty, latz, pbc, sigma, kk, cut2, neval);
                v1 += pw_term(xi, yi, zi, tile[c + 4], i, laty, latz, pbc, sigma, kk, cut2, neval);
            }
            if (c < n) v0 += pw_term(xi, yi, zi, tile[c], i, laty, latz, pbc, sigma, kk, cut2, neval);
        }
    }
    partial[w][lane] = v0 + v1;
    neval = wave_sum_all_i(neval);
    if (lane == 0 && neval) atomicAdd(nevaluated, (unsigned long long)neval);
    __syncthreads();
    if (w == 0 && i < N) out[i] = (partial[0][lane] + partial[1][lane]) + (partial[2][lane] + partial[3][lane]);
}

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
    unsigned long long *d_ne = (unsigned long long *)(cnt + 4);
    HIPCHK(hipMemsetAsync(d_ne, 0, 8, st));
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
            hipLaunchKernelGGL(k_pairwise, dim3((hi - lo + PW_SITES - 1) / PW_SITES), dim3(PW_NT), 0, st, hi, x, y, z, lattice, pbc, sigma, k, list, cnt, xbuf, d_ne, lo, e.pair_cut);
        KCHK();
        rc = comm_allgather_f64(xbuf, (size_t)chunk); if (rc) return rc;
        HIPCHK(hipMemcpyAsync(out, xbuf, (size_t)N * 8, hipMemcpyDeviceToDevice, st));
    } else {
        hipLaunchKernelGGL(k_pairwise, dim3((N + PW_SITES - 1) / PW_SITES), dim3(PW_NT), 0, st, N, x, y, z, lattice, pbc, sigma, k, list, cnt, out, d_ne, 0, e.pair_cut);
        KCHK();
    }
    e.stats.pair_ms = 0.0;
    if (e.profiling) {
        HIPCHK(hipEventRecord(evp[1], st));
        HIPCHK(hipEventSynchronize(evp[1]));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, evp[0], evp[1]));
        e.stats.pair_ms = ms;
        unsigned long long ne = 0;
        HIPCHK(hipMemcpy(&ne, d_ne, 8, hipMemcpyDeviceToHost));
        e.stats.pair_evaluated = (long long)ne;
    }
    return 0;
}
