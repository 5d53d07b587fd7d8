// current.hip -- current / dissipated-power solve.  Replaces update_power_gpu_sparse
// (current_solver_gpu.cu:854-1147) together with Assemble_X_sparsity / Assemble_X2
// (iterative_solvers_gpu.cu:1909-1983, 2113-2156) and the kernels they launch.
//
// X is the conductance matrix over nodes {0: extraction driver, 1: injection driver, a+2: atom a};
// the last atom is the ground and is dropped (Nsub = N_atom + 1 rows).  The reference rebuilds its
// sparsity every step with two O(N_atom^2) scans (thread per row over ALL columns).  Only two kinds
// of entries exist: neighbour pairs -- already in the padded neighbour index -- and tunnelling pairs
// among the set S = {vacancies} U {inner-contact metals}.  So the pattern is built from the
// neighbour rows (O(N_atom * nn)) plus an |S| x |S| sweep done by one workgroup per S-row with the
// S arrays streamed through coalesced loads; rows come out column-sorted exactly as the reference's.
#include "common.h"
#include "xshared.h"
#include <vector>
#include <stdlib.h>

// tiled X (xt.hip)
int xt_assemble_and_solve(dkmc_gpubuf *buf, const XParams &P, int ns, const SEntry *S, const int *aneigh, const int *ancnt, const int *aflag,
                          const int *srank, const int *atom_site, double *rhs, double *y, int *iters_out, double *rr_out, double *yaux, int yaux_valid);
int xt_power(dkmc_gpubuf *buf, const XParams &P, const int *aflag, const int *atom_site, const double *m, double Vd, double alpha);
const xrp_t *xt_xs_rp(); const int *xt_xs_col(); const double *xt_xs_val(); bool xt_valid();
int xt_export_csr(int *rows_out, long long *nnz_out, int *h_rp, int *h_col, double *h_data);

int cg_solve_jacobi64(double *a, const long long *rp, const int *ci, long long nnz, int m, double *x, double *y,
                      int uniform_rows, const int *srank, int ns, int *iters_out, double *rr_out);

// ---- cache maintenance ------------------------------------------------------------------------------------------------
// flags[0] = 1 if the static tables (metal ranks / metal CB snapshot) no longer match the current atoms
__global__ void k_tc_validate(int Na, const int *__restrict__ aflag, const double *__restrict__ acb, const int *__restrict__ mrank_atom,
                              const double *__restrict__ metal_cb, int *flags)
{
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= Na) return;
    const bool mp = aflag[a] & AF_MP_PAT;
    const int r = mrank_atom[a];
    if (mp != (r >= 0) || (mp && metal_cb[r] != acb[a])) flags[0] = 1;
}
__global__ void k_tc_mflags(int Na, const int *__restrict__ aflag, int *f)
{
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a < Na) f[a] = (aflag[a] & AF_MP_PAT) ? 1 : 0;
}
__global__ void k_tc_mtables(int Na, const int *__restrict__ f, const int *__restrict__ off, const double *__restrict__ acb,
                             int *mrank_atom, double *metal_cb, int *metal_atom)
{
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= Na) return;
    if (f[a]) { mrank_atom[a] = off[a]; metal_cb[off[a]] = acb[a]; metal_atom[off[a]] = a; } else mrank_atom[a] = -1;
}
// every vacancy of S gets a cache row; rows that are new (or whose CB no longer matches) are queued for filling.
// ctr: [0] rows in use, [1] overflow flag, [2] rows queued
// [k_lo, k_hi): S ranks whose vacancies get a row (everything for the main part; this rank's column windows for part A of a sharded solve)
__global__ void k_tc_assign(int ns, const SEntry *__restrict__ S, const int *__restrict__ atom_site, int cap,
                            int *slot_of_site, double *slot_cb, int *ctr, int2 *queue, int k_lo, int k_hi)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ns || k < k_lo || k >= k_hi) return;
    const SEntry e = S[k];
    if (!(e.flag & AF_V)) return;
    const int site = atom_site[e.idx];
    int slot = slot_of_site[site];
    bool fill = false;
    if (slot < 0) {
        slot = atomicAdd(&ctr[0], 1);
        if (slot >= cap) { ctr[1] = 1; return; }
        slot_of_site[site] = slot; fill = true;
    } else if (slot_cb[slot] != e.cb) fill = true;
    if (fill) { slot_cb[slot] = e.cb; const int q = atomicAdd(&ctr[2], 1); queue[q] = make_int2(slot, e.idx); }
}
// one thread per (queued row, metal): the same integral the direct path evaluates
// columns col_lo ... col_lo + nM - 1 of the cache (nM = the part's column count)
__global__ __launch_bounds__(256) void k_tc_fill(XParams P, int nq, const int2 *__restrict__ queue, int nM, int col_lo, const int *__restrict__ metal_atom,
                                                 const double *__restrict__ ax, const double *__restrict__ ay, const double *__restrict__ az,
                                                 const double *__restrict__ acb, double *__restrict__ vals)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    const int qi = blockIdx.y;
    if (r >= nM || qi >= nq) return;
    const int2 q = queue[qi];
    const int a = q.y, b = metal_atom[col_lo + r];
    const double prefac = -(sqrt(2 * P.m_e) / DKMC_HBAR) * (2.0 / 3.0);
    const double dA = site_dist(ax[a], ay[a], az[a], ax[b], ay[b], az[b], P.laty, P.latz, P.pbc);
    const double drop = fabs(acb[a] - acb[b]);
    vals[(size_t)q.x * nM + r] = (drop > P.tol) ? wkb_T(1, 1e-10 * dA, drop, prefac, P.V0) : 0.0;
}
// sharded solve: where S splits into left-contact metals | vacancies | right-contact metals, and which right-contact columns sit in this
// rank's windows.  out[0] = atom index of the first vacancy of S (Na if none), out[1] / out[2] = smallest / largest + 1 metal column whose S
// rank lies in [c_lo, c_hi) among the columns at or beyond nL (the caller passes nL from a first launch with c range empty).
__global__ void k_tc_split(int ns, const SEntry *__restrict__ S, const int *__restrict__ mrank_atom, int nL, int c_lo, int c_hi, int *out)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ns) return;
    const SEntry e = S[k];
    if (e.flag & AF_V) atomicMin(&out[0], e.idx);
    const int mr = mrank_atom[e.idx];
    if (mr >= nL && k >= c_lo && k < c_hi) { atomicMin(&out[1], mr); atomicMax(&out[2], mr + 1); }
}
// per step: a right-contact column of this rank's windows that the part does not hold -> flags[0] (the part is re-sized)
__global__ void k_tc_check_cols(int ns, const SEntry *__restrict__ S, const int *__restrict__ mrank_atom, int nL, int c_lo, int c_hi, int col_lo, int ncols, int *flags)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ns || k < c_lo || k >= c_hi) return;
    const int mr = mrank_atom[S[k].idx];
    if (mr >= nL && (mr < col_lo || mr >= col_lo + ncols)) flags[0] = 1;
}
__global__ void k_fill_i32(int *p, int n, int v) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = v; }

struct TCacheState {
    int N = 0, Na = 0, nM = 0, cap = 0, valid = 0;
    int *slot_of_site = nullptr, *mrank_atom = nullptr, *metal_atom = nullptr, *ctr = nullptr;
    double *metal_cb = nullptr, *slot_cb = nullptr, *vals = nullptr;
    int2 *queue = nullptr;
    int col_lo = 0, ncols = 0;          // columns of the main part (one GPU / CSR X: all nM)
    // part A of a sharded solve on the tiled X (TCacheView): the nL left-contact columns for the vacancies of this rank's windows
    int sharded = 0, nL = 0, capA = 0;
    int nL_split = 0;                   // left-contact metals in front of the first vacancy of S, whether part A is held or not (nL = 0: not held)
    int *slotA_of_site = nullptr, *ctrA = nullptr; double *slotA_cb = nullptr, *valsA = nullptr; int2 *queueA = nullptr;
    size_t bytes() const { return ((size_t)cap * ncols + (size_t)capA * nL) * 8; }
};
// Solver state that outlives a call -- the coefficient cache and the private warm-start copy -- is kept PER GPUBuffers (keyed by
// its site_x array), so that several devices in one process (e.g. one per crossbar cell) do not share or thrash it.  Up to 8
// buffers, least recently used one evicted.  The engine's scratch buffers are shared: they carry nothing across calls.
struct XBufState { const void *key = nullptr; TCacheState tc; double *warm = nullptr; int warm_n = 0; unsigned long long stamp = 0;
                   double *warm_aux = nullptr; int warm_aux_n = 0, warm_aux_valid = 0;      // [warm_aux_n][16]: unscaled solutions of the block-CG's auxiliary columns (dkmc_set_x_aux_warm)
                   double lat[3] = {0, 0, 0}; bool lat_ok = false; };       // lattice: constant per GPUBuffers, fetched once
static XBufState g_states[8];
static XBufState *g_cur = &g_states[0];
static unsigned long long g_stamp = 0;
#define g_tc (g_cur->tc)
#define g_warm (g_cur->warm)
#define g_warm_n (g_cur->warm_n)

static XBufState *xstate_find(const void *key) { for (auto &st : g_states) if (st.key == key && key) return &st; return nullptr; }
void tcache_invalidate(const void *key) { if (XBufState *st = xstate_find(key)) st->tc.valid = 0; }     // new bias point (potential.hip)

static void tc_release();
// new structure behind the same buffer (initialize_sparsity) or buffer freed: drop everything kept for it
void pairsum_invalidate();      // potential.hip: the cached site grouping of the pair sum is keyed on the same arrays
void xstate_reset(const void *key)
{
    pairsum_invalidate();
    XBufState *st = xstate_find(key);
    if (!st) return;
    XBufState *save = g_cur; g_cur = st;
    (void)hipStreamSynchronize(eng().stream);
    tc_release();
    if (st->warm) (void)hipFree(st->warm);
    if (st->warm_aux) (void)hipFree(st->warm_aux);
    *st = XBufState();
    g_cur = save;
}
static void xstate_select(const void *key)
{
    XBufState *st = xstate_find(key);
    if (!st) {
        st = &g_states[0];
        for (auto &c : g_states) { if (!c.key) { st = &c; break; } if (c.stamp < st->stamp) st = &c; }
        if (st->key) xstate_reset(st->key);
        st->key = key;
    }
    st->stamp = ++g_stamp;
    g_cur = st;
}
static void tc_release()
{
    void *ptrs[] = { g_tc.slot_of_site, g_tc.mrank_atom, g_tc.metal_atom, g_tc.ctr, g_tc.metal_cb, g_tc.slot_cb, g_tc.vals, g_tc.queue,
                     g_tc.slotA_of_site, g_tc.ctrA, g_tc.slotA_cb, g_tc.valsA, g_tc.queueA };
    for (void *q : ptrs) if (q) (void)hipFree(q);
    g_tc = TCacheState();
}

// (re)build the static tables and empty the cache.  share != nullptr: a sharded solve on the tiled X -- {c_lo, c_hi} = the S ranks of this
// rank's column windows; the cache then holds what this rank's tiles read (TCacheView), sized with a window of slack on either side so that
// the executed events (which shift S ranks by one at a time) do not re-size it every step.
#define TC_SLACK 256
static int tc_reset(const XParams &P, int N, const int *aflag, const double *acb, int n_vac, int ns, const SEntry *S, const int *share)
{
    Engine &e = eng(); hipStream_t st = e.stream;
    HIPCHK(hipStreamSynchronize(st));
    tc_release();
    const int Na = P.Na;
    int *f = (int *)scratch(S_MISC0, (size_t)Na * 4), *off = (int *)scratch(S_MISC1, (size_t)(Na + 4) * 4);
    if (!f || !off) return e.err_code;
    const int nb = (Na + 255) / 256;
    hipLaunchKernelGGL(k_tc_mflags, dim3(nb), dim3(256), 0, st, Na, aflag, f);
    int rc = dkmc_exclusive_scan_i32(f, off, Na, off + Na); if (rc) return rc;
    int nM = 0;
    HIPCHK(hipMemcpyAsync(&nM, off + Na, sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (nM <= 0) { g_tc.valid = 0; return 0; }
    const int n_vacancies = n_vac - nM > 64 ? n_vac - nM : 64;          // n_vac carries |S| = vacancies + inner-contact metals
    g_tc.N = N; g_tc.Na = Na; g_tc.nM = nM;
    HIPCHK(hipMalloc((void **)&g_tc.mrank_atom, (size_t)Na * 4));
    HIPCHK(hipMalloc((void **)&g_tc.metal_atom, (size_t)nM * 4));
    HIPCHK(hipMalloc((void **)&g_tc.metal_cb, (size_t)nM * 8));
    HIPCHK(hipMalloc((void **)&g_tc.ctr, 8 * sizeof(int)));
    hipLaunchKernelGGL(k_tc_mtables, dim3(nb), dim3(256), 0, st, Na, (const int *)f, (const int *)off, acb, g_tc.mrank_atom, g_tc.metal_cb, g_tc.metal_atom);
    // ---- which columns, which rows ----
    int col_lo = 0, ncols = nM, nL = 0, nA = 0;
    if (share && ns > 0) {
        int h[3] = {Na, 0x7fffffff, 0};
        int *d3 = g_tc.ctr + 4;
        HIPCHK(hipMemcpyAsync(d3, h, sizeof(h), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_tc_split, dim3((ns + 255) / 256), dim3(256), 0, st, ns, S, (const int *)g_tc.mrank_atom, nM, 0, 0, d3);      // first vacancy only
        int first_vac = Na;
        HIPCHK(hipMemcpyAsync(&first_vac, d3, sizeof(int), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (first_vac < Na) { HIPCHK(hipMemcpy(&nL, off + first_vac, sizeof(int), hipMemcpyDeviceToHost)); } else nL = nM;      // metals in front of the first vacancy
        const int c_lo = std::max(0, share[0] - TC_SLACK), c_hi = share[1] > 0x7fffffff - TC_SLACK ? share[1] : share[1] + TC_SLACK;
        h[0] = Na; h[1] = 0x7fffffff; h[2] = 0;
        HIPCHK(hipMemcpyAsync(d3, h, sizeof(h), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_tc_split, dim3((ns + 255) / 256), dim3(256), 0, st, ns, S, (const int *)g_tc.mrank_atom, nL, c_lo, c_hi, d3);
        HIPCHK(hipMemcpyAsync(h, d3, sizeof(h), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (h[2] > h[1]) { col_lo = h[1]; ncols = h[2] - h[1]; } else { col_lo = nL; ncols = 0; }
        // rows of part A: the vacancies of this rank's windows (+ slack), at most all of them
        nA = std::min(n_vacancies, std::max(0, std::min(c_hi, ns) - c_lo));
        g_tc.sharded = 1;
    }
    size_t cap = (size_t)4 * n_vacancies + 256, capA = nL > 0 ? (size_t)2 * nA + 256 : 0;
    // budget: a third of the device memory that is free now, between 8 and 128 GiB (1.9e6 sites need 25 GB for one row per vacancy,
    // 3.8e6 sites 106 GB on ONE GPU; the tiles of a rank's share come on top)
    size_t mem_free = 0, mem_total = 0;
    if (hipMemGetInfo(&mem_free, &mem_total) != hipSuccess) mem_free = 0;
    size_t budget = std::min((size_t)128 << 30, std::max((size_t)8 << 30, mem_free / 3));
    if (e.tcache_budget >= 0) budget = (size_t)e.tcache_budget;        // dkmc_set_tcache_budget (tests: force the uncached path)
    const int nL_split = nL;
    if (capA * nL * 8 > budget / 2) { capA = 0; nL = 0; }              // part A does not fit: its pairs are integrated directly
    const size_t left = budget - capA * nL * 8;
    if (ncols > 0 && cap * ncols * 8 > left) cap = left / ((size_t)ncols * 8);
    if (ncols > 0 && cap < (size_t)n_vacancies + 16) { g_tc.valid = 0; return 0; }         // does not fit: run uncached
    if (ncols == 0) cap = 16;
    g_tc.cap = (int)cap; g_tc.col_lo = col_lo; g_tc.ncols = ncols; g_tc.nL = nL; g_tc.nL_split = nL_split; g_tc.capA = (int)capA;
    HIPCHK(hipMalloc((void **)&g_tc.slot_of_site, (size_t)N * 4));
    HIPCHK(hipMalloc((void **)&g_tc.slot_cb, cap * 8));
    HIPCHK(hipMalloc((void **)&g_tc.vals, std::max<size_t>(cap * ncols, 1) * 8));
    HIPCHK(hipMalloc((void **)&g_tc.queue, cap * sizeof(int2)));
    HIPCHK(hipMemsetAsync(g_tc.ctr, 0, 8 * sizeof(int), st));
    hipLaunchKernelGGL(k_fill_i32, dim3((N + 255) / 256), dim3(256), 0, st, g_tc.slot_of_site, N, -1);
    if (nL > 0) {
        HIPCHK(hipMalloc((void **)&g_tc.slotA_of_site, (size_t)N * 4));
        HIPCHK(hipMalloc((void **)&g_tc.slotA_cb, capA * 8));
        HIPCHK(hipMalloc((void **)&g_tc.valsA, capA * nL * 8));
        HIPCHK(hipMalloc((void **)&g_tc.queueA, capA * sizeof(int2)));
        HIPCHK(hipMalloc((void **)&g_tc.ctrA, 4 * sizeof(int)));
        HIPCHK(hipMemsetAsync(g_tc.ctrA, 0, 4 * sizeof(int), st));
        hipLaunchKernelGGL(k_fill_i32, dim3((N + 255) / 256), dim3(256), 0, st, g_tc.slotA_of_site, N, -1);
    }
    KCHK();
    g_tc.valid = 1;
    return 0;
}

// per step: validate, assign rows to the current vacancies, fill the new rows.  Leaves g_tc.valid = 0 if the cache cannot be used.
// share: see tc_reset (nullptr = the whole cache: one GPU, CSR X)
static int tc_update(const XParams &P, int N, int ns, const SEntry *S, const int *aflag, const int *atom_site,
                     const double *ax, const double *ay, const double *az, const double *acb, int n_vac_hint, const int *share = nullptr)
{
    Engine &e = eng(); hipStream_t st = e.stream;
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (!g_tc.valid || g_tc.N != N || g_tc.Na != P.Na || g_tc.sharded != (share ? 1 : 0)) {
            int rc = tc_reset(P, N, aflag, acb, n_vac_hint, ns, S, share); if (rc) return rc; if (!g_tc.valid) return 0;
        }
        int h[4] = {0, 0, 0, 0}, hA[4] = {0, 0, 0, 0};
        int *flags = g_tc.ctr + 3;
        HIPCHK(hipMemsetAsync(g_tc.ctr + 1, 0, 3 * sizeof(int), st));
        hipLaunchKernelGGL(k_tc_validate, dim3((P.Na + 255) / 256), dim3(256), 0, st, P.Na, aflag, acb, (const int *)g_tc.mrank_atom,
                           (const double *)g_tc.metal_cb, flags);
        if (share) hipLaunchKernelGGL(k_tc_check_cols, dim3((ns + 255) / 256), dim3(256), 0, st, ns, S, (const int *)g_tc.mrank_atom, g_tc.nL_split, share[0], std::min(share[1], ns),
                                      g_tc.col_lo, g_tc.ncols, flags);      // (the TRUE left-contact count: with part A not held nL is 0 and every left-contact metal would read as a missing right-contact column)
        if (g_tc.ncols > 0)
            hipLaunchKernelGGL(k_tc_assign, dim3((ns + 255) / 256), dim3(256), 0, st, ns, S, atom_site, g_tc.cap, g_tc.slot_of_site, g_tc.slot_cb,
                               g_tc.ctr, g_tc.queue, 0, ns);
        if (g_tc.nL > 0) {
            HIPCHK(hipMemsetAsync(g_tc.ctrA + 1, 0, 3 * sizeof(int), st));
            hipLaunchKernelGGL(k_tc_assign, dim3((ns + 255) / 256), dim3(256), 0, st, ns, S, atom_site, g_tc.capA, g_tc.slotA_of_site, g_tc.slotA_cb,
                               g_tc.ctrA, g_tc.queueA, share[0], std::min(share[1], ns));
            HIPCHK(hipMemcpyAsync(hA, g_tc.ctrA, 4 * sizeof(int), hipMemcpyDeviceToHost, st));
        }
        HIPCHK(hipMemcpyAsync(h, g_tc.ctr, 4 * sizeof(int), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (h[3] || h[1] || hA[1]) { g_tc.valid = 0; continue; }      // stale tables, a window that left the part, or full: rebuild once
        if (h[2] > 0 && g_tc.ncols > 0)
            hipLaunchKernelGGL(k_tc_fill, dim3((g_tc.ncols + 255) / 256, h[2]), dim3(256), 0, st, P, h[2], (const int2 *)g_tc.queue, g_tc.ncols, g_tc.col_lo,
                               (const int *)g_tc.metal_atom, ax, ay, az, acb, g_tc.vals);
        if (hA[2] > 0)
            hipLaunchKernelGGL(k_tc_fill, dim3((g_tc.nL + 255) / 256, hA[2]), dim3(256), 0, st, P, hA[2], (const int2 *)g_tc.queueA, g_tc.nL, 0,
                               (const int *)g_tc.metal_atom, ax, ay, az, acb, g_tc.valsA);
        KCHK();
        e.stats.tcache_bytes = (long long)g_tc.bytes();
        return 0;
    }
    g_tc.valid = 0;
    e.stats.tcache_bytes = 0;
    return 0;
}
static void tc_view(TCacheView *TC)
{
    TC->enabled = g_tc.valid; TC->nM = g_tc.nM; TC->cap = g_tc.cap; TC->slot_of_site = g_tc.slot_of_site; TC->mrank_atom = g_tc.mrank_atom;
    TC->vals = g_tc.vals; TC->col_lo = g_tc.col_lo; TC->ncols = g_tc.ncols;
    TC->nL = g_tc.valid ? g_tc.nL : 0; TC->slotA_of_site = g_tc.slotA_of_site; TC->valsA = g_tc.valsA;
}
// the cache as the tiles of THIS rank read it (xt.hip calls this once its share of the tile list is known)
int tc_prepare_tiled(const XParams &P, dkmc_gpubuf *buf, int ns, const SEntry *S, const int *aflag, const int *srank, const int *atom_site,
                     int c_lo, int c_hi, int sharded, TCacheView *out)
{
    (void)srank;
    const int share[2] = {c_lo, c_hi};
    int rc = tc_update(P, buf->N_, ns, S, aflag, atom_site, buf->atom_x, buf->atom_y, buf->atom_z, buf->atom_CB_edge, ns, sharded ? share : nullptr);
    if (rc) return rc;
    tc_view(out);
    return 0;
}

// ---- post-solve ------------------------------------------------------------------------------------
__global__ void k_set_rhs(double *m, int n, double loop_G, double Vd)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) m[i] = (i == 0) ? -loop_G * Vd : (i == 1) ? loop_G * Vd : 0.0;
}
__global__ void k_scale(double *v, int n, double s)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = v[i] * s;
}
// get_imacro_sparse (current_solver_gpu.cu:781-821): single block, fixed-order reduction
__global__ __launch_bounds__(256) void k_imacro(const double *__restrict__ xv, const xrp_t *__restrict__ rp, const int *__restrict__ ci,
                                                const double *__restrict__ m, double *imacro)
{
    __shared__ double red[4];
    const xrp_t row_start = rp[1] + 2, row_end = rp[2];
    double s = 0.0;
    for (xrp_t p = row_start + threadIdx.x; p < row_end; p += 256) { const int c = ci[p]; if (c >= 2) s += xv[p] * (m[c] - m[1]); }
    const double t = block_sum_all<256>(s, red);
    if (threadIdx.x == 0) *imacro = t;
}
// update_m (current_solver_gpu.cu:447-457, 1044-1047): m += |min(m[2 .. Na+1])|
// (min is exact in any order: per-workgroup minima, then every workgroup of the shift kernel takes the minimum of those)
#define MINM_BLOCKS 128
__global__ __launch_bounds__(256) void k_min_m(const double *__restrict__ m, int lo, int hi, double *__restrict__ part)
{
    __shared__ double red[256];
    double v = INFINITY;
    for (int i = lo + (int)(blockIdx.x * 256 + threadIdx.x); i < hi; i += MINM_BLOCKS * 256) v = fmin(v, m[i]);
    red[threadIdx.x] = v; __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) red[threadIdx.x] = fmin(red[threadIdx.x], red[threadIdx.x + s]); __syncthreads(); }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(256) void k_shift(double *m, int n, const double *__restrict__ part)
{
    __shared__ double red[MINM_BLOCKS];
    if (threadIdx.x < MINM_BLOCKS) red[threadIdx.x] = part[threadIdx.x];
    __syncthreads();
    for (int s = MINM_BLOCKS / 2; s > 0; s >>= 1) { if (threadIdx.x < s) red[threadIdx.x] = fmin(red[threadIdx.x], red[threadIdx.x + s]); __syncthreads(); }
    const double shift = fabs(red[0]);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) m[i] += shift;
}
// Dissipated power on X's pattern: host formula (current_solver.cpp:288-357) restricted to the atoms kept
// in the sparse system; set_ineg_sparse + reduce_rows_into_diag + SpMV + copy_pdisp fused (see SURVEY B8/B9
// for the index slips of the CUDA kernels this replaces).
template <int LPR>
__global__ __launch_bounds__(256) void k_power(int Na, int nrows, const int *__restrict__ rows, const xrp_t *__restrict__ rp,
                                               const int *__restrict__ ci, const double *__restrict__ xv, const double *__restrict__ m,
                                               double Vd, const int *__restrict__ aflag, const int *__restrict__ atom_site,
                                               double alpha, double *__restrict__ site_power)
{
    const int gpb = 256 / LPR, g = threadIdx.x / LPR, l = threadIdx.x % LPR;
    const int ridx = blockIdx.x * gpb + g;
    if (ridx >= nrows) return;
    const int i = rows[ridx];
    if (i < 2) return;
    const double mi = m[i];
    double p = 0.0;
    for (xrp_t q = rp[i] + l; q < rp[i + 1]; q += LPR) {
        const int c = ci[q];
        if (c < 2 || c == i) continue;
        const double ical = xv[q] * (mi - m[c]);
        double v = 0.0;
        if ((ical < 0 && Vd > 0) || (ical > 0 && Vd < 0)) v = -ical;
        p += v * (m[c] - mi);
    }
    p = group_sum<LPR>(p);
    const int a = i - 2;
    if (l == 0 && !(aflag[a] & AF_METAL)) site_power[atom_site[a]] = -1 * alpha * p;
}
// rank in S per system node (0, 1 = drivers: -1; a + 2 = atom a)
__global__ void k_node_srank(int Nsub, const int *__restrict__ srank, int *__restrict__ node_srank)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < Nsub) node_srank[i] = i < 2 ? -1 : srank[i - 2];
}
__global__ void k_iota_rows(int n, int *rows)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) rows[i] = i;
}

// state of the last call (for dkmc_get_last_X and the private warm start)
static int g_last_rows = 0; static long long g_last_nnz = 0; static bool g_last_tiled = false;

static int update_power_body(dkmc_gpubuf *buf, int n_src, int n_gnd, int nlc, double Vd, int pbc,
                             double high_G, double low_G, double loop_G, double G0, double tol, double nn_dist,
                             double m_e, double V0, int num_metals, double *h_imacro,
                             int heat_local, int heat_global, double alpha_disp);
extern "C" int dkmc_update_power_gpu_sparse(dkmc_gpubuf *buf, int n_src, int n_gnd, int nlc, double Vd, int pbc,
                                            double high_G, double low_G, double loop_G, double G0, double tol, double nn_dist,
                                            double m_e, double V0, int num_metals, double *h_imacro,
                                            int heat_local, int heat_global, double alpha_disp)
{
    // Sharded solve on the tiled X: every rank passes exactly one agreement point per call (comm_agree, in xt_assemble_and_solve, after
    // all local set-up and before the first collective).  A rank that fails BEFORE it (compaction, S, coefficient cache) reaches it
    // here instead, so that its peers learn of the failure rather than wait for a collective that never comes.
    const int agreed_before = comm_agree_count();
    const int rc = update_power_body(buf, n_src, n_gnd, nlc, Vd, pbc, high_G, low_G, loop_G, G0, tol, nn_dist, m_e, V0, num_metals, h_imacro,
                                     heat_local, heat_global, alpha_disp);
    if (rc && comm_attached() && eng().x_format != 0 && comm_agree_count() == agreed_before) (void)comm_agree(rc, "set-up of the current solve");
    return rc;
}
static int update_power_body(dkmc_gpubuf *buf, int n_src, int n_gnd, int nlc, double Vd, int pbc,
                             double high_G, double low_G, double loop_G, double G0, double tol, double nn_dist,
                             double m_e, double V0, int num_metals, double *h_imacro,
                             int heat_local, int heat_global, double alpha_disp)
{
    Engine &e = eng(); hipStream_t st = e.stream;
    const int N = buf->N_, nn = buf->nn_;
    xstate_select(buf->site_x);
    if (nn > 64) return dkmc_fail(11, "update_power: more than 64 neighbours per site is not supported", __FILE__, __LINE__);
    MetalSet ms = load_metals(buf->metal_types, num_metals);
    // ---- 1. atoms ----
    int *flag = (int *)scratch(S_AT_FLAG, (size_t)N * 4), *off = (int *)scratch(S_AT_OFSITE, (size_t)(N + 4) * 4);
    int *atom_site = (int *)scratch(S_AT_SITE, (size_t)N * 4 * 2);
    if (!flag || !off || !atom_site) return e.err_code;
    int *site_atom = atom_site + N;
    const int nb = (N + 255) / 256;
    hipLaunchKernelGGL(k_atom_flags, dim3(nb), dim3(256), 0, st, N, buf->site_element, flag);
    int rc = dkmc_exclusive_scan_i32(flag, off, N, off + N); if (rc) return rc;
    int Na = 0;
    HIPCHK(hipMemcpyAsync(&Na, off + N, sizeof(int), hipMemcpyDeviceToHost, st));
    hipLaunchKernelGGL(k_atom_gather, dim3(nb), dim3(256), 0, st, N, flag, off, buf->site_x, buf->site_y, buf->site_z, buf->site_charge,
                       buf->site_element, buf->site_CB_edge, buf->atom_x, buf->atom_y, buf->atom_z, buf->atom_charge, buf->atom_element,
                       buf->atom_CB_edge, atom_site, site_atom);
    HIPCHK(hipStreamSynchronize(st));
    if (Na < 3) return dkmc_fail(8, "update_power: fewer than 3 atoms", __FILE__, __LINE__);
    if (Na > buf->N_atom_) return dkmc_fail(9, "update_power: more atoms than GPUBuffers was sized for (N_atom_)", __FILE__, __LINE__);
    e.stats.N_atom = Na;
    const int Nsub = Na + 1;
    if (!g_cur->lat_ok) { HIPCHK(hipMemcpy(g_cur->lat, buf->lattice, 3 * sizeof(double), hipMemcpyDeviceToHost)); g_cur->lat_ok = true; }
    const double *lat = g_cur->lat;
    XParams P; P.Na = Na; P.nn = nn; P.n_src = n_src; P.n_gnd = n_gnd; P.nlc = nlc; P.pbc = pbc; P.tol = tol; P.nn_dist = nn_dist;
    P.high_G = high_G; P.low_G = low_G; P.loop_G = loop_G; P.m_e = m_e; P.V0 = V0; P.laty = lat[1]; P.latz = lat[2];

    // ---- 2. atom rows, class flags, S ----
    int *aneigh = (int *)scratch(S_AT_NEIGH, (size_t)Na * nn * 4);
    int *aflag = (int *)scratch(S_X_SFLAG, (size_t)Na * 4 * 4);
    SEntry *S = (SEntry *)scratch(S_X_SLIST, (size_t)Na * sizeof(SEntry));
    if (!aneigh || !aflag || !S) return e.err_code;
    int *ancnt = aflag + Na, *inS = aflag + 2 * Na, *srank = aflag + 3 * Na;
    int *soff = (int *)scratch(S_X_SRANK, (size_t)(Na + 4) * 4);
    if (!soff) return e.err_code;
    const int nba = (Na + 255) / 256;
    hipLaunchKernelGGL(k_atom_rows, dim3(nba), dim3(256), 0, st, P, buf->neigh_idx, atom_site, site_atom, buf->atom_element, buf->atom_charge,
                       ms, aflag, aneigh, ancnt, inS);
    rc = dkmc_exclusive_scan_i32(inS, soff, Na, soff + Na); if (rc) return rc;
    int ns = 0;
    HIPCHK(hipMemcpyAsync(&ns, soff + Na, sizeof(int), hipMemcpyDeviceToHost, st));
    hipLaunchKernelGGL(k_S_scatter, dim3(nba), dim3(256), 0, st, Na, inS, soff, aflag, buf->atom_CB_edge, S, srank);
    HIPCHK(hipStreamSynchronize(st));

    const int nbr = (Nsub + 255) / 256;
    double *rhs = (double *)scratch(S_X_RHS, (size_t)(Na + 2) * 8);
    if (!rhs) return e.err_code;
    double *m = buf->atom_virtual_potentials;
    const xrp_t *rp = nullptr; const int *col = nullptr; const double *data = nullptr;
    g_last_tiled = e.x_format != 0;
    if (e.x_format != 0) {
        // ---- 3-5 (tiled X, xt.hip): neighbour part as a small CSR, tunnelling block straight into symmetric tiles, solve ----
        // (the coefficient cache is brought up to date inside, once this rank's share of the tiles is known: tc_prepare_tiled)
        hipLaunchKernelGGL(k_set_rhs, dim3((Na + 2 + 255) / 256), dim3(256), 0, st, rhs, Na + 2, loop_G, Vd);
        if (e.current_warm_start == 1) {
            if (g_warm && g_warm_n == Nsub) HIPCHK(hipMemcpyAsync(m, g_warm, (size_t)Nsub * 8, hipMemcpyDeviceToDevice, st));
        }
        // the auxiliary columns of the block-CG start from the previous solve's solutions too (kept per GPUBuffers like the start vector itself)
        double *yaux = nullptr; int yaux_valid = 0;
        if (e.current_warm_start == 1 && e.x_aux_warm && e.x_block > 1) {
            if (g_cur->warm_aux_n != Nsub) {
                if (g_cur->warm_aux) (void)hipFree(g_cur->warm_aux);
                g_cur->warm_aux = nullptr; g_cur->warm_aux_n = 0; g_cur->warm_aux_valid = 0;
                HIPCHK(hipMalloc((void **)&g_cur->warm_aux, (size_t)Nsub * 16 * 8));
                g_cur->warm_aux_n = Nsub;
            }
            yaux = g_cur->warm_aux; yaux_valid = g_cur->warm_aux_valid;
        }
        if (yaux) g_cur->warm_aux_valid = 0;                    // (a solve that fails leaves nothing to start from)
        rc = xt_assemble_and_solve(buf, P, ns, S, aneigh, ancnt, aflag, srank, atom_site, rhs, m, &e.stats.cg_iters_X, &e.stats.cg_rr_X, yaux, yaux_valid);
        if (rc) return rc;
        if (yaux && e.stats.xb_width > 1 && !e.stats.xb_fallback) g_cur->warm_aux_valid = 1;
        rp = xt_xs_rp(); col = xt_xs_col(); data = xt_xs_val();
    } else {
        // ---- 3. sparsity: counts -> row_ptr -> columns ----
        int *cnt = (int *)scratch(S_X_CNT, (size_t)(Nsub + 4) * 4);
        xrp_t *rp_ = (xrp_t *)scratch(S_X_ROWPTR, (size_t)(Nsub + 4) * sizeof(xrp_t));
        if (!cnt || !rp_) return e.err_code;
        hipLaunchKernelGGL((k_xpat_plain<0>), dim3(nbr), dim3(256), 0, st, P, inS, aneigh, ancnt, cnt, (const xrp_t *)nullptr, (int *)nullptr);
        if (ns > 0) hipLaunchKernelGGL((k_xpat_S<0>), dim3(ns), dim3(XS_NT), 0, st, P, ns, S, aneigh, ancnt, cnt, (const xrp_t *)nullptr, (int *)nullptr);
        rc = dkmc_exclusive_scan_i32_i64(cnt, rp_, Nsub, rp_ + Nsub); if (rc) return rc;
        long long nnz = 0;
        HIPCHK(hipMemcpyAsync(&nnz, rp_ + Nsub, sizeof(long long), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (nnz <= 0) return dkmc_fail(10, "update_power: empty X", __FILE__, __LINE__);
        e.stats.X_nnz = nnz;
        int *col_ = (int *)scratch(S_X_COL, (size_t)nnz * 4);
        double *data_ = (double *)scratch(S_X_DATA, (size_t)nnz * 8), *data2 = (double *)scratch(S_X_DATA2, (size_t)nnz * 8);
        if (!col_ || !data_ || !data2) return e.err_code;
        hipLaunchKernelGGL((k_xpat_plain<1>), dim3(nbr), dim3(256), 0, st, P, inS, aneigh, ancnt, cnt, rp_, col_);
        if (ns > 0) hipLaunchKernelGGL((k_xpat_S<1>), dim3(ns), dim3(XS_NT), 0, st, P, ns, S, aneigh, ancnt, cnt, rp_, col_);
        // ---- 4. values ----
        rc = tc_update(P, N, ns, S, aflag, atom_site, buf->atom_x, buf->atom_y, buf->atom_z, buf->atom_CB_edge, ns); if (rc) return rc;
        TCacheView TC; tc_view(&TC);
        hipLaunchKernelGGL((k_xval<16>), dim3((Nsub + 15) / 16), dim3(256), 0, st, P, Nsub, S, 0, inS, rp_, col_, buf->atom_x, buf->atom_y, buf->atom_z,
                           aflag, buf->atom_CB_edge, data_, TC, (const int *)atom_site);
        if (ns > 0) hipLaunchKernelGGL((k_xval<64>), dim3((ns + 3) / 4), dim3(256), 0, st, P, ns, S, 1, inS, rp_, col_, buf->atom_x, buf->atom_y, buf->atom_z,
                                       aflag, buf->atom_CB_edge, data_, TC, (const int *)atom_site);
        KCHK();
        g_last_rows = Nsub; g_last_nnz = nnz;

        // ---- 5. solve X m = rhs (current_solver_gpu.cu:963-994) ----
        hipLaunchKernelGGL(k_set_rhs, dim3((Na + 2 + 255) / 256), dim3(256), 0, st, rhs, Na + 2, loop_G, Vd);
        HIPCHK(hipMemcpyAsync(data2, data_, (size_t)nnz * 8, hipMemcpyDeviceToDevice, st));
        if (e.current_warm_start == 1) {
            if (g_warm && g_warm_n == Nsub) HIPCHK(hipMemcpyAsync(m, g_warm, (size_t)Nsub * 8, hipMemcpyDeviceToDevice, st));
        }
        int *node_srank = (int *)scratch(S_X_SCB, (size_t)(Nsub + 4) * 4);
        if (!node_srank) return e.err_code;
        hipLaunchKernelGGL(k_node_srank, dim3(nbr), dim3(256), 0, st, Nsub, srank, node_srank);
        rc = cg_solve_jacobi64(data2, rp_, col_, nnz, Nsub, rhs, m, 0, node_srank, ns, &e.stats.cg_iters_X, &e.stats.cg_rr_X);
        if (rc) return rc;
        rp = rp_; col = col_; data = data_;
    }
    if (e.current_warm_start == 1) {
        if (g_warm_n != Nsub) { if (g_warm) (void)hipFree(g_warm); HIPCHK(hipMalloc((void **)&g_warm, (size_t)Nsub * 8)); g_warm_n = Nsub; }
        HIPCHK(hipMemcpyAsync(g_warm, m, (size_t)Nsub * 8, hipMemcpyDeviceToDevice, st));
    }
    // ---- 6. I_macro (:1015-1029) ----
    hipLaunchKernelGGL(k_scale, dim3((Na + 2 + 255) / 256), dim3(256), 0, st, m, Na + 2, G0);
    double *d_im = (double *)scratch(S_P_IMACRO, (2 + MINM_BLOCKS) * sizeof(double));     // I_macro, spare, per-workgroup minima of m
    if (!d_im) return e.err_code;
    hipLaunchKernelGGL(k_imacro, dim3(1), dim3(256), 0, st, data, rp, col, m, d_im);
    HIPCHK(hipMemcpyAsync(h_imacro, d_im, sizeof(double), hipMemcpyDeviceToHost, st));
    // ---- 7. dissipated power (:1041-1136) ----
    if (heat_local || heat_global) {
        double *minpart = d_im + 2;
        hipLaunchKernelGGL(k_min_m, dim3(MINM_BLOCKS), dim3(256), 0, st, (const double *)m, 2, Na + 2, minpart);
        hipLaunchKernelGGL(k_shift, dim3((Na + 2 + 255) / 256), dim3(256), 0, st, m, Na + 2, (const double *)minpart);
        if (e.x_format != 0) { rc = xt_power(buf, P, aflag, atom_site, m, Vd, alpha_disp); if (rc) return rc; }
        else {
            int *rows = (int *)scratch(S_MISC0, (size_t)Nsub * 4);
            if (!rows) return e.err_code;
            hipLaunchKernelGGL(k_iota_rows, dim3(nbr), dim3(256), 0, st, Nsub, rows);
            hipLaunchKernelGGL((k_power<64>), dim3((Nsub + 3) / 4), dim3(256), 0, st, Na, Nsub, rows, rp, col, data, m, Vd, aflag, atom_site,
                               alpha_disp, buf->site_power);
        }
    }
    KCHK();
    HIPCHK(hipStreamSynchronize(st));
    return e.err_code;
}

// ---- the private start vector of dkmc_set_current_warm_start(1), for snapshot / restart (io.py: the sidecar keeps it) -------------------
// get: *n_out = entries held for this buffer (0: none yet); copies min(n, capacity) doubles when h_out is non-null.
// set: replaces the vector (n = 0 drops it); the next update_power with the same system size starts from it.
extern "C" int dkmc_get_current_warm_vector(const dkmc_gpubuf *buf, double *h_out, int capacity, int *n_out)
{
    Engine &e = eng();
    XBufState *st = buf ? xstate_find(buf->site_x) : nullptr;
    const int n = (st && st->warm) ? st->warm_n : 0;
    if (n_out) *n_out = n;
    if (h_out && n > 0) {
        HIPCHK(hipStreamSynchronize(e.stream));
        HIPCHK(hipMemcpy(h_out, st->warm, (size_t)(n < capacity ? n : capacity) * 8, hipMemcpyDeviceToHost));
    }
    return 0;
}
extern "C" int dkmc_set_current_warm_vector(const dkmc_gpubuf *buf, const double *h_in, int n)
{
    Engine &e = eng();
    if (!buf || n < 0 || (n > 0 && !h_in)) return dkmc_fail(14, "set_current_warm_vector: bad arguments", __FILE__, __LINE__);
    xstate_select(buf->site_x);
    HIPCHK(hipStreamSynchronize(e.stream));
    if (g_warm_n != n) { if (g_warm) (void)hipFree(g_warm); g_warm = nullptr; g_warm_n = 0; if (n > 0) { HIPCHK(hipMalloc((void **)&g_warm, (size_t)n * 8)); g_warm_n = n; } }
    if (n > 0) HIPCHK(hipMemcpy(g_warm, h_in, (size_t)n * 8, hipMemcpyHostToDevice));
    return 0;
}

// ... and the solutions of the block-CG's auxiliary columns the next solve starts from (dkmc_set_x_aux_warm): [rows][16] doubles, n = rows x 16
// (0: none held / not valid)
extern "C" int dkmc_get_current_warm_aux(const dkmc_gpubuf *buf, double *h_out, long long capacity, long long *n_out)
{
    Engine &e = eng();
    XBufState *st = buf ? xstate_find(buf->site_x) : nullptr;
    const long long n = (st && st->warm_aux && st->warm_aux_valid) ? (long long)st->warm_aux_n * 16 : 0;
    if (n_out) *n_out = n;
    if (h_out && n > 0) {
        HIPCHK(hipStreamSynchronize(e.stream));
        HIPCHK(hipMemcpy(h_out, st->warm_aux, (size_t)(n < capacity ? n : capacity) * 8, hipMemcpyDeviceToHost));
    }
    return 0;
}
extern "C" int dkmc_set_current_warm_aux(const dkmc_gpubuf *buf, const double *h_in, long long n)
{
    Engine &e = eng();
    if (!buf || n < 0 || (n % 16) != 0 || (n > 0 && !h_in)) return dkmc_fail(14, "set_current_warm_aux: bad arguments", __FILE__, __LINE__);
    xstate_select(buf->site_x);
    HIPCHK(hipStreamSynchronize(e.stream));
    const int rows = (int)(n / 16);
    if (g_cur->warm_aux_n != rows) {
        if (g_cur->warm_aux) (void)hipFree(g_cur->warm_aux);
        g_cur->warm_aux = nullptr; g_cur->warm_aux_n = 0;
        if (rows > 0) { HIPCHK(hipMalloc((void **)&g_cur->warm_aux, (size_t)n * 8)); g_cur->warm_aux_n = rows; }
    }
    g_cur->warm_aux_valid = 0;
    if (rows > 0) { HIPCHK(hipMemcpy(g_cur->warm_aux, h_in, (size_t)n * 8, hipMemcpyHostToDevice)); g_cur->warm_aux_valid = 1; }
    return 0;
}

extern "C" int dkmc_get_last_X(int *rows_out, long long *nnz_out, int *h_rp, int *h_col, double *h_data)
{
    Engine &e = eng();
    if (g_last_tiled) return xt_export_csr(rows_out, nnz_out, h_rp, h_col, h_data);
    if (rows_out) *rows_out = g_last_rows;
    if (nnz_out) *nnz_out = g_last_nnz;
    HIPCHK(hipStreamSynchronize(e.stream));
    if (h_rp) {
        if (g_last_nnz > 2147483647LL) return dkmc_fail(12, "get_last_X: more than 2^31-1 non-zeros do not fit the int32 row pointers of this call", __FILE__, __LINE__);
        std::vector<xrp_t> tmp((size_t)g_last_rows + 1);
        HIPCHK(hipMemcpy(tmp.data(), e.buf[S_X_ROWPTR], tmp.size() * sizeof(xrp_t), hipMemcpyDeviceToHost));
        for (int i = 0; i <= g_last_rows; ++i) h_rp[i] = (int)tmp[i];
    }
    if (h_col) HIPCHK(hipMemcpy(h_col, e.buf[S_X_COL], (size_t)g_last_nnz * 4, hipMemcpyDeviceToHost));
    if (h_data) HIPCHK(hipMemcpy(h_data, e.buf[S_X_DATA], (size_t)g_last_nnz * 8, hipMemcpyDeviceToHost));
    return 0;
}
