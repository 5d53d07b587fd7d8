// cg.hip -- CSR SpMV and the Jacobi-scaled conjugate-gradient solve.
//
// Replaces solve_sparse_CG_Jacobi (iterative_solvers_gpu.cu:309-480) and the cuSPARSE SpMV /
// cuBLAS dot-axpy-scal calls it makes (:411-448).  Same algorithm and sign convention
// (r = A y - x, p = -r, alpha = r.r / p.Ap, y += alpha p, r += alpha Ap, beta = r'.r'/r.r,
// p = beta p - r; first test on ||r||, later ones on ||r||^2, both against tol^2), but:
//   * alpha, beta and the stop test live on the device; the host only polls a flag every batch
//     of iterations instead of synchronising on every dot product (3 per iteration in the reference);
//   * dot products are fused into the SpMV / axpy kernels; block partials are written to memory and
//     re-reduced in a fixed order by every block of the consuming kernel (no atomics, run-to-run
//     deterministic);
//   * rows are binned by length: sub-wave groups of 16 lanes for short rows (K: ~26 nnz/row), one
//     wave64 per long row (tunnelling rows of X have thousands of entries), so that each row is
//     read as contiguous 64/128-byte segments.
// HBM traffic per iteration (algorithmic): 12*nnz + 4*(m+1) + 96*m bytes.
#include "common.h"
#include <vector>

#define CG_NT 256
#define CG_MAX_PART 1024        // max blocks writing partials per kernel family
#define LONG_ROW_NNZ 192        // rows with more entries go to the wave-per-row bin

struct CgCtrl {                 // device-resident control block
    double rr[2];               // ||r||^2, double-buffered by iteration parity
    double pad;
    int done;
    int iters;
};

enum { M_SCALE = 0, M_INIT = 1, M_AP = 2, M_DIAG = 3 };

// block-uniform read of the stop flag (only the last kernel of an iteration ever sets it)
__device__ __forceinline__ bool cg_done(const CgCtrl *ctrl)
{
    __shared__ int sdone;
    if (threadIdx.x == 0) sdone = ctrl->done;
    __syncthreads();
    return sdone != 0;
}

// fixed-order reduction of `n` partials, identical in every block
__device__ __forceinline__ double reduce_partials(const double *part, int n, double *red)
{
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += CG_NT) s += part[i];
    return block_sum_all<CG_NT>(s, red);
}

// One group of LPR lanes per row.  rows == nullptr: identity row list.
template <int LPR, int MODE>
__global__ __launch_bounds__(CG_NT) void k_spmv(int nrows, const int *__restrict__ rows, const int *__restrict__ rp,
                                                const int *__restrict__ ci, double *__restrict__ a,
                                                const double *__restrict__ vin, double *__restrict__ vout,
                                                double *__restrict__ aux0, double *__restrict__ aux1,
                                                double *__restrict__ part, const CgCtrl *ctrl)
{
    __shared__ double red[CG_NT / 64];
    if (MODE == M_AP) { if (cg_done(ctrl)) return; }
    const int gpb = CG_NT / LPR;                         // groups per block
    const int g = threadIdx.x / LPR, l = threadIdx.x % LPR;
    double acc = 0.0;
    for (int ridx = blockIdx.x * gpb + g; ridx < nrows; ridx += gridDim.x * gpb) {
        const int row = rows ? rows[ridx] : ridx;
        const int p0 = rp[row], p1 = rp[row + 1];
        if (MODE == M_SCALE) {                           // a <- S a S   (jacobi_precondition_matrix :272-291)
            const double si = vin[row];
            for (int p = p0 + l; p < p1; p += LPR) a[p] = a[p] * si * vin[ci[p]];
        } else if (MODE == M_DIAG) {                     // s = 1/sqrt(diag); x *= s; y /= s (:227-306)
            double d = 0.0;
            for (int p = p0 + l; p < p1; p += LPR) if (ci[p] == row) d = a[p];
#pragma unroll
            for (int off = LPR / 2; off > 0; off >>= 1) d += __shfl_xor(d, off, LPR);
            if (l == 0) {
                const double s = 1.0 / sqrt(d);
                vout[row] = s;
                aux0[row] = aux0[row] * s;               // rhs
                aux1[row] = aux1[row] * 1 / s;           // guess
            }
        } else {
            double s = 0.0;
            for (int p = p0 + l; p < p1; p += LPR) s += a[p] * vin[ci[p]];
#pragma unroll
            for (int off = LPR / 2; off > 0; off >>= 1) s += __shfl_xor(s, off, LPR);
            if (l == 0) {
                if (MODE == M_INIT) {                    // r = A y - x ; p = -r
                    const double r = -aux0[row] + s;
                    vout[row] = r; aux1[row] = -r;
                    acc += r * r;
                } else {                                 // t = A p ; partial p.t
                    vout[row] = s;
                    acc += vin[row] * s;
                }
            }
        }
    }
    if (MODE == M_INIT || MODE == M_AP) {
        const double tot = block_sum_all<CG_NT>(acc, red);
        if (threadIdx.x == 0) part[blockIdx.x] = tot;
    }
}

// after M_INIT: rr0 and the first stop test on the 2-norm (cublasDnrm2, :418)
__global__ __launch_bounds__(CG_NT) void k_cg_check0(const double *part, int npart, CgCtrl *ctrl, double tol2)
{
    __shared__ double red[CG_NT / 64];
    const double rr = reduce_partials(part, npart, red);
    if (threadIdx.x == 0) { ctrl->rr[0] = rr; ctrl->rr[1] = rr; ctrl->iters = 0; ctrl->done = !(sqrt(rr) > tol2); }
}

// alpha = rr / pAp ; y += alpha p ; r += alpha t ; partial r.r
__global__ __launch_bounds__(CG_NT) void k_cg_update(int m, int it, const double *__restrict__ part_pAp, int npart,
                                                     const double *__restrict__ p, const double *__restrict__ t,
                                                     double *__restrict__ y, double *__restrict__ r,
                                                     double *__restrict__ part_rr, const CgCtrl *ctrl)
{
    __shared__ double red[CG_NT / 64];
    if (cg_done(ctrl)) return;
    const double pAp = reduce_partials(part_pAp, npart, red);
    const double alpha = ctrl->rr[it & 1] / pAp;
    double acc = 0.0;
    for (int i = blockIdx.x * CG_NT + threadIdx.x; i < m; i += gridDim.x * CG_NT) {
        y[i] += alpha * p[i];
        const double rn = r[i] + alpha * t[i];
        r[i] = rn;
        acc += rn * rn;
    }
    const double tot = block_sum_all<CG_NT>(acc, red);
    if (threadIdx.x == 0) part_rr[blockIdx.x] = tot;
}

// beta = rr' / rr ; p = beta p - r ; stop test on rr' (cublasDdot :448)
__global__ __launch_bounds__(CG_NT) void k_cg_direction(int m, int it, const double *__restrict__ part_rr, int npart,
                                                        const double *__restrict__ r, double *__restrict__ p,
                                                        CgCtrl *ctrl, double tol2)
{
    __shared__ double red[CG_NT / 64];
    if (cg_done(ctrl)) return;
    const double rr_new = reduce_partials(part_rr, npart, red);
    const double beta = rr_new / ctrl->rr[it & 1];
    for (int i = blockIdx.x * CG_NT + threadIdx.x; i < m; i += gridDim.x * CG_NT) p[i] = p[i] * beta - r[i];
    // Block 0 publishes the scalars of the next iteration.  Setting `done` while other blocks of this
    // launch may still be starting is harmless: a block that sees it skips a p update nobody reads.
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        ctrl->rr[(it + 1) & 1] = rr_new;
        ctrl->iters = it + 1;
        if (!(rr_new > tol2)) ctrl->done = 1;
    }
}

__global__ __launch_bounds__(CG_NT) void k_vec_mul(int m, double *__restrict__ y, const double *__restrict__ s)
{
    for (int i = blockIdx.x * CG_NT + threadIdx.x; i < m; i += gridDim.x * CG_NT) y[i] = y[i] * s[i];
}

// row binning: flag long rows, build the two row lists
__global__ void k_row_flags(int m, const int *rp, int *is_long, int *is_short)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) { int L = (rp[i + 1] - rp[i]) > LONG_ROW_NNZ; is_long[i] = L; is_short[i] = !L; }
}
__global__ void k_row_lists(int m, const int *is_long, const int *off_long, const int *off_short, int *long_rows, int *short_rows)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) { if (is_long[i]) long_rows[off_long[i]] = i; else short_rows[off_short[i]] = i; }
}

static inline int grid_for(int work_items, int per_block)
{
    long long b = ((long long)work_items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > CG_MAX_PART / 2) b = CG_MAX_PART / 2;
    return (int)b;
}

// Internal entry: uniform_rows != 0 skips the binning (K: every row is short).
int cg_solve_jacobi(double *a, const int *rp, const int *ci, int nnz, int m, double *x, double *y,
                    int uniform_rows, int *iters_out, double *rr_out)
{
    Engine &e = eng(); hipStream_t st = e.stream;
    if (m <= 0) { if (iters_out) *iters_out = 0; if (rr_out) *rr_out = 0; return 0; }
    double *s = (double *)scratch(S_CG_S, (size_t)m * 8), *r = (double *)scratch(S_CG_R, (size_t)m * 8);
    double *p = (double *)scratch(S_CG_P, (size_t)m * 8), *t = (double *)scratch(S_CG_T, (size_t)m * 8);
    double *part = (double *)scratch(S_CG_PART, (size_t)3 * CG_MAX_PART * 8);
    CgCtrl *ctrl = (CgCtrl *)scratch(S_CG_CTRL, sizeof(CgCtrl));
    if (!s || !r || !p || !t || !part || !ctrl) return e.err_code;
    double *part_pAp = part, *part_rr = part + CG_MAX_PART;
    const double tol2 = e.cg_tol * e.cg_tol;

    // ---- row bins ----
    int n_long = 0, n_short = m; const int *long_rows = nullptr, *short_rows = nullptr;
    if (!uniform_rows) {
        int *fl = (int *)scratch(S_MISC0, (size_t)m * 4), *fs = (int *)scratch(S_MISC1, (size_t)m * 4);
        int *ol = (int *)scratch(S_MISC2, (size_t)(m + 2) * 4), *os = (int *)scratch(S_MISC3, (size_t)(m + 2) * 4);
        int *lists = (int *)scratch(S_SCAN_TMP2, (size_t)(m + 4) * 4);
        if (!fl || !fs || !ol || !os || !lists) return e.err_code;
        hipLaunchKernelGGL(k_row_flags, dim3((m + 255) / 256), dim3(256), 0, st, m, rp, fl, fs);
        int rc = dkmc_exclusive_scan_i32(fl, ol, m, ol + m); if (rc) return rc;
        rc = dkmc_exclusive_scan_i32(fs, os, m, os + m); if (rc) return rc;
        int tot[1];
        HIPCHK(hipMemcpyAsync(tot, ol + m, sizeof(int), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        n_long = tot[0]; n_short = m - n_long;
        int *lr = lists, *sr = lists + n_long;
        hipLaunchKernelGGL(k_row_lists, dim3((m + 255) / 256), dim3(256), 0, st, m, fl, ol, os, lr, sr);
        long_rows = lr; short_rows = sr;
    }
    const int gs = n_short > 0 ? grid_for(n_short, CG_NT / 16) : 0;    // short-row blocks
    const int gl = n_long > 0 ? grid_for(n_long, CG_NT / 64) : 0;      // long-row blocks
    const int gv = grid_for(m, CG_NT);                                 // vector-kernel blocks
    const int np_spmv = gs + gl;

#define SPMV(MODE, vin, vout, a0, a1, partp)                                                                          \
    do {                                                                                                             \
        if (gs) hipLaunchKernelGGL((k_spmv<16, MODE>), dim3(gs), dim3(CG_NT), 0, st, n_short, short_rows, rp, ci, a,  \
                                   vin, vout, a0, a1, partp, ctrl);                                                  \
        if (gl) hipLaunchKernelGGL((k_spmv<64, MODE>), dim3(gl), dim3(CG_NT), 0, st, n_long, long_rows, rp, ci, a,    \
                                   vin, vout, a0, a1, (partp) ? (partp) + gs : nullptr, ctrl);                       \
    } while (0)

    // ---- Jacobi scaling ----
    SPMV(M_DIAG, (const double *)nullptr, s, x, y, (double *)nullptr);
    SPMV(M_SCALE, (const double *)s, (double *)nullptr, (double *)nullptr, (double *)nullptr, (double *)nullptr);
    // ---- r = A y - x, p = -r ----
    SPMV(M_INIT, (const double *)y, r, x, p, part_rr);
    hipLaunchKernelGGL(k_cg_check0, dim3(1), dim3(CG_NT), 0, st, part_rr, np_spmv, ctrl, tol2);
    KCHK();

    // ---- optional kernel profile: HIP events around every A*p launch (bench.py roofline) ----
    const bool prof = e.profiling && !uniform_rows;
    static hipEvent_t evs[3 * 64]; static bool evs_ready = false;
    double prof_short_ms = 0.0, prof_long_ms = 0.0; int prof_short_n = 0, prof_long_n = 0;
    if (prof) {
        if (!evs_ready) { for (auto &ev : evs) HIPCHK(hipEventCreate(&ev)); evs_ready = true; }
        std::vector<int> hrp((size_t)m + 1);
        HIPCHK(hipMemcpy(hrp.data(), rp, ((size_t)m + 1) * 4, hipMemcpyDeviceToHost));
        long long nl = 0, nsh = 0;
        for (int i = 0; i < m; ++i) { const int c = hrp[i + 1] - hrp[i]; if (c > LONG_ROW_NNZ) nl += c; else nsh += c; }
        e.stats.spmv_long_nnz = nl; e.stats.spmv_short_nnz = nsh; e.stats.spmv_long_rows = n_long; e.stats.spmv_short_rows = n_short;
    }
    // ---- iterations, launched in batches; the host polls the control block between batches ----
    int it = 0, batch = 8, launched = 0;
    CgCtrl h{};
    for (;;) {
        HIPCHK(hipMemcpyAsync(&h, ctrl, sizeof(CgCtrl), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (prof && launched) {       // only launches that did work (iteration index below the final count) are counted
            for (int b = 0; b < launched; ++b) {
                if (it - launched + b >= h.iters) break;
                float ms = 0.f;
                if (gs) { HIPCHK(hipEventElapsedTime(&ms, evs[3 * b], evs[3 * b + 1])); prof_short_ms += ms; ++prof_short_n; }
                if (gl) { HIPCHK(hipEventElapsedTime(&ms, evs[3 * b + 1], evs[3 * b + 2])); prof_long_ms += ms; ++prof_long_n; }
            }
        }
        if (h.done) break;
        if (it >= 200000) { dkmc_fail(4, "CG: no convergence after 200000 iterations", __FILE__, __LINE__); break; }
        for (int b = 0; b < batch; ++b, ++it) {
            if (prof) {
                HIPCHK(hipEventRecord(evs[3 * b], st));
                if (gs) hipLaunchKernelGGL((k_spmv<16, M_AP>), dim3(gs), dim3(CG_NT), 0, st, n_short, short_rows, rp, ci, a, (const double *)p, t,
                                           (double *)nullptr, (double *)nullptr, part_pAp, ctrl);
                HIPCHK(hipEventRecord(evs[3 * b + 1], st));
                if (gl) hipLaunchKernelGGL((k_spmv<64, M_AP>), dim3(gl), dim3(CG_NT), 0, st, n_long, long_rows, rp, ci, a, (const double *)p, t,
                                           (double *)nullptr, (double *)nullptr, part_pAp + gs, ctrl);
                HIPCHK(hipEventRecord(evs[3 * b + 2], st));
            } else {
                SPMV(M_AP, (const double *)p, t, (double *)nullptr, (double *)nullptr, part_pAp);
            }
            hipLaunchKernelGGL(k_cg_update, dim3(gv), dim3(CG_NT), 0, st, m, it, part_pAp, np_spmv, p, t, y, r, part_rr, ctrl);
            hipLaunchKernelGGL(k_cg_direction, dim3(gv), dim3(CG_NT), 0, st, m, it, part_rr, gv, r, p, ctrl, tol2);
        }
        launched = batch;
        KCHK();
        if (batch < 64) batch *= 2;
    }
#undef SPMV
    if (prof) {
        e.stats.spmv_long_ms = prof_long_ms; e.stats.spmv_short_ms = prof_short_ms;
        e.stats.spmv_long_launches = prof_long_n; e.stats.spmv_short_launches = prof_short_n;
    }
    // ---- un-scale the solution (:459) ----
    hipLaunchKernelGGL(k_vec_mul, dim3(gv), dim3(CG_NT), 0, st, m, y, s);
    KCHK();
    if (iters_out) *iters_out = h.iters;
    if (rr_out) *rr_out = h.rr[h.iters & 1];
    return e.err_code;
}

extern "C" int dkmc_solve_sparse_CG_Jacobi(double *A, const int *rp, const int *ci, int nnz, int m, double *x, double *y,
                                           int *iters_out, double *rr_out)
{
    return cg_solve_jacobi(A, rp, ci, nnz, m, x, y, 0, iters_out, rr_out);
}
