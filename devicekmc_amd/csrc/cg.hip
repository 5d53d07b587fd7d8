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
//   * rows are binned by length.  Short rows (K: ~26 nnz/row; the neighbour rows of X) take 16 lanes each.  The long rows
//     of X (tunnelling rows, thousands of entries) are not multiplied in CSR form at all inside the iteration loop: their
//     entries are viewed as index-free dense runs cut into <= 2048-entry segments, one wave64 per segment, 8 B per entry
//     (k_build_runs / k_spmv_segs below); a tiny second kernel adds a row's segment partials in a fixed order;
//   * (the default solve of X does not come through here at all: xt.hip generates the tunnelling block straight into symmetric
//     tiles; this file solves K, the local-heat systems and X in its CSR form, dkmc_set_x_format(0));
//   * row pointers are a template parameter (int for K, 64-bit for X whose non-zeros outgrow 2^31 at ~4e5 sites);
//   * with a communicator attached (comm.hip) the matrix stream is dealt to the ranks and one collective per iteration
//     completes the long rows' sums.
// HBM traffic per iteration in the CSR formulation: 12*nnz + 4*(m+1) + 96*m bytes; with the segments the
// matrix part of X drops to 8 B per entry.
#include "common.h"
#include <hip/hip_ext.h>
#include <vector>
#include <algorithm>
#include <stdlib.h>

#define CG_NT 256
#define CG_MAX_PART 8192        // max blocks writing partials per kernel family
#define PROF_STRIDE 8
#define LONG_ROW_NNZ 192        // rows with more entries go to the wave-per-row bin

struct CgCtrl {                 // device-resident control block
    double rr[2];               // ||r||^2, double-buffered by iteration parity
    double pad;
    int done;
    int iters;
};

enum { M_SCALE = 0, M_INIT = 1, M_AP = 2, M_DIAG = 3 };

// dense-run view of long rows (see k_build_runs)
#define RUN_MIN_LEN 32
#define REM_SEG_LEN 256
#define SEG_LEN 2048           // entries per segment = work item of one wave in k_spmv_segs
struct __attribute__((aligned(16))) RunDesc { long long pos; int len, sr0; };   // sr0 < 0: gather segment of long row -1-sr0
#define SEGK_NT 256            // workgroup of k_spmv_segs: 4 waves, one segment each

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

// same, fetching the stop flag in the same barrier round (saves one dependent global load + barrier per kernel)
__device__ __forceinline__ double reduce_partials_and_flag(const double *part, int n, double *red, const CgCtrl *ctrl, bool *done)
{
    __shared__ int sdone;
    if (threadIdx.x == 0) sdone = ctrl->done;
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += CG_NT) s += part[i];
    s = block_sum_all<CG_NT>(s, red);          // its barriers also publish sdone
    *done = sdone != 0;
    return s;
}

// One group of LPR lanes per row.  rows == nullptr: identity row list.
template <int LPR, int MODE, typename RP>
__global__ __launch_bounds__(CG_NT) void k_spmv(int nrows, const int *__restrict__ rows, const RP *__restrict__ rp,
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
        const RP p0 = rp[row], p1 = rp[row + 1];
        if (MODE == M_SCALE) {                           // a <- S a S   (jacobi_precondition_matrix :272-291)
            const double si = vin[row];
            for (RP p = p0 + l; p < p1; p += LPR) a[p] = a[p] * si * vin[ci[p]];
        } else if (MODE == M_DIAG) {                     // s = 1/sqrt(diag); x *= s; y /= s (:227-306)
            double d = 0.0;
            for (RP p = p0 + l; p < p1; p += LPR) if (ci[p] == row) d = a[p];
            d = group_sum<LPR>(d);
            if (l == 0) {
                const double s = 1.0 / sqrt(d);
                vout[row] = s;
                aux0[row] = aux0[row] * s;               // rhs
                aux1[row] = aux1[row] * 1 / s;           // guess
            }
        } else {
            double s = 0.0;
            for (RP p = p0 + l; p < p1; p += LPR) s += a[p] * vin[ci[p]];
            s = group_sum<LPR>(s);
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

// ---- hot kernel of the CG loop: t = A p with the partial p.t, short and long rows in ONE launch -------------------
// Blocks [0, gs) take the short rows (16 lanes per row, 64 rows per pass), blocks [gs, gs+gl) the long rows
// (one wave64 per row, 4 independent 64-entry strips in flight per lane).  1024-thread workgroups: 16 waves per
// workgroup keep <= 1024 partials while every CU holds 32 waves; the matrix stream (12 B per non-zero, read once per
// launch) uses non-temporal loads so that it does not evict the gathered vector from L2.
#define SPMV_NT 1024
template <int VAR, int RUNS, typename RP>
__global__ __launch_bounds__(SPMV_NT) void k_spmv_ap(int n_short, const int *__restrict__ short_rows, int gs,
                                                     int n_long, const int *__restrict__ long_rows,
                                                     const RP *__restrict__ rp, const int *__restrict__ ci,
                                                     const double *__restrict__ a, const double *__restrict__ p,
                                                     double *__restrict__ t, double *__restrict__ part, const CgCtrl *ctrl,
                                                     const RunDesc *__restrict__ runs, const int *__restrict__ nruns,
                                                     const int *__restrict__ rem, const int *__restrict__ nrem,
                                                     const double *__restrict__ pS, const int *__restrict__ seg_off)
{
    __shared__ double red[SPMV_NT / 64];
    __shared__ int sdone;
    if (threadIdx.x == 0) sdone = ctrl->done;
    __syncthreads();
    if (sdone) return;
    double acc = 0.0;
    if ((int)blockIdx.x < gs) {
        const int g = threadIdx.x >> 4, l = threadIdx.x & 15;
        for (int ridx = blockIdx.x * (SPMV_NT / 16) + g; ridx < n_short; ridx += gs * (SPMV_NT / 16)) {
            const int row = short_rows ? short_rows[ridx] : ridx;
            const RP p0 = rp[row], p1 = rp[row + 1];
            double s = 0.0;
            for (RP q = p0 + l; q < p1; q += 16) s += a[q] * p[ci[q]];
            s = group_sum<16>(s);
            if (l == 0) { t[row] = s; acc += p[row] * s; }
        }
    } else if (RUNS) {
        // stage 2 of the long-row product: 16 lanes per row add the row's segment partials (fixed order) and its few
        // entries outside long runs (columns outside S, boundary columns)
        const int g = threadIdx.x >> 4, l = threadIdx.x & 15;
        const int gl = gridDim.x - gs;
        for (int ridx = (blockIdx.x - gs) * (SPMV_NT / 16) + g; ridx < n_long; ridx += gl * (SPMV_NT / 16)) {
            const int row = long_rows[ridx];
            double s = 0.0;
            const int nsg = nruns[ridx];
            const double *sp = pS + seg_off[ridx];            // pS slot carries the segment partials in this mode
            for (int j = l; j < nsg; j += 16) s += sp[j];
            s = group_sum<16>(s);
            if (l == 0) { t[row] = s; acc += p[row] * s; }
        }
    } else {
        const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
        const int gl = gridDim.x - gs;
        // rows are dealt to waves with stride gl, so that every workgroup gets a mix of the heavy rows (left-contact
        // and vacancy rows, ~3x the entries) and the light ones instead of 16 neighbours of the same class
        for (int ridx = w * gl + (blockIdx.x - gs); ridx < n_long; ridx += gl * (SPMV_NT / 64)) {
            const int row = long_rows[ridx];
            const RP p0 = rp[row], p1 = rp[row + 1];
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
            RP q = p0 + lane;
            for (; q + 192 < p1; q += 256) {
                const int c0 = __builtin_nontemporal_load(ci + q), c1 = __builtin_nontemporal_load(ci + q + 64);
                const int c2 = __builtin_nontemporal_load(ci + q + 128), c3 = __builtin_nontemporal_load(ci + q + 192);
                const double a0 = __builtin_nontemporal_load(a + q), a1 = __builtin_nontemporal_load(a + q + 64);
                const double a2 = __builtin_nontemporal_load(a + q + 128), a3 = __builtin_nontemporal_load(a + q + 192);
                s0 += a0 * p[c0]; s1 += a1 * p[c1]; s2 += a2 * p[c2]; s3 += a3 * p[c3];
            }
            for (; q < p1; q += 64) s0 += __builtin_nontemporal_load(a + q) * p[__builtin_nontemporal_load(ci + q)];
            double s = (s0 + s1) + (s2 + s3);
            s = wave_sum(s);
            if (lane == 0) { t[row] = s; acc += p[row] * s; }
        }
    }
    const double tot = block_sum_all<SPMV_NT>(acc, red);
    if (threadIdx.x == 0) part[blockIdx.x] = tot;
}

// ---- sharded solve (comm.hip): stage 2 of the long-row product split around the exchange step ---------------------------
// The long rows are dealt to the ranks at row boundaries (balanced by segment count).  k_rowsum_owner adds the segment
// partials of the rows this rank owns -- the same 16-lane arithmetic as the fused stage 2 above -- into its chunk of the
// exchange buffer; after the all-gather k_rowsum_apply, with the fused kernel's launch shape and thread-to-row mapping,
// stores t and forms the p.t partials.  Values and summation order are those of the single-GPU kernel: bit-identical.
#define MAX_RANKS 64
struct RowParts { int n; int chunk; int lo[MAX_RANKS + 1]; };      // rank r owns long rows [lo[r], lo[r+1]); chunk = slots per rank

__global__ __launch_bounds__(SPMV_NT) void k_rowsum_owner(int row_lo, int row_hi, const int *__restrict__ nruns, const int *__restrict__ seg_off,
                                                          const double *__restrict__ seg_part, double *__restrict__ mine, const CgCtrl *ctrl)
{
    if (ctrl->done) return;
    const int g = threadIdx.x >> 4, l = threadIdx.x & 15;
    for (int ridx = row_lo + blockIdx.x * (SPMV_NT / 16) + g; ridx < row_hi; ridx += gridDim.x * (SPMV_NT / 16)) {
        double s = 0.0;
        const int nsg = nruns[ridx];
        const double *sp = seg_part + seg_off[ridx];
        for (int j = l; j < nsg; j += 16) s += sp[j];
        s = group_sum<16>(s);
        if (l == 0) mine[ridx - row_lo] = s;
    }
}

__global__ __launch_bounds__(SPMV_NT) void k_rowsum_apply(int n_long, const int *__restrict__ long_rows, const RowParts *__restrict__ rp_, const double *__restrict__ gathered,
                                                          const double *__restrict__ p, double *__restrict__ t, double *__restrict__ part,
                                                          const CgCtrl *ctrl)
{
    __shared__ double red[SPMV_NT / 64];
    __shared__ int sdone;
    if (threadIdx.x == 0) sdone = ctrl->done;
    __syncthreads();
    if (sdone) return;
    double acc = 0.0;
    const int g = threadIdx.x >> 4, l = threadIdx.x & 15;
    for (int ridx = blockIdx.x * (SPMV_NT / 16) + g; ridx < n_long; ridx += gridDim.x * (SPMV_NT / 16)) {
        if (l == 0) {
            int r = 0;
            const int nr = rp_->n;
            while (r + 1 < nr && ridx >= rp_->lo[r + 1]) ++r;
            const double s = gathered[(size_t)r * rp_->chunk + (ridx - rp_->lo[r])];
            const int row = long_rows[ridx];
            t[row] = s; acc += p[row] * s;
        }
    }
    const double tot = block_sum_all<SPMV_NT>(acc, red);
    if (threadIdx.x == 0) part[blockIdx.x] = tot;
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
    bool done;
    const double pAp = reduce_partials_and_flag(part_pAp, npart, red, ctrl, &done);
    if (done) return;
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
                                                        CgCtrl *ctrl, double tol2, const int *__restrict__ srank, double *__restrict__ pS)
{
    __shared__ double red[CG_NT / 64];
    bool done;
    const double rr_new = reduce_partials_and_flag(part_rr, npart, red, ctrl, &done);
    if (done) return;
    const double beta = rr_new / ctrl->rr[it & 1];
    for (int i = blockIdx.x * CG_NT + threadIdx.x; i < m; i += gridDim.x * CG_NT) {
        const double pn = p[i] * beta - r[i];
        p[i] = pn;
        if (srank) { const int k = srank[i]; if (k >= 0) pS[k] = pn; }      // compact copy over the tunnelling set
    }
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

// ---- index-free "dense run" view of the long rows (tunnelling block of X) ------------------------------------------------
// In a long row of X nearly every entry belongs to a long sequence of consecutive columns *in the numbering of the
// tunnelling set S* (99 % of the entries at 85 k sites sit in runs of >= 32): the block is dense up to class structure.
// A run (pos, len, sr0) says: a[pos .. pos+len) multiplies pS[sr0 .. sr0+len), where pS is the direction vector compacted
// over S.  The hot kernel then streams 8 B per entry (values only), reads pS with contiguous loads and needs no column
// indices; the few entries outside long runs (neighbours outside S, boundary columns) go through a per-row remainder list.

template <typename RP>
__global__ __launch_bounds__(256) void k_build_runs(int n_long, const int *__restrict__ long_rows, const RP *__restrict__ rp,
                                                    const int *__restrict__ ci, const int *__restrict__ srank,
                                                    RunDesc *__restrict__ runs, int *__restrict__ nruns,
                                                    int *__restrict__ rem, int *__restrict__ nrem, int seg_len)
{
    const int lane = threadIdx.x & 63;
    const int ridx = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ridx >= n_long) return;
    const int row = long_rows[ridx];
    const RP p0 = rp[row], p1 = rp[row + 1];
    const long long run_base = (long long)(p0 / RUN_MIN_LEN) + ridx;
    int nr = 0, nrm = 0;
    RP cur_start = p0; int cur_sr0 = -1, carry = -2;
    auto finalize = [&](RP sA, RP eA, int sr0) {
        const int len = (int)(eA - sA);
        if (len <= 0) return;
        if (len >= RUN_MIN_LEN && sr0 >= 0) {       // long run: emitted as segments of at most SEG_LEN entries
            for (long long c = 0; c < len; c += seg_len) {
                if (lane == 0) { RunDesc d; d.pos = (long long)sA + c; d.len = (int)min((long long)seg_len, (long long)len - c); d.sr0 = sr0 + (int)c; runs[run_base + nr] = d; }
                ++nr;
            }
        } else {
            for (int k = lane; k < len; k += WAVE) rem[p0 + nrm + k] = (int)(sA - p0) + k;      // offsets relative to the row start
            nrm += len;
        }
    };
    for (RP c0 = p0; c0 < p1; c0 += WAVE) {
        const RP q = c0 + lane;
        const bool valid = q < p1;
        const int sr = valid ? srank[ci[q]] : -1;
        int prev = __shfl_up(sr, 1, WAVE);
        if (lane == 0) prev = carry;
        const bool brk = valid && (q == p0 || sr < 0 || prev < 0 || sr != prev + 1);
        unsigned long long mask = __ballot(brk);
        while (mask) {
            const int b = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            const RP s_new = c0 + b;
            finalize(cur_start, s_new, cur_sr0);
            cur_start = s_new;
            cur_sr0 = __shfl(sr, b, WAVE);
        }
        carry = __shfl(sr, 63, WAVE);
    }
    finalize(cur_start, p1, cur_sr0);
    // the entries outside long runs become "gather segments" (sr0 = -1): chunks of the row's remainder list
    for (int c = 0; c < nrm; c += REM_SEG_LEN) {
        if (lane == 0) { RunDesc d; d.pos = (long long)p0 + c; d.len = min(REM_SEG_LEN, nrm - c); d.sr0 = -1 - ridx; runs[run_base + nr] = d; }
        ++nr;
    }
    if (lane == 0) { nruns[ridx] = nr; nrem[ridx] = nrm; }
}

__global__ __launch_bounds__(256) void k_compact_pS(int m, const int *__restrict__ srank, const double *__restrict__ p, double *__restrict__ pS)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < m) { const int r = srank[i]; if (r >= 0) pS[r] = p[i]; }
}

// gather the per-row descriptor lists (stored at capacity offsets) into one dense segment array
template <typename RP>
__global__ __launch_bounds__(256) void k_compact_segs(int n_long, const int *__restrict__ long_rows, const RP *__restrict__ rp,
                                                      const RunDesc *__restrict__ runs, const int *__restrict__ nruns,
                                                      const int *__restrict__ seg_off, RunDesc *__restrict__ segs)
{
    const int lane = threadIdx.x & 63;
    const int ridx = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ridx >= n_long) return;
    const RunDesc *src = runs + ((long long)(rp[long_rows[ridx]] / RUN_MIN_LEN) + ridx);
    RunDesc *dst = segs + seg_off[ridx];
    for (int j = lane; j < nruns[ridx]; j += WAVE) dst[j] = src[j];
}

// stage 1 of the long-row product: one wave64 per segment, seg_part[seg] = sum a[pos+k] * pS[sr0+k].
// Streams 8 B per entry, 4 independent 1-KiB strips in flight per wave; every wave has the same amount of work.
// Default cache policy on the matrix stream (NTL = 0): the same 240 MB are re-read every CG iteration and partly stay in
// the 256 MiB Infinity Cache -- measured 45 us per launch against 53 us with non-temporal loads (NTL = 1).
// Blocks [0, nsb): segments; the rest: the short rows, 16 lanes per row, p.t partial per block.
template <int NTL, typename RP>
__global__ __launch_bounds__(SEGK_NT) void k_spmv_segs(int nseg, const RunDesc *__restrict__ segs, const double *__restrict__ a,
                                                       const double *__restrict__ pS, double *__restrict__ seg_part, const CgCtrl *ctrl,
                                                       const int *__restrict__ rem, const int *__restrict__ ci, const double *__restrict__ p,
                                                       int nsb, int n_short, const int *__restrict__ short_rows, const RP *__restrict__ rp,
                                                       const int *__restrict__ long_rows, double *__restrict__ t, double *__restrict__ part)
{
    __shared__ double red[SEGK_NT / 64];
    __shared__ int sdone;
    typedef double dbl2 __attribute__((ext_vector_type(2)));
#define LDM(ptr) (NTL ? __builtin_nontemporal_load(ptr) : *(ptr))
    const int bid = (int)blockIdx.x;
    if (bid >= nsb) {
        // the short rows ride along in the same launch (independent of the segments): 16 lanes per row, p.t partial per block
        if (threadIdx.x == 0) sdone = ctrl->done;
        __syncthreads();
        if (sdone) return;
        const int g = threadIdx.x >> 4, l = threadIdx.x & 15;
        const int nb = gridDim.x - nsb;
        double acc = 0.0;
        for (int ridx = (bid - nsb) * (SEGK_NT / 16) + g; ridx < n_short; ridx += nb * (SEGK_NT / 16)) {
            const int row = short_rows[ridx];
            double s = 0.0;
            const RP p0 = rp[row], p1 = rp[row + 1];
            for (RP q = p0 + l; q < p1; q += 16) s += a[q] * p[ci[q]];
            s = group_sum<16>(s);
            if (l == 0) { t[row] = s; acc += p[row] * s; }
        }
        const double tot = block_sum_all<SEGK_NT>(acc, red);
        if (threadIdx.x == 0) part[bid - nsb] = tot;
        return;
    }
    if (ctrl->done) return;                  // no barrier on this path: a per-wave read is fine
    const int lane = threadIdx.x & 63;
    const int seg = bid * (SEGK_NT / 64) + (threadIdx.x >> 6);
    if (seg >= nseg) return;
    const RunDesc d = segs[seg];
    if (d.sr0 < 0) {                         // gather segment: entries outside long runs (<1 % of the matrix)
        double g = 0.0;
        const RP rowp0 = rp[long_rows[-1 - d.sr0]];
        for (int k = lane; k < d.len; k += 64) { const RP q = rowp0 + rem[d.pos + k]; g += a[q] * p[ci[q]]; }
        g = wave_sum(g);
        if (lane == 0) seg_part[seg] = g;
        return;
    }
    // 16-byte loads of the matrix stream: peel one entry if the segment starts on an odd element, then every lane
    // reads pairs (1 KiB per wave-instruction, 4 instructions in flight)
    const int head = (int)(d.pos & 1);
    const double *av = a + d.pos + head, *pv = pS + d.sr0 + head;
    const int len = d.len - head;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (head && lane == 0) s0 = a[d.pos] * pS[d.sr0];
    const int npair = len >> 1;
    const dbl2 *av2 = reinterpret_cast<const dbl2 *>(av);
    int k = lane;
    for (; k + 192 < npair; k += 256) {
        const dbl2 a0 = LDM(av2 + k), a1 = LDM(av2 + k + 64);
        const dbl2 a2 = LDM(av2 + k + 128), a3 = LDM(av2 + k + 192);
        s0 += a0.x * pv[2 * k] + a0.y * pv[2 * k + 1];
        s1 += a1.x * pv[2 * (k + 64)] + a1.y * pv[2 * (k + 64) + 1];
        s2 += a2.x * pv[2 * (k + 128)] + a2.y * pv[2 * (k + 128) + 1];
        s3 += a3.x * pv[2 * (k + 192)] + a3.y * pv[2 * (k + 192) + 1];
    }
    for (; k < npair; k += 64) { const dbl2 a0 = LDM(av2 + k); s0 += a0.x * pv[2 * k] + a0.y * pv[2 * k + 1]; }
    if ((len & 1) && lane == 1) s1 += av[len - 1] * pv[len - 1];
    double s = (s0 + s1) + (s2 + s3);
    s = wave_sum(s);
    if (lane == 0) seg_part[seg] = s;
#undef LDM
}

// row binning: flag long rows, build the two row lists
template <typename RP>
__global__ void k_row_flags(int m, const RP *rp, int *is_long, int *is_short)
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

// Internal entry: uniform_rows != 0 skips the binning (K: every row is short).  srank (optional, per row/column of
// the system: rank in the tunnelling set or -1) enables the dense-run view of the long rows.
template <typename RP>
static int cg_solve_jacobi_t(double *a, const RP *rp, const int *ci, long long nnz, int m, double *x, double *y,
                             int uniform_rows, const int *srank, int ns, int *iters_out, double *rr_out, int *iter_hint = nullptr)
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
        hipLaunchKernelGGL((k_row_flags<RP>), dim3((m + 255) / 256), dim3(256), 0, st, m, rp, fl, fs);
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
    // hot SpMV (k_spmv_ap): 1024-thread workgroups, short rows 64 per pass, long rows 16 per pass
    const int hs = n_short > 0 ? grid_for(n_short, SPMV_NT / 16) : 0;
    const int hl = n_long > 0 ? grid_for(n_long, SPMV_NT / 64) : 0;
    const int hl2 = n_long > 0 ? grid_for(n_long, SPMV_NT / 16) : 0;    // long rows in the two-stage (segment) mode
    int np_ap = hs + hl;

#define SPMV(MODE, vin, vout, a0, a1, partp)                                                                          \
    do {                                                                                                             \
        if (gs) hipLaunchKernelGGL((k_spmv<16, MODE, RP>), dim3(gs), dim3(CG_NT), 0, st, n_short, short_rows, rp, ci, a,  \
                                   vin, vout, a0, a1, partp, ctrl);                                                  \
        if (gl) hipLaunchKernelGGL((k_spmv<64, MODE, RP>), dim3(gl), dim3(CG_NT), 0, st, n_long, long_rows, rp, ci, a,    \
                                   vin, vout, a0, a1, (partp) ? (partp) + gs : nullptr, ctrl);                       \
    } while (0)

    // ---- dense-run view of the long rows ----
    const bool use_runs = srank && n_long > 0 && ns > 0;
    RunDesc *runs = nullptr, *segs = nullptr; int *nruns = nullptr, *rem = nullptr, *nrem = nullptr, *seg_off = nullptr; double *pS = nullptr, *seg_part = nullptr;
    int nseg = 0, nseg_loc = 0, seg_lo = 0; bool sharded = false; RowParts parts{}, *dparts = nullptr; double *xbuf = nullptr;
    if (use_runs) {
        runs = (RunDesc *)scratch(S_CG_RUNS, ((size_t)nnz / RUN_MIN_LEN + n_long + 2) * sizeof(RunDesc));
        rem = (int *)scratch(S_CG_REM, (size_t)nnz * 4);
        nruns = (int *)scratch(S_CG_NRUNS, (size_t)n_long * 2 * 4);
        pS = (double *)scratch(S_CG_PS, (size_t)(ns + 2) * 8);
        if (!runs || !rem || !nruns || !pS) return e.err_code;
        nrem = nruns + n_long;
        hipLaunchKernelGGL((k_build_runs<RP>), dim3((n_long + 3) / 4), dim3(256), 0, st, n_long, long_rows, rp, ci, srank, runs, nruns, rem, nrem, SEG_LEN);
        seg_off = (int *)scratch(S_CG_SEGOFF, (size_t)(n_long + 4) * 4);
        if (!seg_off) return e.err_code;
        int rc = dkmc_exclusive_scan_i32(nruns, seg_off, n_long, seg_off + n_long); if (rc) return rc;
        HIPCHK(hipMemcpyAsync(&nseg, seg_off + n_long, sizeof(int), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        // sharded solve (comm.hip): the long rows are dealt to the ranks at row boundaries, balanced by segment count; a
        // rank multiplies the segments of its rows and sums them per row; the row sums are what is exchanged
        if (comm_attached()) {
            sharded = true;
            const int nr = comm_nranks(), me = comm_rank();
            if (nr > MAX_RANKS) return dkmc_fail(46, "CG: more ranks than MAX_RANKS", __FILE__, __LINE__);
            std::vector<int> hoff((size_t)n_long + 1);
            HIPCHK(hipMemcpy(hoff.data(), seg_off, ((size_t)n_long + 1) * 4, hipMemcpyDeviceToHost));
            parts.n = nr; parts.lo[0] = 0;
            for (int r = 1; r < nr; ++r) {           // first row whose segments start at or beyond r/nr of the total
                const long long want = (long long)nseg * r / nr;
                parts.lo[r] = (int)(std::lower_bound(hoff.begin(), hoff.begin() + n_long, (int)want) - hoff.begin());
            }
            parts.lo[nr] = n_long;
            int mx = 0; for (int r = 0; r < nr; ++r) mx = std::max(mx, parts.lo[r + 1] - parts.lo[r]);
            parts.chunk = (mx + 1) & ~1;
            if (parts.chunk == 0) parts.chunk = 2;
            seg_lo = hoff[parts.lo[me]]; nseg_loc = hoff[parts.lo[me + 1]] - seg_lo;
            xbuf = (double *)scratch(S_CG_XCHG, std::max((size_t)nr * parts.chunk, (size_t)n_long + 2) * 8);
            dparts = (RowParts *)scratch(S_CG_PARTS, sizeof(RowParts));
            if (!xbuf || !dparts) return e.err_code;
            HIPCHK(hipMemcpy(dparts, &parts, sizeof(RowParts), hipMemcpyHostToDevice));
            e.stats.comm_ranks = nr; e.stats.comm_local_segments = nseg_loc; e.stats.comm_count_per_rank = parts.chunk;
        } else { nseg_loc = nseg; e.stats.comm_ranks = 0; }
        e.stats.spmv_segments = nseg;
        segs = (RunDesc *)scratch(S_CG_SEGS, (size_t)(nseg + 1) * sizeof(RunDesc));
        seg_part = (double *)scratch(S_CG_SEGPART, (size_t)(nseg + 1) * 8);
        if (!segs || !seg_part) return e.err_code;
        hipLaunchKernelGGL((k_compact_segs<RP>), dim3((n_long + 3) / 4), dim3(256), 0, st, n_long, long_rows, rp, (const RunDesc *)runs,
                           (const int *)nruns, (const int *)seg_off, segs);
        e.stats.spmv_tiles = 0; e.stats.spmv_tile_entries = 0;
    }
    e.stats.xt_subblocks = 0; e.stats.xt_local_subblocks = 0; e.stats.xt_items = 0;
    // blocks of k_spmv_segs: segments (a wave each), then the short rows (16 lanes each)
    const int nsb = (nseg_loc + SEGK_NT / 64 - 1) / (SEGK_NT / 64);
    const int hsA = (use_runs && n_short > 0) ? grid_for(n_short, SEGK_NT / 16) : 0;
    if (use_runs) np_ap = hsA + hl2;
    // ---- Jacobi scaling ----
    SPMV(M_DIAG, (const double *)nullptr, s, x, y, (double *)nullptr);
    SPMV(M_SCALE, (const double *)s, (double *)nullptr, (double *)nullptr, (double *)nullptr, (double *)nullptr);
    // ---- r = A y - x, p = -r ----
    SPMV(M_INIT, (const double *)y, r, x, p, part_rr);
    hipLaunchKernelGGL(k_cg_check0, dim3(1), dim3(CG_NT), 0, st, part_rr, np_spmv, ctrl, tol2);
    if (use_runs) hipLaunchKernelGGL(k_compact_pS, dim3((m + 255) / 256), dim3(256), 0, st, m, srank, (const double *)p, pS);
    KCHK();

    // ---- optional kernel profile: HIP events around every A*p launch (bench.py roofline) ----
    const bool prof = e.profiling && !uniform_rows;
    static hipEvent_t evs[4 * 64], evc[64 / PROF_STRIDE]; static bool evs_ready = false;
    double prof_short_ms = 0.0, prof_long_ms = 0.0, prof_comm_ms = 0.0; int prof_short_n = 0, prof_long_n = 0, prof_comm_n = 0;
    if (prof) {
        if (!evs_ready) { for (auto &ev : evs) HIPCHK(hipEventCreate(&ev)); for (auto &ev : evc) HIPCHK(hipEventCreate(&ev)); evs_ready = true; }
        std::vector<RP> hrp((size_t)m + 1);
        HIPCHK(hipMemcpy(hrp.data(), rp, ((size_t)m + 1) * sizeof(RP), hipMemcpyDeviceToHost));
        long long nl = 0, nsh = 0;
        for (int i = 0; i < m; ++i) { const long long c = hrp[i + 1] - hrp[i]; if (c > LONG_ROW_NNZ) nl += c; else nsh += c; }
        e.stats.spmv_long_nnz = nl; e.stats.spmv_short_nnz = nsh; e.stats.spmv_long_rows = n_long; e.stats.spmv_short_rows = n_short;
        e.stats.spmv_segments = nseg; e.stats.spmv_segment_entries = 0;
        if (use_runs && nseg > 0) {
            std::vector<RunDesc> hs_((size_t)nseg);
            HIPCHK(hipMemcpy(hs_.data(), segs, (size_t)nseg * sizeof(RunDesc), hipMemcpyDeviceToHost));
            long long tot = 0; for (auto &d : hs_) tot += d.len;
            e.stats.spmv_segment_entries = tot;
        }
    }
    // ---- iterations, launched in batches; the host polls the control block between batches ----
    // Batch plan: the iteration count of consecutive solves of the same system changes slowly (X: 666 +- 10 at 85 k sites), so
    // the first batch is sized just below the previous count; after it, short batches keep the no-op tail small.
    int it = 0, launched = 0;
    int batch = 8;
    if (iter_hint && *iter_hint > 24) batch = *iter_hint - 8;
    // matrix stream: default cache policy while the values of one sweep fit the 256 MiB Infinity Cache (they are re-read
    // every iteration), non-temporal beyond that (measured: 45 vs 53 us at 240 MB, 478 vs 456 us at 1.86 GB); a rank of a
    // sharded solve streams only its share
    const int seg_nt = nnz * 8 / (sharded ? comm_nranks() : 1) > (300ll << 20);
    CgCtrl h{};
    for (;;) {
        HIPCHK(hipMemcpyAsync(&h, ctrl, sizeof(CgCtrl), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (prof && launched) {       // only launches that did work (iteration index below the final count) are counted
            for (int b = 0; b < launched && b < 64; b += PROF_STRIDE) {
                if (it - launched + b >= h.iters) break;
                float ms = 0.f;
                if (use_runs) {
                    HIPCHK(hipEventElapsedTime(&ms, evs[4 * b], evs[4 * b + 1])); prof_long_ms += ms; ++prof_long_n;
                    if (sharded) { HIPCHK(hipEventElapsedTime(&ms, evs[4 * b + 1], evc[b / PROF_STRIDE])); prof_comm_ms += ms; ++prof_comm_n; }
                    HIPCHK(hipEventElapsedTime(&ms, evs[4 * b + 2], evs[4 * b + 3])); prof_short_ms += ms; ++prof_short_n;
                } else { HIPCHK(hipEventElapsedTime(&ms, evs[4 * b], evs[4 * b + 1])); prof_long_ms += ms; ++prof_long_n; }
            }
        }
        if (h.done) break;
        if (it >= 200000) { dkmc_fail(4, "CG: no convergence after 200000 iterations", __FILE__, __LINE__); break; }
        for (int b = 0; b < batch; ++b, ++it) {
            // sampled launches (1 in 8 of the first 64 of a batch) carry start/stop events of the dispatch itself
            // (hipExtLaunchKernelGGL): the kernel's own begin/end timestamps, no extra marker packets in the stream
            const bool pb = prof && b < 64 && (b % PROF_STRIDE == 0);
            hipEvent_t e0 = pb ? evs[4 * b] : nullptr, e1 = pb ? evs[4 * b + 1] : nullptr, e2 = pb ? evs[4 * b + 2] : nullptr, e3 = pb ? evs[4 * b + 3] : nullptr;
            if (use_runs) {
#define SEG_ARGS nseg_loc, (const RunDesc *)segs + seg_lo, (const double *)a, (const double *)pS, seg_part + seg_lo, (const CgCtrl *)ctrl, (const int *)rem, ci, \
                 (const double *)p, nsb, n_short, short_rows, rp, long_rows, t, part_pAp
                const dim3 sg(nsb + hsA);
                if (seg_nt) hipExtLaunchKernelGGL((k_spmv_segs<1, RP>), sg, dim3(SEGK_NT), 0, st, e0, e1, 0, SEG_ARGS);
                else hipExtLaunchKernelGGL((k_spmv_segs<0, RP>), sg, dim3(SEGK_NT), 0, st, e0, e1, 0, SEG_ARGS);
#undef SEG_ARGS
                if (sharded) {
                    // row sums of the owned rows -> exchange step -> t and the p.t partials on every rank.  Every rank enqueues
                    // exactly the same sequence of collectives (the batch plan and the stop decisions depend only on values
                    // that are identical on all ranks).
                    const int me = comm_rank(), r0 = parts.lo[me], r1 = parts.lo[me + 1];
                    if (r1 > r0) hipLaunchKernelGGL(k_rowsum_owner, dim3(grid_for(r1 - r0, SPMV_NT / 16)), dim3(SPMV_NT), 0, st, r0, r1, (const int *)nruns,
                                                    (const int *)seg_off, (const double *)seg_part, xbuf + (size_t)me * parts.chunk, ctrl);
                    if (int rc = comm_allgather_f64(xbuf, (size_t)parts.chunk)) return rc;
                    if (pb) HIPCHK(hipEventRecord(evc[b / PROF_STRIDE], st));
                    hipExtLaunchKernelGGL(k_rowsum_apply, dim3(hl2), dim3(SPMV_NT), 0, st, e2, e3, 0, n_long, long_rows, (const RowParts *)dparts, (const double *)xbuf,
                                          (const double *)p, t, part_pAp + hsA, (const CgCtrl *)ctrl);
                } else
                hipExtLaunchKernelGGL((k_spmv_ap<0, 1, RP>), dim3(hl2), dim3(SPMV_NT), 0, st, e2, e3, 0, 0, short_rows, 0, n_long, long_rows, rp, ci, (const double *)a,
                                      (const double *)p, t, part_pAp + hsA, (const CgCtrl *)ctrl, (const RunDesc *)runs, (const int *)nruns, (const int *)rem,
                                      (const int *)nrem, (const double *)seg_part, (const int *)seg_off);
            }
            else hipExtLaunchKernelGGL((k_spmv_ap<0, 0, RP>), dim3(np_ap), dim3(SPMV_NT), 0, st, e0, e1, 0, n_short, short_rows, hs, n_long, long_rows, rp, ci,
                                       (const double *)a, (const double *)p, t, part_pAp, (const CgCtrl *)ctrl, (const RunDesc *)runs, (const int *)nruns,
                                       (const int *)rem, (const int *)nrem, (const double *)pS, (const int *)seg_off);
            hipLaunchKernelGGL(k_cg_update, dim3(gv), dim3(CG_NT), 0, st, m, it, part_pAp, np_ap, p, t, y, r, part_rr, ctrl);
            hipLaunchKernelGGL(k_cg_direction, dim3(gv), dim3(CG_NT), 0, st, m, it, part_rr, gv, r, p, ctrl, tol2, use_runs ? srank : (const int *)nullptr, pS);
        }
        launched = batch;
        KCHK();
        if (iter_hint && *iter_hint > 24) batch = 8;           // after the sized first batch: short ones
        else if (batch < 64) batch *= 2;
    }
#undef SPMV
    if (prof) {
        e.stats.spmv_long_ms = prof_long_ms; e.stats.spmv_short_ms = prof_short_ms;
        e.stats.spmv_long_launches = prof_long_n; e.stats.spmv_short_launches = prof_short_n;
        e.stats.comm_ms = prof_comm_ms; e.stats.comm_launches = prof_comm_n;
    }
    // ---- un-scale the solution (:459) ----
    hipLaunchKernelGGL(k_vec_mul, dim3(gv), dim3(CG_NT), 0, st, m, y, s);
    KCHK();
    if (iters_out) *iters_out = h.iters;
    if (iter_hint) *iter_hint = h.iters;
    if (rr_out) *rr_out = h.rr[h.iters & 1];
    return e.err_code;
}

int cg_solve_jacobi(double *a, const int *rp, const int *ci, int nnz, int m, double *x, double *y,
                    int uniform_rows, const int *srank, int ns, int *iters_out, double *rr_out)
{
    return cg_solve_jacobi_t<int>(a, rp, ci, nnz, m, x, y, uniform_rows, srank, ns, iters_out, rr_out);
}
// 64-bit row pointers: the tunnelling block of X outgrows 2^31 non-zeros beyond ~4e5 sites
int cg_solve_jacobi64(double *a, const long long *rp, const int *ci, long long nnz, int m, double *x, double *y,
                      int uniform_rows, const int *srank, int ns, int *iters_out, double *rr_out)
{
    // batch sizing from the iteration count of the previous solve of X; the hint lives in the engine because attaching a
    // communicator resets it (the batch plan must be identical on all ranks of a sharded solve)
    return cg_solve_jacobi_t<long long>(a, rp, ci, nnz, m, x, y, uniform_rows, srank, ns, iters_out, rr_out, &eng().x_iter_hint);
}

extern "C" int dkmc_solve_sparse_CG_Jacobi(double *A, const int *rp, const int *ci, int nnz, int m, double *x, double *y,
                                           int *iters_out, double *rr_out)
{
    return cg_solve_jacobi_t<int>(A, rp, ci, nnz, m, x, y, 0, nullptr, 0, iters_out, rr_out);
}
