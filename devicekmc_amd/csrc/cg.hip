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
//   * X is symmetric and its tunnelling part is dense by classes: 32 x 256 blocks of it ("symmetric tiles", k_tile_count ff.)
//     are copied into tile-major storage once per solve and read ONCE per iteration for both triangles;
//   * row pointers are a template parameter (int for K, 64-bit for X whose non-zeros outgrow 2^31 at ~4e5 sites);
//   * with a communicator attached (comm.hip) the matrix stream is dealt to the ranks and one collective per iteration
//     completes the long rows' sums.
// HBM traffic per iteration in the CSR formulation: 12*nnz + 4*(m+1) + 96*m bytes; with the segments the
// matrix part of X drops to 8 B per entry, with the tiles to about 4 B per entry.
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
// per long row, everything stage 2 needs in one 16-byte load (tile mode)
struct __attribute__((aligned(16))) LRowMeta { int sr, nseg, segoff, row, wbeg, wend, kend, pad; };   // [wbeg, wend): windows of the row block's tiles; kend: 1 + last row block with a tile in the row's window
// symmetric tiles of the tunnelling block (see k_tile_count)
#define SEGK_NT 256            // workgroup of k_spmv_segs / k_spmv_tiles: 4 waves, one work item each
#define TILE_SEG_LEN 256       // tile mode: what the tiles leave behind is cut into pieces of at most this many entries (16 lanes each)
#define TILE_R 32
#define TILE_C 256
struct TileDesc { int k, w; };     // S-rows [32k, 32k+32) x S-cols [256w, 256w+256); values in tile-major storage, zero where X has no entry
#define TILE_MIN_FILL 0.1      // a grid cell becomes a tile when at least this fraction of its slots holds an entry.  In bytes the break-even is 0.5, but
                               // what a rejected cell leaves behind are short latency-bound pieces: measured at 235 k sites 333 us per launch at 0.6,
                               // 256 us at 0.3, 241 us at 0.1 (runs only: 444 us)

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
            for (RP p = p0 + l; p < p1; p += LPR) s += a[p] * vin[ci[p]];
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
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off, 16);
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
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off, 16);
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
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off, 16);
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

// Large tunnelling sets: a row's window can hold hundreds of tiles, and their column partials lie 2 KiB apart.  This pre-pass adds
// them with coalesced reads -- one workgroup per (window, slice of the row blocks), one thread per column -- into COLSUM_SLICES
// partial column sums per S-rank, which stage 2 then adds in slice order.
#define COLSUM_SLICES 8
__global__ __launch_bounds__(TILE_C) void k_tile_colsum(int ns, int nK, int nW, const int *__restrict__ kend, const double *__restrict__ colpart,
                                                        double *__restrict__ csum, int csum_pitch, const CgCtrl *ctrl)
{
    if (ctrl->done) return;
    const int w = blockIdx.x / COLSUM_SLICES, sl = blockIdx.x % COLSUM_SLICES, c = threadIdx.x;
    const int chunk = (nK + COLSUM_SLICES - 1) / COLSUM_SLICES;
    const int k0 = sl * chunk, k1 = min(min(k0 + chunk, nK), kend[w]);
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    const double *cp = colpart + (size_t)w * TILE_C + c;
    const size_t stride = (size_t)nW * TILE_C;
    int k = k0;
    for (; k + 3 < k1; k += 4) { s0 += cp[k * stride]; s1 += cp[(k + 1) * stride]; s2 += cp[(k + 2) * stride]; s3 += cp[(k + 3) * stride]; }
    for (; k < k1; ++k) s0 += cp[k * stride];
    if (w * TILE_C + c < ns) csum[(size_t)sl * csum_pitch + w * TILE_C + c] = (s0 + s1) + (s2 + s3);
}

// stage 2 in symmetric-tile mode: t[row] = the row's segment partials + the row partials of the tiles of its row block
// (ascending window) + the column partials of the tiles of its window (ascending row block), lane-strided over 16 lanes and
// combined in a fixed order; launch shape of the fused stage 2.  The partial arrays are indexed by the tile GRID position
// (k * nW + w; cells without a tile stay zero from the memset at the start of the solve), so that after the row's 32-byte
// descriptor every load address is known: two dependent loads per row instead of three.
__global__ __launch_bounds__(SPMV_NT) void k_rowsum_tiles(int n_long, const LRowMeta *__restrict__ meta, const double *__restrict__ seg_part,
                                                          int nW, const double *__restrict__ rowpart, const double *__restrict__ colpart,
                                                          const double *__restrict__ csum, int csum_pitch,
                                                          const double *__restrict__ p, double *__restrict__ t, double *__restrict__ part,
                                                          const CgCtrl *ctrl, double *__restrict__ xout)
{
    // xout != nullptr (sharded solve): only this rank's share of every long row's sum is formed and stored to xout[ridx]; the
    // all-reduce and k_xbuf_apply finish the row
    __shared__ double red[SPMV_NT / 64];
    __shared__ int sdone;
    if (threadIdx.x == 0) sdone = ctrl->done;
    __syncthreads();
    if (sdone) return;
    double acc = 0.0;
    const int g = threadIdx.x >> 4, l = threadIdx.x & 15;
    for (int ridx = blockIdx.x * (SPMV_NT / 16) + g; ridx < n_long; ridx += gridDim.x * (SPMV_NT / 16)) {
        const LRowMeta mt = meta[ridx];
        double s = 0.0;
        const double *sp = seg_part + mt.segoff;
        for (int j = l; j < mt.nseg; j += 16) s += sp[j];
        if (mt.sr >= 0) {
            const int k = mt.sr / TILE_R, w2 = mt.sr / TILE_C;
            const double *rpp = rowpart + ((size_t)k * nW) * TILE_R + (mt.sr % TILE_R);
            for (int base = mt.wbeg + l; base < mt.wend; base += 64) {
                double v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int w = base + 16 * u; v[u] = w < mt.wend ? rpp[(size_t)w * TILE_R] : 0.0; }
                s += (v[0] + v[1]) + (v[2] + v[3]);
            }
            const double *cpp = colpart + (size_t)w2 * TILE_C + (mt.sr % TILE_C);
            if (csum) { if (l < COLSUM_SLICES && mt.kend > 0) s += csum[(size_t)l * csum_pitch + mt.sr]; }      // pre-summed slices (k_tile_colsum)
            else for (int base = l; base < mt.kend; base += 64) {
                double v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int k2 = base + 16 * u; v[u] = k2 < mt.kend ? cpp[((size_t)k2 * nW) * TILE_C] : 0.0; }
                s += (v[0] + v[1]) + (v[2] + v[3]);
            }
        }
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off, 16);
        if (l == 0) { if (xout) xout[ridx] = s; else { t[mt.row] = s; acc += p[mt.row] * s; } }
    }
    if (xout) return;
    const double tot = block_sum_all<SPMV_NT>(acc, red);
    if (threadIdx.x == 0) part[blockIdx.x] = tot;
}
// sharded tile solve, after the all-reduce: t and the p.t partials from the completed row sums (launch shape of stage 2)
__global__ __launch_bounds__(SPMV_NT) void k_xbuf_apply(int n_long, const LRowMeta *__restrict__ meta, const double *__restrict__ xbuf,
                                                        const double *__restrict__ p, double *__restrict__ t, double *__restrict__ part, const CgCtrl *ctrl)
{
    __shared__ double red[SPMV_NT / 64];
    __shared__ int sdone;
    if (threadIdx.x == 0) sdone = ctrl->done;
    __syncthreads();
    if (sdone) return;
    double acc = 0.0;
    const int g = threadIdx.x >> 4, l = threadIdx.x & 15;         // thread-to-row mapping of k_rowsum_tiles: same p.t partials
    for (int ridx = blockIdx.x * (SPMV_NT / 16) + g; ridx < n_long; ridx += gridDim.x * (SPMV_NT / 16)) {
        if (l == 0) { const int row = meta[ridx].row; const double s = xbuf[ridx]; t[row] = s; acc += p[row] * s; }
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
                                                    int *__restrict__ rem, int *__restrict__ nrem, int seg_len, int diag_break)
{
    const int lane = threadIdx.x & 63;
    const int ridx = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ridx >= n_long) return;
    const int row = long_rows[ridx];
    // symmetric-tile mode: a run never crosses the row's own S-rank, so that every run is wholly in one triangle
    const int self_sr = diag_break ? srank[row] : -1;
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
        const bool brk = valid && (q == p0 || sr < 0 || prev < 0 || sr != prev + 1 || (self_sr >= 0 && (sr == self_sr || prev == self_sr)));
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

// ---- symmetric tiles of the tunnelling block --------------------------------------------------------------------------------
// X is symmetric and its tunnelling block is dense by classes (contact x contact, vacancy x contact: nearly every pair present), so
// nearly every long-run entry a_ij has its mirror a_ji stored in row j.  The S x S part is covered by a grid of TILE_R (32) S-rows x
// TILE_C (256) S-columns; a cell strictly above the diagonal becomes a *tile* when it is at least TILE_MIN_FILL full and its
// mirror cell holds exactly as many entries.  The values of a tile are copied (after the Jacobi scaling) into tile-major storage,
// 32 strips of 256 doubles, zero where X has no entry: one wave reads the 64 KiB once per iteration and forms both the row
// products (t_i += a_ij p_j) and the column products (t_j += a_ij p_i).  All entries of X inside a tile's cell, and all mirror
// entries inside its mirror cell, are cut out of their rows' segment lists and are never read in the loop.  Everything else
// (cells near the diagonal, sparse cells, rows 0/1, columns outside S) stays in the segment path, both triangles.
// Partial sums: 32 row sums and 256 column sums per tile (3.5 % of the bytes read), combined in stage 2.
template <typename RP>
__device__ __forceinline__ const RunDesc *row_runs(int ridx, const int *long_rows, const RP *rp, const RunDesc *runs)
{
    return runs + ((long long)(rp[long_rows[ridx]] / RUN_MIN_LEN) + ridx);
}
__device__ __forceinline__ bool cell_eligible(int k, int w) { return w * TILE_C >= k * TILE_R + TILE_R; }      // strictly above the diagonal

// entries per grid cell (choice heuristic): cntU[k * nW + w] = entries of the upper long runs of the rows of block k inside window w.
// One thread per long row; integer atomics (order-independent).
template <typename RP>
__global__ __launch_bounds__(256) void k_tile_count(int n_long, const int *__restrict__ long_rows, const RP *__restrict__ rp, const int *__restrict__ srank,
                                                    const RunDesc *__restrict__ runs, const int *__restrict__ nruns, int nW, int *__restrict__ cntU)
{
    const int ridx = blockIdx.x * blockDim.x + threadIdx.x;
    if (ridx >= n_long) return;
    const int s = srank[long_rows[ridx]];
    if (s < 0) return;
    const RunDesc *rr = row_runs(ridx, long_rows, rp, runs);
    const int n = nruns[ridx], k = s / TILE_R;
    for (int q = 0; q < n; ++q) {
        const RunDesc d = rr[q];
        if (d.sr0 < 0 || d.sr0 <= s) continue;
        const int lo = d.sr0, hi = d.sr0 + d.len;
        for (int w = lo / TILE_C; w * TILE_C < hi; ++w)
            if (cell_eligible(k, w)) atomicAdd(&cntU[(long long)k * nW + w], min(hi, w * TILE_C + TILE_C) - max(lo, w * TILE_C));
    }
}

// choose[cell] = 1 when the cell becomes a tile; covered[cell] = entries of X it takes out of the segment path (both triangles)
__global__ void k_tile_choose(int ns, int nK, int nW, double min_fill, const int *__restrict__ cntU, int *__restrict__ choose, int *__restrict__ covered)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)nK * nW) return;
    const int k = (int)(i / nW), w = (int)(i % nW);
    const bool ok = cell_eligible(k, w) && (double)cntU[i] >= min_fill * TILE_R * TILE_C;
    choose[i] = ok ? 1 : 0;
    covered[i] = ok ? 2 * cntU[i] : 0;
}
// Is the S x S entry (row rank s, column rank sc) inside a tile?  Returns the cell index or -1; upper = entry above the diagonal.
__device__ __forceinline__ long long tiled_cell(int s, int sc, int nW, const int *__restrict__ choose, bool &upper)
{
    if (s < 0 || sc < 0 || s == sc) return -1;
    upper = sc > s;
    const int k = (upper ? s : sc) / TILE_R, w = (upper ? sc : s) / TILE_C;
    if (!cell_eligible(k, w)) return -1;
    const long long cell = (long long)k * nW + w;
    return choose[cell] ? cell : -1;
}
__global__ void k_tile_list(int nK, int nW, const int *__restrict__ choose, const int *__restrict__ toff, TileDesc *__restrict__ tiles)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (long long)nK * nW && choose[i]) { TileDesc d; d.k = (int)(i / nW); d.w = (int)(i % nW); tiles[toff[i]] = d; }
}

// copy the (scaled) values of the upper entries inside tiles into the tile-major storage; one wave per long row
template <typename RP>
__global__ __launch_bounds__(256) void k_tile_scatter(int n_long, const int *__restrict__ long_rows, const RP *__restrict__ rp, const int *__restrict__ srank,
                                                      const RunDesc *__restrict__ runs, const int *__restrict__ nruns, int nW,
                                                      const int *__restrict__ choose, const int *__restrict__ toff, const double *__restrict__ a,
                                                      double *__restrict__ tval, int *__restrict__ chk)
{
    const int lane = threadIdx.x & 63;
    const int ridx = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ridx >= n_long) return;
    const int s = srank[long_rows[ridx]];
    if (s < 0) return;
    const RunDesc *rr = row_runs(ridx, long_rows, rp, runs);
    const int n = nruns[ridx], k = s / TILE_R;
    for (int q = 0; q < n; ++q) {
        const RunDesc d = rr[q];
        if (d.sr0 < 0 || d.sr0 <= s) continue;                       // upper runs only
        const int lo = d.sr0, hi = d.sr0 + d.len;
        for (int w = lo / TILE_C; w * TILE_C < hi; ++w) {
            const long long cell = (long long)k * nW + w;
            if (!choose[cell]) continue;
            const int c0 = max(lo, w * TILE_C), c1 = min(hi, w * TILE_C + TILE_C);
            double *dst = tval + ((size_t)toff[cell] * TILE_R + (s % TILE_R)) * TILE_C;       // the strip of this row in the tile
            const double *src = a + d.pos;                                                     // a[pos + (c - lo)] is column rank c
            for (int c = c0 + lane; c < c1; c += 64) dst[c - w * TILE_C] = src[c - lo];
            if (lane == 0) atomicAdd(&chk[cell], c1 - c0);              // upper entries put into the tile (balanced by the mirror entries removed)
        }
    }
}

// lsr: long-row index -> S-rank (or -1)
__global__ void k_lsr(int n_long, const int *__restrict__ long_rows, const int *__restrict__ srank, int *__restrict__ lsr)
{
    const int ridx = blockIdx.x * blockDim.x + threadIdx.x;
    if (ridx < n_long) lsr[ridx] = srank[long_rows[ridx]];
}
// per row block: range of windows that hold a tile; per window: 1 + the last row block that holds a tile (0 = none)
__global__ void k_tile_ranges(int nK, int nW, const int *__restrict__ dense, int *__restrict__ wbeg, int *__restrict__ wend, int *__restrict__ kend)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nK) {
        int b = nW, e = 0;
        for (int w = 0; w < nW; ++w) if (dense[(long long)i * nW + w]) { b = min(b, w); e = w + 1; }
        wbeg[i] = b; wend[i] = e;
    }
    if (i < nW) {
        int e = 0;
        for (int k = 0; k < nK; ++k) if (dense[(long long)k * nW + i]) e = k + 1;
        kend[i] = e;
    }
}
__global__ void k_lrow_meta(int n_long, const int *__restrict__ long_rows, const int *__restrict__ lsr, const int *__restrict__ nsegs,
                            const int *__restrict__ seg_off, const int *__restrict__ wbeg, const int *__restrict__ wend, const int *__restrict__ kend,
                            int own_lo, int own_hi, LRowMeta *__restrict__ meta)
{
    const int ridx = blockIdx.x * blockDim.x + threadIdx.x;
    if (ridx >= n_long) return;
    // sharded solve: the segments of rows outside [own_lo, own_hi) belong to other ranks (their partials here are stale)
    LRowMeta m; m.sr = lsr[ridx]; m.nseg = (ridx >= own_lo && ridx < own_hi) ? nsegs[ridx] : 0; m.segoff = seg_off[ridx]; m.row = long_rows[ridx]; m.wbeg = 0; m.wend = 0; m.kend = 0; m.pad = 0;
    if (m.sr >= 0) { m.wbeg = wbeg[m.sr / TILE_R]; m.wend = wend[m.sr / TILE_R]; m.kend = kend[m.sr / TILE_C]; }
    meta[ridx] = m;
}

// Segment list of one long row = its raw runs minus everything the tiles cover, cut into <= seg_len pieces, followed by its
// gather segments.  One thread per row; FILL = 0 counts, FILL = 1 writes at seg_off[ridx].
template <int FILL, typename RP>
__global__ __launch_bounds__(256) void k_emit_segs(int n_long, const int *__restrict__ long_rows, const RP *__restrict__ rp, const int *__restrict__ srank,
                                                   const RunDesc *__restrict__ runs, const int *__restrict__ nruns, int ns, int nW,
                                                   const int *__restrict__ dense, int seg_len, int *__restrict__ nsegs, const int *__restrict__ seg_off,
                                                   RunDesc *__restrict__ segs, const int *__restrict__ goff, int *__restrict__ chk)
{
    const int ridx = blockIdx.x * blockDim.x + threadIdx.x;
    if (ridx >= n_long) return;
    const int s = srank[long_rows[ridx]];
    const RunDesc *rr = row_runs(ridx, long_rows, rp, runs);
    const int n = nruns[ridx];
    int cnt = 0;
    RunDesc *out = FILL ? segs + seg_off[ridx] : nullptr;
    auto emit = [&](long long pos, int sr0, int len) {              // one untiled stretch, cut into segments
        for (int c = 0; c < len; c += seg_len) {
            if (FILL) { RunDesc d; d.pos = pos + c; d.len = min(seg_len, len - c); d.sr0 = sr0 + c; out[cnt] = d; }
            ++cnt;
        }
    };
    for (int q = 0; q < n; ++q) {
        const RunDesc d = rr[q];
        if (d.sr0 < 0) {                                                          // gather segment
            if (goff == nullptr) { if (FILL) out[cnt] = d; ++cnt; continue; }    // (legacy layout: unchanged)
            // packed layout: position in the compact (value, column) arrays of the remainder entries, 64 entries per wave
            const long long base = (long long)goff[ridx] + (d.pos - (long long)rp[long_rows[ridx]]);
            for (int c = 0; c < d.len; c += 64) {
                if (FILL) { RunDesc g; g.pos = base + c; g.len = min(64, d.len - c); g.sr0 = d.sr0; out[cnt] = g; }
                ++cnt;
            }
            continue;
        }
        const int lo = d.sr0, hi = d.sr0 + d.len;
        if (s < 0 || dense == nullptr) { emit(d.pos, lo, d.len); continue; }
        int start = lo;                                                         // start of the current untiled stretch
        if (lo > s) {
            // upper run: the part inside window w is skipped when cell (s/TILE_R, w) is a tile
            const int k = s / TILE_R;
            for (int w = lo / TILE_C; w * TILE_C < hi; ++w) {
                if (!cell_eligible(k, w) || !dense[(long long)k * nW + w]) continue;
                const int c0 = max(lo, w * TILE_C), c1 = min(hi, w * TILE_C + TILE_C);
                if (c0 > start) emit(d.pos + (start - lo), start, c0 - start);
                start = c1;
            }
        } else {
            // lower run: the part inside the TILE_R columns of row block k' is skipped when cell (k', s/TILE_C) is a tile (this row is
            // one of its columns)
            const int w = s / TILE_C;
            for (int k = lo / TILE_R; k * TILE_R < hi; ++k) {
                if (!cell_eligible(k, w) || !dense[(long long)k * nW + w]) continue;
                const int c0 = max(lo, k * TILE_R), c1 = min(hi, k * TILE_R + TILE_R);
                if (c0 > start) emit(d.pos + (start - lo), start, c0 - start);
                start = c1;
                if (FILL) atomicSub(&chk[(long long)k * nW + w], c1 - c0);      // mirror entries removed
            }
        }
        if (hi > start) emit(d.pos + (start - lo), start, hi - start);
    }
    if (!FILL) nsegs[ridx] = cnt;
}

// Packed copies for the latency-bound roles of the tile-mode launch (fewer dependent loads per wave: descriptor -> packed
// (value, column) -> p): the remainder entries of the long rows and the short rows, copied after the Jacobi scaling.
// An entry of S x S that lies in a tile is represented by the tile alone, wherever X stores it: long runs are cut in k_emit_segs;
// here the remainder entries and the short rows drop theirs (value 0, column 0 in the packed copy) and, for the upper ones, put
// the value into the tile.  chk counts upper entries placed minus mirror entries removed per cell: all zero iff X is
// structurally symmetric where it is tiled (checked on the host once per solve).
__device__ __forceinline__ bool tile_take(int s, int sc, double v, int nW, const int *__restrict__ choose, const int *__restrict__ toff,
                                          double *__restrict__ tval, int *__restrict__ chk)
{
    bool upper;
    const long long cell = tiled_cell(s, sc, nW, choose, upper);
    if (cell < 0) return false;
    if (upper) { tval[((size_t)toff[cell] * TILE_R + (s % TILE_R)) * TILE_C + (sc % TILE_C)] = v; atomicAdd(&chk[cell], 1); }
    else atomicSub(&chk[cell], 1);
    return true;
}
template <typename RP>
__global__ __launch_bounds__(256) void k_pack_rem(int n_long, const int *__restrict__ long_rows, const RP *__restrict__ rp, const int *__restrict__ ci,
                                                  const double *__restrict__ a, const int *__restrict__ rem, const int *__restrict__ nrem,
                                                  const int *__restrict__ goff, double *__restrict__ gval, int *__restrict__ gcol,
                                                  const int *__restrict__ srank, int nW, const int *__restrict__ choose, const int *__restrict__ toff,
                                                  double *__restrict__ tval, int *__restrict__ chk)
{
    const int lane = threadIdx.x & 63;
    const int ridx = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ridx >= n_long) return;
    const int row = long_rows[ridx];
    const RP p0 = rp[row];
    const int n = nrem[ridx], o = goff[ridx], s = srank[row];
    for (int k = lane; k < n; k += 64) {
        const RP q = p0 + rem[p0 + k];
        double v = a[q]; int c = ci[q];
        if (tile_take(s, srank[c], v, nW, choose, toff, tval, chk)) { v = 0.0; c = 0; }
        gval[o + k] = v; gcol[o + k] = c;
    }
}
template <typename RP>
__global__ void k_short_len(int n_short, const int *__restrict__ short_rows, const RP *__restrict__ rp, int *__restrict__ len)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_short) { const int row = short_rows[i]; len[i] = (int)(rp[row + 1] - rp[row]); }
}
template <typename RP>
__global__ __launch_bounds__(256) void k_pack_short(int n_short, const int *__restrict__ short_rows, const RP *__restrict__ rp, const int *__restrict__ ci,
                                                    const double *__restrict__ a, const int *__restrict__ srp, double *__restrict__ sval, int *__restrict__ scol,
                                                    const int *__restrict__ srank, int nW, const int *__restrict__ choose, const int *__restrict__ toff,
                                                    double *__restrict__ tval, int *__restrict__ chk)
{
    const int g = threadIdx.x >> 4, l = threadIdx.x & 15;
    const int ridx = blockIdx.x * 16 + g;
    if (ridx >= n_short) return;
    const int row = short_rows[ridx];
    const RP p0 = rp[row]; const int n = (int)(rp[row + 1] - p0), o = srp[ridx], s = srank[row];
    for (int k = l; k < n; k += 16) {
        double v = a[p0 + k]; int c = ci[p0 + k];
        if (s >= 0 && tile_take(s, srank[c], v, nW, choose, toff, tval, chk)) { v = 0.0; c = 0; }
        sval[o + k] = v; scol[o + k] = c;
    }
}
__global__ void k_chk_max(long long n, const int *__restrict__ chk, int *__restrict__ out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && chk[i] != 0) atomicMax(out, abs(chk[i]));
}

// stage 1 of the long-row product: one wave64 per segment, seg_part[seg] = sum a[pos+k] * pS[sr0+k].
// Streams 8 B per entry, 4 independent 1-KiB strips in flight per wave; every wave has the same amount of work.
// Default cache policy on the matrix stream (NTL = 0): the same 240 MB are re-read every CG iteration and partly stay in
// the 256 MiB Infinity Cache -- measured 45 us per launch against 53 us with non-temporal loads (NTL = 1, DKMC_SPMV_VAR=3).
template <int NTL, typename RP, int TILES>
__global__ __launch_bounds__(SEGK_NT) void k_spmv_segs(int nseg, const RunDesc *__restrict__ segs, const double *__restrict__ a,
                                                       const double *__restrict__ pS, double *__restrict__ seg_part, const CgCtrl *ctrl,
                                                       const int *__restrict__ rem, const int *__restrict__ ci, const double *__restrict__ p,
                                                       int nsb, int n_short, const int *__restrict__ short_rows, const RP *__restrict__ rp,
                                                       const int *__restrict__ long_rows,
                                                       double *__restrict__ t, double *__restrict__ part,
                                                       int ntb, int ntiles, int nW_t, int ns_t, const TileDesc *__restrict__ tiles, const double *__restrict__ tval,
                                                       double *__restrict__ rowpart, double *__restrict__ colpart,
                                                       const double *__restrict__ gval, const int *__restrict__ gcol,
                                                       const int *__restrict__ srp, const double *__restrict__ sval, const int *__restrict__ scol)
{
    __shared__ double red[SEGK_NT / 64];
    __shared__ int sdone;
    typedef double dbl2 __attribute__((ext_vector_type(2)));
#define LDM(ptr) (NTL ? __builtin_nontemporal_load(ptr) : *(ptr))
    if (TILES && (int)blockIdx.x < ntb) {
        // symmetric tiles (FIRST ntb blocks, so that the bandwidth-bound part of the launch starts at once and the latency-bound
        // segment / short-row blocks fill in behind it; one wave per tile; see k_tile_count): 32 strips of 256 values, contiguous
        // in the tile-major storage, read once, give the row products and the column products (four columns per lane,
        // accumulated in registers over the strips).  Only instantiated (TILES = 1) in tile mode: the extra registers cost the
        // segment waves occupancy, which does not matter once the tiles carry most of the matrix.
        if (ctrl->done) return;
        const int lane = threadIdx.x & 63;
        // the tile index is wave-uniform: say so (scalar loads for the descriptor and the strip addresses)
        const int tile = __builtin_amdgcn_readfirstlane((int)blockIdx.x * (SEGK_NT / 64) + (int)(threadIdx.x >> 6));
        if (tile >= ntiles) return;
        const TileDesc td = tiles[tile];
        const size_t cell = (size_t)td.k * nW_t + td.w;                    // position in the tile grid: where the partial sums go
        const double *pc = pS + (size_t)td.w * TILE_C, *pr = pS + (size_t)td.k * TILE_R;
        const int ncols = min(TILE_C, ns_t - td.w * TILE_C);               // the last window may be narrower (its slots beyond are zero)
        const double *tv = tval + (size_t)tile * TILE_R * TILE_C;
        double pcx[2], pcy[2], cax[2], cay[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int col = 2 * lane + 128 * u;
            pcx[u] = col < ncols ? pc[col] : 0.0; pcy[u] = col + 1 < ncols ? pc[col + 1] : 0.0;
            cax[u] = 0.0; cay[u] = 0.0;
        }
        // Strips in 4 phases of 8 (a real loop, so that the register budget stays at one phase): the 16 16-byte loads of a phase
        // are issued together, 16 KiB in flight per wave.  Row sums: 32 sums over 64 lanes with 32 shuffles instead of 32 x 6.  In
        // every butterfly step a lane keeps the half of its values whose index bit matches its lane bit and adds the partner's:
        // xor 32, 16, 8 fold the 8 strips of a phase into one value per lane, xor 4, 2 fold the 4 phases, xor 1 completes the sum.
        // Fixed order: deterministic.
        double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
#pragma unroll 1
        for (int ph = 0; ph < TILE_R / 8; ++ph) {
            double x0[8], y0[8], x1[8], y1[8], ra[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const dbl2 *strip = reinterpret_cast<const dbl2 *>(tv + (size_t)(8 * ph + q) * TILE_C);
                const dbl2 v0 = LDM(strip + lane), v1 = LDM(strip + 64 + lane);
                x0[q] = v0.x; y0[q] = v0.y; x1[q] = v1.x; y1[q] = v1.y;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const double prow = pr[8 * ph + q];
                ra[q] = (x0[q] * pcx[0] + y0[q] * pcy[0]) + (x1[q] * pcx[1] + y1[q] * pcy[1]);
                cax[0] += x0[q] * prow; cay[0] += y0[q] * prow; cax[1] += x1[q] * prow; cay[1] += y1[q] * prow;
            }
#pragma unroll
            for (int half = 4, bit = 32; half >= 1; half >>= 1, bit >>= 1) {
                const bool up = (lane & bit) != 0;
#pragma unroll
                for (int j = 0; j < half; ++j) {
                    const double keep = up ? ra[j + half] : ra[j];
                    const double send = up ? ra[j] : ra[j + half];
                    ra[j] = keep + __shfl_xor(send, bit, WAVE);
                }
            }
            acc0 = ph == 0 ? ra[0] : acc0; acc1 = ph == 1 ? ra[0] : acc1; acc2 = ph == 2 ? ra[0] : acc2; acc3 = ph == 3 ? ra[0] : acc3;
        }
        {
            const bool up4 = (lane & 4) != 0, up2 = (lane & 2) != 0;
            const double b0 = (up4 ? acc2 : acc0) + __shfl_xor(up4 ? acc0 : acc2, 4, WAVE);      // phases 0|2 by lane bit 2
            const double b1 = (up4 ? acc3 : acc1) + __shfl_xor(up4 ? acc1 : acc3, 4, WAVE);      // phases 1|3
            const double c0 = (up2 ? b1 : b0) + __shfl_xor(up2 ? b0 : b1, 2, WAVE);              // +1 by lane bit 1
            const double rsum = c0 + __shfl_xor(c0, 1, WAVE);
            // strip of this lane: phase = 2*bit2 + bit1, index in the phase = 4*bit5 + 2*bit4 + bit3
            const int rr = 8 * (((lane >> 2) & 1) * 2 + ((lane >> 1) & 1)) + ((lane >> 5) & 1) * 4 + ((lane >> 4) & 1) * 2 + ((lane >> 3) & 1);
            if ((lane & 1) == 0) rowpart[cell * TILE_R + rr] = rsum;
        }
        double *cp = colpart + cell * TILE_C;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int col = 2 * lane + 128 * u;
            if (col + 1 < ncols) { dbl2 v; v.x = cax[u]; v.y = cay[u]; *reinterpret_cast<dbl2 *>(cp + col) = v; }
            else if (col < ncols) cp[col] = cax[u];
        }
        return;
    }
    const int bid = (int)blockIdx.x - (TILES ? ntb : 0);          // block index among the segment + short-row blocks
    if (bid >= nsb) {
        // the short rows ride along in the same launch (independent of the segments): 16 lanes per row, p.t partial per block
        if (threadIdx.x == 0) sdone = ctrl->done;
        __syncthreads();
        if (sdone) return;
        if (TILES) {
            // tile mode: this work is no longer hidden behind a long matrix stream, and what it costs is dependent loads per wave
            // times the number of wave rounds: 8 lanes per row (the rows average 17 entries) from the packed copy, one row per
            // group and launch, so that half as many waves go through the descriptor -> (value, column) -> p chain once
            const int g8 = threadIdx.x >> 3, l8 = threadIdx.x & 7;
            const int nb8 = gridDim.x - nsb - ntb;
            double acc = 0.0;
            for (int ridx = (bid - nsb) * (SEGK_NT / 8) + g8; ridx < n_short; ridx += nb8 * (SEGK_NT / 8)) {      // one pass unless the grid is capped
                const int q0 = srp[ridx], q1 = srp[ridx + 1];
                double s = 0.0;
                for (int q = q0 + l8; q < q1; q += 8) s += sval[q] * p[scol[q]];
                s += __shfl_xor(s, 4, 8); s += __shfl_xor(s, 2, 8); s += __shfl_xor(s, 1, 8);
                if (l8 == 0) { const int row = short_rows[ridx]; t[row] = s; acc += p[row] * s; }
            }
            const double tot = block_sum_all<SEGK_NT>(acc, red);
            if (threadIdx.x == 0) part[bid - nsb] = tot;
            return;
        }
        const int g = threadIdx.x >> 4, l = threadIdx.x & 15;
        const int nb = gridDim.x - nsb;
        double acc = 0.0;
        for (int ridx = (bid - nsb) * (SEGK_NT / 16) + g; ridx < n_short; ridx += nb * (SEGK_NT / 16)) {
            const int row = short_rows[ridx];
            double s = 0.0;
            const RP p0 = rp[row], p1 = rp[row + 1];
            for (RP q = p0 + l; q < p1; q += 16) s += a[q] * p[ci[q]];
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off, 16);
            if (l == 0) { t[row] = s; acc += p[row] * s; }
        }
        const double tot = block_sum_all<SEGK_NT>(acc, red);
        if (threadIdx.x == 0) part[bid - nsb] = tot;
        return;
    }
    if (ctrl->done) return;                  // no barrier on this path: a per-wave read is fine
    if (TILES) {
        // tile mode: what the tiles leave behind are short pieces (63 entries on average at 85 k sites): 16 lanes per segment,
        // 16 segments per workgroup, for the same reason as above
        const int l16 = threadIdx.x & 15;
        const int seg = bid * (SEGK_NT / 16) + (threadIdx.x >> 4);
        if (seg >= nseg) return;
        const RunDesc d = segs[seg];
        double g = 0.0;
        if (d.sr0 < 0) { for (int k = l16; k < d.len; k += 16) g += gval[d.pos + k] * p[gcol[d.pos + k]]; }      // packed remainder entries
        else { const double *av = a + d.pos, *pv = pS + d.sr0; for (int k = l16; k < d.len; k += 16) g += LDM(av + k) * pv[k]; }
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) g += __shfl_xor(g, off, 16);
        if (l16 == 0) seg_part[seg] = g;
        return;
    }
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
    static const int use_runs_env = getenv("DKMC_NO_RUNS") ? 0 : 1;
    // experiments only; the descriptor capacity per row (nnz/RUN_MIN_LEN + 1) needs segments of at least 2 * RUN_MIN_LEN entries
    static const int seg_len = getenv("DKMC_SEG_LEN") ? (atoi(getenv("DKMC_SEG_LEN")) < 2 * RUN_MIN_LEN ? 2 * RUN_MIN_LEN : atoi(getenv("DKMC_SEG_LEN"))) : SEG_LEN;
    const bool use_runs = use_runs_env && srank && n_long > 0 && ns > 0;
    RunDesc *runs = nullptr, *segs = nullptr; int *nruns = nullptr, *rem = nullptr, *nrem = nullptr, *seg_off = nullptr; double *pS = nullptr, *seg_part = nullptr;
    static const double tile_min_fill = getenv("DKMC_TILE_FILL") ? atof(getenv("DKMC_TILE_FILL")) : TILE_MIN_FILL;
    static const double tile_min_cover = getenv("DKMC_TILE_COVER") ? atof(getenv("DKMC_TILE_COVER")) : 0.8;   // fraction of X that must sit in tiles
    bool use_tiles = false; int nK = 0, nW = 0, ntiles = 0; int *dense = nullptr, *toff = nullptr, *nsegs = nullptr;
    TileDesc *tiles = nullptr; double *rowpart = nullptr, *colpart = nullptr, *tval = nullptr, *csum = nullptr; int csum_pitch = 0; int *trange = nullptr, *lsr = nullptr, *chk = nullptr, *goff = nullptr, *gcol = nullptr, *srp = nullptr, *scol = nullptr;
    double *gval = nullptr, *sval = nullptr; LRowMeta *lmeta = nullptr;
    int nseg = 0, nseg_loc = 0, seg_lo = 0, tile_lo = 0, ntiles_loc = 0; bool sharded = false; RowParts parts{}, *dparts = nullptr; double *xbuf = nullptr;
    if (use_runs) {
        runs = (RunDesc *)scratch(S_CG_RUNS, ((size_t)nnz / RUN_MIN_LEN + n_long + 2) * sizeof(RunDesc));
        rem = (int *)scratch(S_CG_REM, (size_t)nnz * 4);
        nruns = (int *)scratch(S_CG_NRUNS, (size_t)n_long * 2 * 4);
        pS = (double *)scratch(S_CG_PS, (size_t)(ns + TILE_C) * 8);      // + zero padding for the last tile row block / window
        if (!runs || !rem || !nruns || !pS) return e.err_code;
        nrem = nruns + n_long;
        // symmetric tiles (also in the sharded solve: then one all-reduce per iteration instead of the bit-identical all-gather scheme)
        use_tiles = e.symmetric_tiles && ns > TILE_C;
        hipLaunchKernelGGL((k_build_runs<RP>), dim3((n_long + 3) / 4), dim3(256), 0, st, n_long, long_rows, rp, ci, srank, runs, nruns, rem, nrem,
                           use_tiles ? 0x7fffffff : seg_len, use_tiles ? 1 : 0);
        seg_off = (int *)scratch(S_CG_SEGOFF, (size_t)(n_long + 4) * 4);
        if (!seg_off) return e.err_code;
        int rc = 0;
        if (use_tiles) {
            // raw runs -> entries per grid cell -> choice of tiles -> segment list of what the tiles do not cover
            nK = (ns + TILE_R - 1) / TILE_R; nW = (ns + TILE_C - 1) / TILE_C;
            const long long ncand = (long long)nK * nW;
            if (ncand > 0x7fffff00ll) return dkmc_fail(47, "CG: too many tile candidates", __FILE__, __LINE__);
            dense = (int *)scratch(S_CG_TDENSE, (size_t)(ncand + 4) * 4);
            toff = (int *)scratch(S_CG_TOFF, (size_t)(ncand + 4) * 4);
            nsegs = (int *)scratch(S_CG_NSEGS, (size_t)(n_long + 4) * 4);
            lsr = (int *)scratch(S_CG_LSR, (size_t)(n_long + 4) * 4);
            int *cnt = (int *)scratch(S_CG_S2R, (size_t)(3 * ncand + 8) * 4);                 // cntU | symmetry check | covered
            long long *cov = (long long *)scratch(S_MISC3, (size_t)(ncand + 4) * 8);
            if (!dense || !toff || !nsegs || !lsr || !cnt || !cov) return e.err_code;
            HIPCHK(hipMemsetAsync(cnt, 0, (size_t)(2 * ncand) * 4, st));
            hipLaunchKernelGGL(k_lsr, dim3((n_long + 255) / 256), dim3(256), 0, st, n_long, long_rows, srank, lsr);
            chk = cnt + ncand;
            hipLaunchKernelGGL((k_tile_count<RP>), dim3((n_long + 255) / 256), dim3(256), 0, st, n_long, long_rows, rp, srank, (const RunDesc *)runs,
                               (const int *)nruns, nW, cnt);
            hipLaunchKernelGGL(k_tile_choose, dim3((unsigned)((ncand + 255) / 256)), dim3(256), 0, st, ns, nK, nW, tile_min_fill, (const int *)cnt, dense, cnt + 2 * ncand);
            rc = dkmc_exclusive_scan_i32(dense, toff, (int)ncand, toff + ncand); if (rc) return rc;
            rc = dkmc_exclusive_scan_i32_i64(cnt + 2 * ncand, cov, (int)ncand, cov + ncand); if (rc) return rc;
            long long covered = 0;
            HIPCHK(hipMemcpyAsync(&ntiles, toff + ncand, sizeof(int), hipMemcpyDeviceToHost, st));
            HIPCHK(hipMemcpyAsync(&covered, cov + ncand, sizeof(long long), hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            // Tiles pay off when they take most of X out of the segment path; otherwise the leftovers fragment into short segments
            // and the plain segment path is faster: fall back to it.
            if ((double)covered < tile_min_cover * (double)nnz) {
                use_tiles = false; ntiles = 0;
                hipLaunchKernelGGL((k_build_runs<RP>), dim3((n_long + 3) / 4), dim3(256), 0, st, n_long, long_rows, rp, ci, srank, runs, nruns, rem, nrem, seg_len, 0);
            }
            e.stats.spmv_tile_entries = use_tiles ? covered / 2 : 0;
        }
        if (use_tiles) {
            const long long ncand = (long long)nK * nW;
            tiles = (TileDesc *)scratch(S_CG_TILES, (size_t)(ntiles + 1) * sizeof(TileDesc));
            tval = (double *)scratch(S_CG_TVAL, (size_t)(ntiles + 1) * TILE_R * TILE_C * 8);
            rowpart = (double *)scratch(S_CG_ROWPART, (size_t)(ncand + 1) * TILE_R * 8);      // one cell per grid position, zero where no tile
            colpart = (double *)scratch(S_CG_COLPART, (size_t)(ncand + 1) * TILE_C * 8);
            trange = (int *)scratch(S_CG_CSUM, (size_t)(2 * nK + nW + 8) * 4);
            if (!tiles || !tval || !rowpart || !colpart || !trange) return e.err_code;
            if (nK >= 512) {          // many row blocks per window: pre-sum the column partials (k_tile_colsum; at 262 row blocks the extra launch costs more than it saves)
                csum_pitch = (ns + 63) & ~63;
                csum = (double *)scratch(S_CG_CSUM2, (size_t)COLSUM_SLICES * csum_pitch * 8);
                if (!csum) return e.err_code;
            }
            HIPCHK(hipMemsetAsync(rowpart, 0, (size_t)ncand * TILE_R * 8, st));
            HIPCHK(hipMemsetAsync(colpart, 0, (size_t)ncand * TILE_C * 8, st));
            HIPCHK(hipMemsetAsync(tval, 0, (size_t)ntiles * TILE_R * TILE_C * 8, st));
            HIPCHK(hipMemsetAsync(pS + ns, 0, (size_t)TILE_C * 8, st));              // padding read by tiles of a partial last row block / window
            hipLaunchKernelGGL(k_tile_list, dim3((unsigned)((ncand + 255) / 256)), dim3(256), 0, st, nK, nW, (const int *)dense, (const int *)toff, tiles);
            hipLaunchKernelGGL(k_tile_ranges, dim3((std::max(nK, nW) + 255) / 256), dim3(256), 0, st, nK, nW, (const int *)dense, trange, trange + nK, trange + 2 * nK);
            // packed remainder entries: offsets now, values after the Jacobi scaling
            goff = (int *)scratch(S_CG_GOFF, (size_t)(n_long + 4) * 4);
            if (!goff) return e.err_code;
            rc = dkmc_exclusive_scan_i32(nrem, goff, n_long, goff + n_long); if (rc) return rc;
            hipLaunchKernelGGL((k_emit_segs<0, RP>), dim3((n_long + 255) / 256), dim3(256), 0, st, n_long, long_rows, rp, srank, (const RunDesc *)runs,
                               (const int *)nruns, ns, nW, (const int *)dense, TILE_SEG_LEN, nsegs, (const int *)nullptr, (RunDesc *)nullptr, (const int *)goff, (int *)nullptr);
            rc = dkmc_exclusive_scan_i32(nsegs, seg_off, n_long, seg_off + n_long); if (rc) return rc;
        } else {
            rc = dkmc_exclusive_scan_i32(nruns, seg_off, n_long, seg_off + n_long); if (rc) return rc;
            nsegs = nruns;
        }
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
            if (use_tiles) { tile_lo = (int)((long long)ntiles * me / nr); ntiles_loc = (int)((long long)ntiles * (me + 1) / nr) - tile_lo; }
            dparts = (RowParts *)scratch(S_CG_PARTS, sizeof(RowParts));
            if (!xbuf || !dparts) return e.err_code;
            HIPCHK(hipMemcpy(dparts, &parts, sizeof(RowParts), hipMemcpyHostToDevice));
            e.stats.comm_ranks = nr; e.stats.comm_local_segments = nseg_loc; e.stats.comm_count_per_rank = parts.chunk;
        } else { nseg_loc = nseg; e.stats.comm_ranks = 0; ntiles_loc = ntiles; }
        if (sharded && use_tiles) e.stats.comm_count_per_rank = n_long;
        e.stats.spmv_segments = nseg;
        segs = (RunDesc *)scratch(S_CG_SEGS, (size_t)(nseg + 1) * sizeof(RunDesc));
        seg_part = (double *)scratch(S_CG_SEGPART, (size_t)(nseg + 1) * 8);
        if (!segs || !seg_part) return e.err_code;
        if (use_tiles)
            hipLaunchKernelGGL((k_emit_segs<1, RP>), dim3((n_long + 255) / 256), dim3(256), 0, st, n_long, long_rows, rp, srank, (const RunDesc *)runs,
                               (const int *)nruns, ns, nW, (const int *)dense, TILE_SEG_LEN, nsegs, (const int *)seg_off, segs, (const int *)goff, chk);
        else
            hipLaunchKernelGGL((k_compact_segs<RP>), dim3((n_long + 3) / 4), dim3(256), 0, st, n_long, long_rows, rp, (const RunDesc *)runs,
                               (const int *)nruns, (const int *)seg_off, segs);
        if (use_tiles) {
            lmeta = (LRowMeta *)scratch(S_CG_LMETA, (size_t)(n_long + 1) * sizeof(LRowMeta));
            if (!lmeta) return e.err_code;
            hipLaunchKernelGGL(k_lrow_meta, dim3((n_long + 255) / 256), dim3(256), 0, st, n_long, long_rows, (const int *)lsr, (const int *)nsegs,
                               (const int *)seg_off, (const int *)trange, (const int *)(trange + nK), (const int *)(trange + 2 * nK),
                               sharded ? parts.lo[comm_rank()] : 0, sharded ? parts.lo[comm_rank() + 1] : n_long, lmeta);
        }
        e.stats.spmv_tiles = ntiles;
    }
    // blocks of k_spmv_segs: segments (a wave each; 16 lanes each in tile mode), tiles (a wave each), short rows (16 / 8 lanes each)
    const int nsb = use_tiles ? (nseg_loc + SEGK_NT / 16 - 1) / (SEGK_NT / 16) : (nseg_loc + SEGK_NT / 64 - 1) / (SEGK_NT / 64);
    const int ntb = (ntiles_loc + SEGK_NT / 64 - 1) / (SEGK_NT / 64);            // (this rank's share of the tiles)
    const int hsA = (use_runs && n_short > 0) ? grid_for(n_short, use_tiles ? SEGK_NT / 8 : SEGK_NT / 16) : 0;
    if (use_runs) np_ap = hsA + hl2;
    // ---- Jacobi scaling ----
    SPMV(M_DIAG, (const double *)nullptr, s, x, y, (double *)nullptr);
    SPMV(M_SCALE, (const double *)s, (double *)nullptr, (double *)nullptr, (double *)nullptr, (double *)nullptr);
    if (use_tiles) {
        // packed copies of the scaled remainder entries and short rows (k_pack_rem / k_pack_short)
        int h_tot[2] = {0, 0};
        int *slen = (int *)scratch(S_MISC0, (size_t)(n_short + 4) * 4);
        srp = (int *)scratch(S_CG_SRP, (size_t)(n_short + 4) * 4);
        if (!slen || !srp) return e.err_code;
        hipLaunchKernelGGL((k_short_len<RP>), dim3((n_short + 255) / 256), dim3(256), 0, st, n_short, short_rows, rp, slen);
        int rc = dkmc_exclusive_scan_i32(slen, srp, n_short, srp + n_short); if (rc) return rc;
        HIPCHK(hipMemcpyAsync(&h_tot[0], srp + n_short, sizeof(int), hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(&h_tot[1], goff + n_long, sizeof(int), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        sval = (double *)scratch(S_CG_SVAL, (size_t)(h_tot[0] + 2) * 8); scol = (int *)scratch(S_CG_SCOL, (size_t)(h_tot[0] + 2) * 4);
        gval = (double *)scratch(S_CG_GVAL, (size_t)(h_tot[1] + 2) * 8); gcol = (int *)scratch(S_CG_GCOL, (size_t)(h_tot[1] + 2) * 4);
        if (!sval || !scol || !gval || !gcol) return e.err_code;
        if (n_short > 0) hipLaunchKernelGGL((k_pack_short<RP>), dim3((n_short + 15) / 16), dim3(256), 0, st, n_short, short_rows, rp, ci, (const double *)a,
                                            (const int *)srp, sval, scol, srank, nW, (const int *)dense, (const int *)toff, tval, chk);
        hipLaunchKernelGGL((k_pack_rem<RP>), dim3((n_long + 3) / 4), dim3(256), 0, st, n_long, long_rows, rp, ci, (const double *)a, (const int *)rem,
                           (const int *)nrem, (const int *)goff, gval, gcol, srank, nW, (const int *)dense, (const int *)toff, tval, chk);
        if (ntiles > 0) {
            hipLaunchKernelGGL((k_tile_scatter<RP>), dim3((n_long + 3) / 4), dim3(256), 0, st, n_long, long_rows, rp, srank, (const RunDesc *)runs,
                               (const int *)nruns, nW, (const int *)dense, (const int *)toff, (const double *)a, tval, chk);
            // every upper entry placed in a tile must have had its mirror entry removed from the segment path, cell by cell
            const long long ncand = (long long)nK * nW;
            int *d_bad = (int *)scratch(S_MISC1, 16), h_bad = 0;
            if (!d_bad) return e.err_code;
            HIPCHK(hipMemsetAsync(d_bad, 0, 4, st));
            hipLaunchKernelGGL(k_chk_max, dim3((unsigned)((ncand + 255) / 256)), dim3(256), 0, st, ncand, (const int *)chk, d_bad);
            HIPCHK(hipMemcpyAsync(&h_bad, d_bad, 4, hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            if (h_bad) return dkmc_fail(49, "CG: symmetric tiles: X is not structurally symmetric inside a tile (dkmc_set_symmetric_tiles(0) avoids the tiles)", __FILE__, __LINE__);
        }
    }
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
    static const int spmv_var = getenv("DKMC_SPMV_VAR") ? atoi(getenv("DKMC_SPMV_VAR")) : 0;   // experiments only
    // matrix stream: default cache policy while the values of one sweep fit the 256 MiB Infinity Cache (they are re-read
    // every iteration), non-temporal beyond that (measured: 45 vs 53 us at 240 MB, 478 vs 456 us at 1.86 GB); a rank of a
    // sharded solve streams only its share
    const int seg_nt = (spmv_var == 3) ? 1 : (spmv_var == 2) ? 0 : (nnz * 8 / (sharded ? comm_nranks() : 1) > (300ll << 20));
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
                 (const double *)p, nsb, n_short, short_rows, rp, long_rows, t, part_pAp, ntb, ntiles_loc, nW, ns, (const TileDesc *)tiles + tile_lo, (const double *)tval + (size_t)tile_lo * TILE_R * TILE_C, rowpart, colpart, \
                 (const double *)gval, (const int *)gcol, (const int *)srp, (const double *)sval, (const int *)scol
                const dim3 sg(nsb + hsA + ntb);
                if (use_tiles) {       // tile role compiled in; short rows and remainder entries from their packed copies
                    if (seg_nt) hipExtLaunchKernelGGL((k_spmv_segs<1, RP, 1>), sg, dim3(SEGK_NT), 0, st, e0, e1, 0, SEG_ARGS);
                    else hipExtLaunchKernelGGL((k_spmv_segs<0, RP, 1>), sg, dim3(SEGK_NT), 0, st, e0, e1, 0, SEG_ARGS);
                }
                else if (seg_nt) hipExtLaunchKernelGGL((k_spmv_segs<1, RP, 0>), sg, dim3(SEGK_NT), 0, st, e0, e1, 0, SEG_ARGS);
                else hipExtLaunchKernelGGL((k_spmv_segs<0, RP, 0>), sg, dim3(SEGK_NT), 0, st, e0, e1, 0, SEG_ARGS);
#undef SEG_ARGS
                if (use_tiles) {
                    if (csum) hipLaunchKernelGGL(k_tile_colsum, dim3(nW * COLSUM_SLICES), dim3(TILE_C), 0, st, ns, nK, nW, (const int *)(trange + 2 * nK),
                                                 (const double *)colpart, csum, csum_pitch, (const CgCtrl *)ctrl);
                    if (sharded) {
                        // this rank's share of every long row's sum -> one all-reduce -> t and the p.t partials on every rank.  All ranks
                        // receive the same bits and enqueue exactly the same sequence of collectives.
                        hipLaunchKernelGGL(k_rowsum_tiles, dim3(hl2), dim3(SPMV_NT), 0, st, n_long, (const LRowMeta *)lmeta, (const double *)seg_part, nW,
                                           (const double *)rowpart, (const double *)colpart, (const double *)csum, csum_pitch, (const double *)p, t,
                                           part_pAp + hsA, (const CgCtrl *)ctrl, xbuf);
                        if (int rc = comm_allreduce_sum_f64(xbuf, (size_t)n_long)) return rc;
                        if (pb) HIPCHK(hipEventRecord(evc[b / PROF_STRIDE], st));
                        hipExtLaunchKernelGGL(k_xbuf_apply, dim3(hl2), dim3(SPMV_NT), 0, st, e2, e3, 0, n_long, (const LRowMeta *)lmeta, (const double *)xbuf,
                                              (const double *)p, t, part_pAp + hsA, (const CgCtrl *)ctrl);
                    } else
                    hipExtLaunchKernelGGL(k_rowsum_tiles, dim3(hl2), dim3(SPMV_NT), 0, st, e2, e3, 0, n_long, (const LRowMeta *)lmeta,
                                          (const double *)seg_part, nW, (const double *)rowpart, (const double *)colpart, (const double *)csum, csum_pitch,
                                          (const double *)p, t, part_pAp + hsA, (const CgCtrl *)ctrl, (double *)nullptr);
                } else
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
