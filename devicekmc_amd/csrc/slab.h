// slab.h -- partition of the rows of a sparse symmetric pattern into spatial slabs, shared by the slab-distributed block-CG on X (xtb_slab.inc)
// and the slab-distributed CG on K (kcg.hip); SURVEY 8(e): "row slabs; halo = sites within 3.5 A of a cut".  No counterpart in the reference.
// Rows [first, m) are distributed (X: first = 2, the two driver rows are replicated; K: first = 0); coord[row - first] is the lateral coordinate.
// Everything here works on replicated data with integer counts only: every rank computes the same partition.
#pragma once
#include "common.h"
#define XS_MAXR 32                                   // ranks (owner masks are 32 bit)
#define XS_BINS 4096                                 // histogram bins of the lateral coordinate

static __global__ void k_slab_hist(int m, int first, const double *__restrict__ coord, double lo, double hi, int *__restrict__ hist)
{
    const int row = first + blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= m) return;
    const double u = hi > lo ? (coord[row - first] - lo) / (hi - lo) : 0.0;
    int b = (int)(u * XS_BINS); b = b < 0 ? 0 : (b >= XS_BINS ? XS_BINS - 1 : b);
    atomicAdd(&hist[b], 1);                          // integer counts: the result does not depend on the order
}
// cuts[r] = first bin of slab r: the smallest bin with at least r / nr of the rows below it (one thread: 4096 bins)
static __global__ void k_slab_cuts(int nr, int nrows, const int *__restrict__ hist, int *__restrict__ cuts)
{
    if (blockIdx.x || threadIdx.x) return;
    int r = 1; long long cum = 0;
    cuts[0] = 0;
    for (int b = 0; b < XS_BINS && r < nr; ++b) {
        while (r < nr && cum * nr >= (long long)r * nrows) cuts[r++] = b;
        cum += hist[b];
    }
    for (; r <= nr; ++r) cuts[r] = XS_BINS;
}
static __global__ void k_slab_owner(int m, int first, const double *__restrict__ coord, double lo, double hi, int nr, const int *__restrict__ cuts, int *__restrict__ owner)
{
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= m) return;
    if (row < first) { owner[row] = 0; return; }     // (replicated rows: the entry is never used as an owner)
    const double u = hi > lo ? (coord[row - first] - lo) / (hi - lo) : 0.0;
    int b = (int)(u * XS_BINS); b = b < 0 ? 0 : (b >= XS_BINS ? XS_BINS - 1 : b);
    int o = 0;
    for (int r = 1; r < nr; ++r) o += cuts[r] <= b ? 1 : 0;
    owner[row] = o;
}
// bit d of mask[row]: row has an entry in a row owned by d != owner[row] (the pattern is symmetric: d reads this row's vector entry).
// Column words may carry a class bit 31 (K): masked off.
template <typename RP>
static __global__ void k_slab_mask(int m, int first, const RP *__restrict__ rp, const int *__restrict__ ci, const int *__restrict__ owner, unsigned *__restrict__ mask)
{
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= m) return;
    unsigned mk = 0;
    if (row >= first) {
        const int o = owner[row];
        for (RP p = rp[row]; p < rp[row + 1]; ++p) { const int c = ci[p] & 0x7fffffff; if (c >= first) { const int oc = owner[c]; if (oc != o) mk |= 1u << oc; } }
    }
    mask[row] = mk;
}
// tab: [nr] rows per owner | [nr] marked rows per owner (mark[row] >= 0; X: the rows of S) | [nr * nr] halo rows s -> d
static __global__ __launch_bounds__(256) void k_slab_count(int m, int first, int nr, const int *__restrict__ owner, const unsigned *__restrict__ mask, const int *__restrict__ mark, int *__restrict__ tab)
{
    __shared__ int lt[XS_MAXR * (XS_MAXR + 2)];
    const int nt = nr * (nr + 2);
    for (int i = threadIdx.x; i < nt; i += blockDim.x) lt[i] = 0;
    __syncthreads();
    for (int row = first + blockIdx.x * blockDim.x + threadIdx.x; row < m; row += gridDim.x * blockDim.x) {
        const int o = owner[row];
        atomicAdd(&lt[o], 1);
        if (mark && mark[row] >= 0) atomicAdd(&lt[nr + o], 1);
        unsigned mk = mask[row];
        while (mk) { const int d = __ffs((int)mk) - 1; mk &= mk - 1; atomicAdd(&lt[2 * nr + o * nr + d], 1); }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nt; i += blockDim.x) if (lt[i]) atomicAdd(&tab[i], lt[i]);
}
// list building: flag -> exclusive scan -> scatter (ascending order kept)
static __global__ void k_slab_flag_rows(int m, int first, const int *__restrict__ owner, int r, int *__restrict__ flag)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) flag[i] = (i >= first && owner[i] == r) ? 1 : 0;
}
static __global__ void k_slab_scatter_rows(int m, const int *__restrict__ flag, const int *__restrict__ pos, int base, int *__restrict__ rows_by_owner)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m && flag[i]) rows_by_owner[base + pos[i]] = i;
}
// over positions [j0, j1) of rows_by_owner: rows with bit `bit` of their mask set
static __global__ void k_slab_flag_halo(int j0, int j1, const int *__restrict__ rows_by_owner, const unsigned *__restrict__ mask, int bit, int *__restrict__ flag)
{
    const int j = j0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (j < j1) flag[j - j0] = (mask[rows_by_owner[j]] >> bit) & 1u;
}
static __global__ void k_slab_scatter_halo(int j0, int j1, const int *__restrict__ rows_by_owner, const int *__restrict__ flag, const int *__restrict__ pos, int base, int *__restrict__ out)
{
    const int j = j0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (j < j1 && flag[j - j0]) out[base + pos[j - j0]] = rows_by_owner[j];
}
