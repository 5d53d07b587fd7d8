// xshared.h -- device code shared by the two assemblies of the current-solve matrix X: the CSR form (current.hip, the
// reference's layout, kept for inspection and as the reference-order solver) and the tiled form (xt.hip, the default).
// Entry values follow populate_sparse_X_gpu2 (iterative_solvers_gpu.cu:1525-1721), the tunnelling predicate :887-912.
#pragma once
#include "common.h"

typedef long long xrp_t;      // row pointers of X are 64 bit: its tunnelling block outgrows 2^31 non-zeros beyond ~4e5 sites

enum { AF_V = 1, AF_MP_PAT = 2, AF_MP_VAL = 4, AF_METAL = 8, AF_CVAC = 16 };

struct XParams {
    int Na, nn, n_src, n_gnd, nlc, pbc;
    double tol, nn_dist, high_G, low_G, loop_G, m_e, V0, laty, latz;
};

// ---- site -> atom compaction (current_solver_gpu.cu:869-879: thrust::sequence + 7x copy_if) ------
static __global__ void k_atom_flags(int N, const int *__restrict__ element, int *flag)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) { const int e = element[i]; flag[i] = (e != DEFECT) && (e != OXYGEN_DEFECT); }
}

static __global__ void k_atom_gather(int N, const int *__restrict__ flag, const int *__restrict__ off,
                              const double *__restrict__ sx, const double *__restrict__ sy, const double *__restrict__ sz,
                              const int *__restrict__ sq, const int *__restrict__ sel, const double *__restrict__ scb,
                              double *ax, double *ay, double *az, int *aq, int *ael, double *acb, int *atom_site, int *site_atom)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    if (flag[i]) {
        const int a = off[i];
        ax[a] = sx[i]; ay[a] = sy[i]; az[a] = sz[i]; aq[a] = sq[i]; ael[a] = sel[i]; acb[a] = scb[i];
        atom_site[a] = i; site_atom[i] = a;
    } else site_atom[i] = -1;
}

// per-atom class flags + neighbour rows in atom numbering (ground atom removed), ascending, -1 padded
static __global__ void k_atom_rows(XParams P, const int *__restrict__ neigh, const int *__restrict__ atom_site, const int *__restrict__ site_atom,
                            const int *__restrict__ ael, const int *__restrict__ aq, MetalSet ms,
                            int *__restrict__ aflag, int *__restrict__ aneigh, int *__restrict__ ancnt, int *__restrict__ inS)
{
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= P.Na) return;
    const int el = ael[a];
    const bool metal = is_metal(el, ms);
    const int N_full = P.Na + 2;
    int f = 0;
    if (el == VACANCY) f |= AF_V;
    if (metal) f |= AF_METAL;
    if (el == VACANCY && aq[a] == 0) f |= AF_CVAC;
    // inner-contact window: bound = Natom in the pattern kernels (iterative_solvers_gpu.cu:894-900,1102-1108),
    // N_full in the value kernel (:1630-1636)
    if (metal && a > (P.nlc - 1) * P.n_src && a < P.Na - (P.nlc - 1) * P.n_gnd) f |= AF_MP_PAT;
    if (metal && a > (P.nlc - 1) * P.n_src && a < N_full - (P.nlc - 1) * P.n_gnd) f |= AF_MP_VAL;
    aflag[a] = f;
    inS[a] = (a < P.Na - 1) && (f & (AF_V | AF_MP_PAT)) ? 1 : 0;
    const int *row = neigh + (size_t)atom_site[a] * P.nn;
    int n = 0;
    for (int s = 0; s < P.nn; ++s) {
        const int j = row[s];
        if (j < 0) continue;
        const int b = site_atom[j];
        if (b >= 0 && b != P.Na - 1) aneigh[(size_t)a * P.nn + n++] = b;
    }
    ancnt[a] = n;
    for (; n < P.nn; ++n) aneigh[(size_t)a * P.nn + n] = -1;
}

struct __attribute__((aligned(16))) SEntry { double cb; int idx; int flag; };

static __global__ void k_S_scatter(int Na, const int *__restrict__ inS, const int *__restrict__ off, const int *__restrict__ aflag,
                            const double *__restrict__ acb, SEntry *S, int *srank)
{
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= Na) return;
    if (inS[a]) { SEntry e; e.cb = acb[a]; e.idx = a; e.flag = aflag[a]; S[off[a]] = e; srank[a] = off[a]; }
    else srank[a] = -1;
}

// tunnelling predicate (iterative_solvers_gpu.cu:887-912, 1095-1121, 1624-1646); MPBIT selects the window
template <int MPBIT>
__device__ __forceinline__ int tunnel_kind(int fa, int fb, double cba, double cbb, double tol)
{
    const bool v1 = fa & AF_V, v2 = fb & AF_V, m1 = fa & MPBIT, m2 = fb & MPBIT;
    const bool t2t = v1 && v2, c2t = (v1 && m2) || (v2 && m1), c2c = m1 && m2;
    if ((t2t || c2t || c2c) && (fabs(cba - cbb) > tol)) return c2t ? 1 : 2;
    return 0;
}

// ---- pattern: rows outside S (thread per row) ----------------------------------------------------
// MODE 0: count, MODE 1: fill
template <int MODE>
__global__ void k_xpat_plain(XParams P, const int *__restrict__ inS, const int *__restrict__ aneigh, const int *__restrict__ ancnt,
                             int *__restrict__ cnt, const xrp_t *__restrict__ rp, int *__restrict__ col)
{
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    const int Nsub = P.Na + 1, N_full = P.Na + 2;
    if (row >= Nsub) return;
    if (row == 0) {                                                // :1035-1045
        int n = 0;
        if (MODE == 0) { cnt[0] = 2 + max(0, (N_full - 2) - max(N_full - P.n_gnd, 1)); return; }
        col[rp[0] + n++] = 0; col[rp[0] + n++] = 1;
        for (int j = max(N_full - P.n_gnd + 1, 2); j < N_full - 1; ++j) col[rp[0] + n++] = j;
        return;
    }
    if (row == 1) {                                                // :1047-1054
        if (MODE == 0) { cnt[1] = P.n_src + 2; return; }
        for (int j = 0; j < P.n_src + 2; ++j) col[rp[1] + j] = j;
        return;
    }
    const int a = row - 2;
    if (inS && inS[a]) return;                                     // handled by k_xpat_S (inS == nullptr: neighbour pattern of every row, xt.hip)
    const int pre0 = row > N_full - P.n_gnd, pre1 = row < P.n_src + 2;
    const int nnb = ancnt[a];
    if (MODE == 0) { cnt[row] = pre0 + pre1 + nnb + 1; return; }
    xrp_t p = rp[row];
    if (pre0) col[p++] = 0;
    if (pre1) col[p++] = 1;
    bool self_done = false;
    for (int s = 0; s < nnb; ++s) {
        const int b = aneigh[(size_t)a * P.nn + s];
        if (!self_done && b > a) { col[p++] = a + 2; self_done = true; }
        col[p++] = b + 2;
    }
    if (!self_done) col[p++] = a + 2;
}

// ---- pattern: rows of S (one workgroup per row) ----------------------------------------------------
#define XS_NT 256
template <int MODE>
__global__ __launch_bounds__(XS_NT) void k_xpat_S(XParams P, int ns, const SEntry *__restrict__ S, const int *__restrict__ aneigh,
                                                  const int *__restrict__ ancnt, int *__restrict__ cnt, const xrp_t *__restrict__ rp,
                                                  int *__restrict__ col)
{
    __shared__ int nb[72], hist[72], wtot[XS_NT / 64];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const SEntry me = S[blockIdx.x];
    const int a = me.idx, row = a + 2, N_full = P.Na + 2;
    const int nnb = ancnt[a], nN = nnb + 1;
    if (tid == 0) {                                                // N' = neighbours U {a}, ascending
        int n = 0; bool self_done = false;
        for (int s = 0; s < nnb; ++s) { const int b = aneigh[(size_t)a * P.nn + s]; if (!self_done && b > a) { nb[n++] = a; self_done = true; } nb[n++] = b; }
        if (!self_done) nb[n++] = a;
    }
    if (tid < 72) hist[tid] = 0;
    __syncthreads();
    const int pre = (row > N_full - P.n_gnd) + (row < P.n_src + 2);
    const xrp_t row_start = (MODE == 1) ? rp[row] : 0;
    int total_before = 0;
    for (int base = 0; base < ns; base += XS_NT) {
        const int k = base + tid;
        bool match = false; int lt = 0; int b = -1;
        if (k < ns) {
            const SEntry o = S[k];
            b = o.idx;
            int lo = 0, hi = nN;                                   // lower_bound in N'
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (nb[mid] < b) lo = mid + 1; else hi = mid; }
            lt = lo;
            const bool is_nb = (lt < nN) && (nb[lt] == b);        // neighbours and self are "direct"/diagonal terms
            match = !is_nb && tunnel_kind<AF_MP_PAT>(me.flag, o.flag, me.cb, o.cb, P.tol) != 0;
        }
        const unsigned long long bal = __ballot(match);
        const int wexcl = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wtot[w] = __popcll(bal);
        __syncthreads();
        int wbase = 0, ctot = 0;
#pragma unroll
        for (int q = 0; q < XS_NT / 64; ++q) { if (q < w) wbase += wtot[q]; ctot += wtot[q]; }
        if (MODE == 1 && match) {
            col[row_start + pre + total_before + wbase + wexcl + lt] = b + 2;
            atomicAdd(&hist[lt], 1);
        }
        total_before += ctot;
        __syncthreads();
    }
    if (MODE == 0) { if (tid == 0) cnt[row] = pre + nN + total_before; return; }
    if (tid == 0) { xrp_t p = row_start; if (row > N_full - P.n_gnd) col[p++] = 0; if (row < P.n_src + 2) col[p++] = 1; }
    if (tid < nN) {
        int cum = 0;
        for (int t = 0; t <= tid; ++t) cum += hist[t];
        col[row_start + pre + tid + cum] = nb[tid] + 2;
    }
}

// ---- cache of contact->trap tunnelling coefficients ------------------------------------------------------------------
// The contact->trap entries integrate up to 500 energy levels each (iterative_solvers_gpu.cu:1652-1676) and are 95 % of the
// value-assembly time, yet T(vacancy site, contact atom) depends only on the two positions and the two CB edges, which are
// fixed for a bias point: per KMC step only the handful of sites that BECAME vacancies need new integrals.  One cache row
// per vacancy site (slot_of_site), one column per inner-contact metal (mrank of the atom).  Rows are filled by the same
// wkb_T() the direct path uses, so cached and recomputed values are bit-identical.  The cache validates itself every step
// against the current CB edges (metal snapshot + per-row vacancy CB) and is dropped on any mismatch.
struct TCacheView {
    int enabled, nM, cap;
    const int *slot_of_site;      // [N]   cache row of a site, -1 if none
    const int *mrank_atom;        // [Na]  cache column of an atom, -1 if it is not an inner-contact metal
    const double *vals;           // [cap][ncols]: columns col_lo ... col_lo + ncols - 1 of every cached vacancy (one GPU: all nM columns)
    int col_lo, ncols;
    // Sharded solve on the tiled X only (nL = 0 otherwise): a rank caches what ITS tiles read.  S is in atom order -- left-contact metals,
    // vacancies, right-contact metals -- and a tile belongs to the rank that owns its COLUMN window, so a rank needs (a) the nL left-contact
    // columns for the vacancies of its own windows (part A: slotA_of_site, valsA[capA][nL]) and (b) for ALL vacancies the right-contact
    // columns of its own windows (the part above, col_lo >= nL).  Anything a tile reads outside these is integrated directly: same
    // function, same bits (tc_lookup misses), so a window that moved since the parts were sized costs time, never correctness.
    int nL;
    const int *slotA_of_site;
    const double *valsA;
};
__device__ __forceinline__ bool tc_lookup(const TCacheView &TC, int slot, int slotA, int mr, double &val)
{
    if (mr < 0) return false;
    if (mr < TC.nL) { if (slotA < 0) return false; val = TC.valsA[(size_t)slotA * TC.nL + mr]; return true; }
    const int c = mr - TC.col_lo;
    if (slot < 0 || c < 0 || c >= TC.ncols) return false;
    val = TC.vals[(size_t)slot * TC.ncols + c];
    return true;
}

// ---- values (populate_sparse_X_gpu2 :1525-1721 + calc_diagonal_X_gpu :2053-2076) ---------------------
__device__ __forceinline__ double pow15(double e) { return e * sqrt(e); }

__device__ __forceinline__ double wkb_T(int kind, double dist, double drop, double prefac, double V0)
{
    if (kind == 1) {                                               // contact -> trap: integrate over the occupied levels
        const double dE = DKMC_Q * 0.01;
        double T = 0.0;
        const double c = prefac * (dist / drop);
        for (double iv = 0; iv < drop; iv += dE) {
            const double E1 = DKMC_Q * V0 + iv, E2 = E1 - drop;
            if (E2 > 0) T += exp(c * (pow15(E1) - pow15(E2)));
            if (E2 < 0) T += exp(c * pow15(E1));
        }
        return T;
    }
    const double E1 = DKMC_Q * V0, E2 = E1 - drop;
    const double c = prefac * (dist / fabs(E1 - E2));
    if (E2 > 0) return exp(c * (pow15(E1) - pow15(E2)));
    if (E2 < 0) return exp(c * pow15(E1));
    return 0.0;
}

// value of entry (row i >= 2, column c); returns the value, flags the diagonal
__device__ __forceinline__ double x_entry(const XParams &P, int i, int c, const double *__restrict__ ax, const double *__restrict__ ay,
                                          const double *__restrict__ az, const int *__restrict__ aflag, const double *__restrict__ acb,
                                          double xa, double ya, double za, int fa, double cba, double prefac,
                                          const TCacheView &TC, const int *__restrict__ atom_site)
{
    const int N_full = P.Na + 2;
    if (c == 0) return (i > N_full - P.n_gnd) ? -P.high_G : 0.0;  // :1602-1605
    if (c == 1) return (i < P.n_src + 2) ? -P.high_G : 0.0;       // :1608-1611
    if (c == i) {                                                  // :1588-1599 (ground = last atom)
        const int g = P.Na - 1;
        const double d = site_dist(xa, ya, za, ax[g], ay[g], az[g], P.laty, P.latz, P.pbc);
        return d < P.nn_dist ? P.high_G : 0.0;
    }
    const int b = c - 2;
    const double dA = site_dist(xa, ya, za, ax[b], ay[b], az[b], P.laty, P.latz, P.pbc);
    const int fb = aflag[b];
    if (dA < P.nn_dist) {                                          // direct terms :1698-1716
        const bool mm = (fa & AF_METAL) && (fb & AF_METAL), cc = (fa & AF_CVAC) && (fb & AF_CVAC);
        return (mm || cc) ? -P.high_G : -P.low_G;
    }
    const double cbb = acb[b];
    const int kind = tunnel_kind<AF_MP_VAL>(fa, fb, cba, cbb, P.tol);
    if (!kind) return 0.0;
    if (kind == 1 && TC.enabled) {
        const int va = (fa & AF_V) ? i - 2 : b, ma = (fa & AF_V) ? b : i - 2;       // vacancy atom, metal atom
        const int site = atom_site[va];
        double cv;
        if (tc_lookup(TC, TC.slot_of_site[site], TC.nL ? TC.slotA_of_site[site] : -1, TC.mrank_atom[ma], cv)) return -cv;
    }
    return -wkb_T(kind, 1e-10 * dA, fabs(cba - cbb), prefac, P.V0);
}

// LPR lanes per row over a row list (rows == nullptr: rows 0..nrows-1 are node rows 0..)
template <int LPR>
__global__ __launch_bounds__(256) void k_xval(XParams P, int nrows, const SEntry *__restrict__ S, int use_S,
                                              const int *__restrict__ inS, const xrp_t *__restrict__ rp, const int *__restrict__ ci,
                                              const double *__restrict__ ax, const double *__restrict__ ay, const double *__restrict__ az,
                                              const int *__restrict__ aflag, const double *__restrict__ acb, double *__restrict__ data,
                                              TCacheView TC, const int *__restrict__ atom_site, xrp_t *__restrict__ diag_pos = nullptr)
{
    const int gpb = 256 / LPR, g = threadIdx.x / LPR, l = threadIdx.x % LPR;
    const int ridx = blockIdx.x * gpb + g;
    if (ridx >= nrows) return;
    int i;
    if (use_S) i = S[ridx].idx + 2;
    else { i = ridx; if (i >= 2 && inS && inS[i - 2]) return; }
    const int N_full = P.Na + 2;
    const xrp_t p0 = rp[i], p1 = rp[i + 1];
    const double prefac = -(sqrt(2 * P.m_e) / DKMC_HBAR) * (2.0 / 3.0);
    double off = 0.0, dval = 0.0; xrp_t dpos = -1;
    if (i == 0) {                                                  // :1550-1567
        for (xrp_t p = p0 + l; p < p1; p += LPR) {
            const int c = ci[p]; double v = 0.0;
            if (c == 0) v = +P.high_G;
            if (c == 1) v = -P.loop_G;
            if (c > N_full - P.n_gnd) v = -P.high_G;
            if (c == 0) { dpos = p; dval = v; } else { data[p] = v; off += v; }
        }
    } else if (i == 1) {                                           // :1570-1582
        for (xrp_t p = p0 + l; p < p1; p += LPR) {
            const int c = ci[p]; double v = 0.0;
            if (c == 0) v = -P.loop_G;
            if (c >= 2 || (c > N_full - P.n_gnd)) v = -P.high_G;
            if (c == 1) { dpos = p; dval = v; } else { data[p] = v; off += v; }
        }
    } else {
        const int a = i - 2;
        const double xa = ax[a], ya = ay[a], za = az[a], cba = acb[a];
        const int fa = aflag[a];
        for (xrp_t p = p0 + l; p < p1; p += LPR) {
            const int c = ci[p];
            const double v = x_entry(P, i, c, ax, ay, az, aflag, acb, xa, ya, za, fa, cba, prefac, TC, atom_site);
            if (c == i) { dpos = p; dval = v; } else { data[p] = v; off += v; }
        }
    }
    off = group_sum<LPR>(off);
    if (dpos >= 0) { data[dpos] = dval + -off; if (diag_pos) diag_pos[i] = dpos; }      // calc_diagonal_X_gpu
}

