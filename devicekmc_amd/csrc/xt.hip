// xt.hip -- the current solve on the TILED form of X (default; dkmc_set_x_format(1)).
//
// Replaces, for the solve itself, Assemble_X_sparsity / Assemble_X2 (iterative_solvers_gpu.cu:1909-1983, 2113-2156) and
// solve_sparse_CG_Jacobi (:309-480) as update_power_gpu_sparse uses them (current_solver_gpu.cu:854-1147).  The reference
// (and the CSR path of current.hip, kept for inspection) stores every entry of X in CSR.  X has two very different parts:
//   * the neighbour part Xs: drivers, diagonal, <= nn direct couplings per atom -- O(N_atom) entries, a small CSR;
//   * the tunnelling block T over the set S = {vacancies} U {inner-contact metals}: every non-neighbour pair (a, b) of S
//     whose conduction-band edges differ by more than tol carries -T(a, b) (WKB, :1649-1693).  |S|^2-ish entries (3.7e9 at
//     9.4e5 sites), symmetric bit for bit, dense by classes (contact x contact, vacancy x contact).
// Here T is never written as CSR.  Its UPPER triangle is generated straight into tile-major storage: the S x S index space is
// cut into cells of 32 S-rows x 256 S-columns, a cell into 8 sub-blocks of 32 x 32; a census (predicate only) finds the
// non-empty sub-blocks, the fill kernel writes their values (zero where no entry, lower triangle of a diagonal sub-block
// zero).  8 KiB per stored sub-block, no column indices, no mirror entries, no second (scaled) copy: the Jacobi scaling is
// applied to the vectors (A p = S X (S p)), so the same unscaled tiles serve the solve, I_macro and the dissipated power.
// One wave streams a tile once per CG iteration and forms both the row products (t_i += x_ij q_j) and the column products
// (t_j += x_ij q_i): 4 B of HBM traffic per entry of X.  Work items are runs of up to KC tiles of one 256-column strip:
// column sums stay in registers across the run, row sums (32 per tile) go to a grid-indexed array; a small second kernel
// adds, per S-row, its row partials, its column partials and the sparse part.
// Where the neighbour part runs depends on the size of the sweep: inside the tile launch, tile and neighbour workgroups alternating
// (the sweep fits the Infinity Cache: the launch is ramp-bound and the neighbour rows hide beside the stream); inside it, behind
// the tile workgroups (up to ~2 GB); as its own full-occupancy kernel behind the tile pass (multi-GB sweeps).  See k_xt_apply.
// Multi-GPU (comm.hip): work items are dealt to the ranks in contiguous, byte-balanced shares (runs shrink towards the end of a
// share); a rank generates, stores and streams only its tiles; one all-reduce of |S| + 1 doubles per matrix-vector product
// completes the rows, with the neighbour part on a second stream beside it; results are re-published from rank 0.
#include "xtiles.h"
#include <hip/hip_ext.h>
int tc_prepare_tiled(const XParams &P, dkmc_gpubuf *buf, int ns, const SEntry *S, const int *aflag, const int *srank, const int *atom_site,
                     int c_lo, int c_hi, int sharded, TCacheView *out);       // current.hip
#include <vector>
#include <algorithm>
#include <stdlib.h>

// value of the tunnelling entry between two members of S (0 = no entry); symmetric in its two atoms bit for bit
__device__ __forceinline__ double xt_tvalue(const XParams &P, double prefac, const TCacheView &TC,
                                            double x1, double y1, double z1, double cb1, int f1, int slot1, int slotA1, int mr1,
                                            double x2, double y2, double z2, double cb2, int f2, int slot2, int slotA2, int mr2)
{
    const double d = site_dist(x1, y1, z1, x2, y2, z2, P.laty, P.latz, P.pbc);
    if (d < P.nn_dist) return 0.0;                                   // neighbour pair: direct term, part of Xs
    const int kind = tunnel_kind<AF_MP_VAL>(f1, f2, cb1, cb2, P.tol);
    if (!kind) return 0.0;
    if (kind == 1 && TC.enabled) {
        const bool v1 = f1 & AF_V;
        double cv;
        if (tc_lookup(TC, v1 ? slot1 : slot2, v1 ? slotA1 : slotA2, v1 ? mr2 : mr1, cv)) return -cv;
    }
    return -wkb_T(kind, 1e-10 * d, fabs(cb1 - cb2), prefac, P.V0);
}

// ---- S in solver order ---------------------------------------------------------------------------------------------------
__global__ void k_xt_snodes(int ns, int ns_pad, const SEntry *__restrict__ S, const double *__restrict__ ax, const double *__restrict__ ay,
                            const double *__restrict__ az, double *sx, double *sy, double *sz, double *scb, int *sflag, int *srow)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= ns_pad) return;
    if (s < ns) {
        const SEntry e = S[s];
        sx[s] = ax[e.idx]; sy[s] = ay[e.idx]; sz[s] = az[e.idx]; scb[s] = e.cb; sflag[s] = e.flag; srow[s] = e.idx + 2;
    } else { sx[s] = 0; sy[s] = 0; sz[s] = 0; scb[s] = 0; sflag[s] = 0; srow[s] = -1; }
}
// rows / columns of the coefficient cache of every member of S (after the cache has been brought up to date for this rank's share)
__global__ void k_xt_snode_slots(int ns, int ns_pad, const SEntry *__restrict__ S, const int *__restrict__ atom_site, TCacheView TC,
                                 int *sslot, int *sslotA, int *smr)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= ns_pad) return;
    int sl = -1, sa = -1, mr = -1;
    if (s < ns && TC.enabled) {
        const SEntry e = S[s];
        if (e.flag & AF_V) { const int site = atom_site[e.idx]; sl = TC.slot_of_site[site]; if (TC.nL) sa = TC.slotA_of_site[site]; }
        mr = TC.mrank_atom[e.idx];
    }
    sslot[s] = sl; sslotA[s] = sa; smr[s] = mr;
}

// ---- census: which 32 x 32 sub-blocks of the upper triangle hold an entry -----------------------------------------------------
// One wave per cell, cells in strip-major order (ci = w * nK + k).  Predicate only (no WKB value): class flags, |dE| > tol,
// not a neighbour pair, column rank above row rank.  A full cell is recognised after its first row.
__global__ __launch_bounds__(XT_NT) void k_xt_census(XParams P, int ns, int nK, int nW, SNodes S, unsigned *__restrict__ cmask)
{
    const int lane = threadIdx.x & 63;
    const long long ci = (long long)blockIdx.x * (XT_NT / 64) + (threadIdx.x >> 6);
    if (ci >= (long long)nK * nW) return;
    const int w = (int)(ci / nK), k = (int)(ci % nK);
    unsigned m = 0;
    if (XT_C * w + XT_C - 1 > XT_R * k) {                            // the cell reaches above the diagonal
        double cx[4], cy[4], cz[4], ccb[4]; int cf[4], csc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int sc = XT_C * w + lane + 64 * j;                 // lanes 0-31: sub-block 2j, lanes 32-63: sub-block 2j+1
            csc[j] = sc; cx[j] = S.x[sc]; cy[j] = S.y[sc]; cz[j] = S.z[sc]; ccb[j] = S.cb[sc]; cf[j] = S.flag[sc];
        }
        for (int r = 0; r < XT_R; ++r) {
            const int s = XT_R * k + r;
            if (s >= ns) break;
            const double rx = S.x[s], ry = S.y[s], rz = S.z[s], rcb = S.cb[s]; const int rf = S.flag[s];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool pred = csc[j] > s && tunnel_kind<AF_MP_VAL>(rf, cf[j], rcb, ccb[j], P.tol) != 0 &&
                                  !(site_dist(rx, ry, rz, cx[j], cy[j], cz[j], P.laty, P.latz, P.pbc) < P.nn_dist);
                const unsigned long long bal = __ballot(pred);
                if (bal & 0xffffffffull) m |= 1u << (2 * j);
                if (bal >> 32) m |= 1u << (2 * j + 1);
            }
            if (m == 0xffu) break;
        }
    }
    if (lane == 0) cmask[ci] = m;
}
__global__ void k_xt_cell_counts(long long ncell, const unsigned *__restrict__ cmask, int *__restrict__ is_tile, int *__restrict__ nsub)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < ncell) { const unsigned m = cmask[i]; is_tile[i] = m != 0; nsub[i] = __popc(m); }
}
__global__ void k_xt_tile_list(int nK, long long ncell, const unsigned *__restrict__ cmask, const int *__restrict__ toff, const int *__restrict__ soff,
                               XTile *__restrict__ tiles)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < ncell && cmask[i]) { XTile t; t.k = (int)(i % nK); t.w = (int)(i / nK); t.mask = cmask[i]; t.soff = soff[i]; tiles[toff[i]] = t; }
}
// per row block: first / one-past-last window holding a tile
__global__ void k_xt_wrange(int nK, int nW, const unsigned *__restrict__ cmask, int2 *__restrict__ wrange)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nK) {
        int b = nW, e = 0;
        for (int w = 0; w < nW; ++w) if (cmask[(long long)w * nK + i]) { b = min(b, w); e = w + 1; }
        wrange[i] = make_int2(b, e);
    }
}
// Work items = runs of tiles of one strip.  The tile list (strip-major) is cut into the ranks' shares at tile boundaries, balanced by
// stored sub-blocks; inside a share the run length is kc except towards the END of the share, where it halves every XT_TAPER tiles
// down to single tiles: the last waves of a launch then stream 64 KiB each instead of a whole kc-tile run alone and latency-bound
// (measured at a 1/8 share of the 9.4e5-site stack: 426 us per launch untapered against 357 us at the full-size rate).
#define XT_MAXRANKS 64
#define XT_TAPER 2048
struct XSplit { int n; int taper; int tb[XT_MAXRANKS + 1]; int item_lo[XT_MAXRANKS + 1]; int max_items_per_strip; int nitems;
                int soff[XT_MAXRANKS + 1], w_first[XT_MAXRANKS + 1], w_last[XT_MAXRANKS + 1]; };   // per share boundary r: sub-block offset and window of tile tb[r]; window of tile tb[r] - 1
__global__ void k_xt_split(int ntiles, long long nsub_total, const XTile *__restrict__ tiles, int n, XSplit *sp)
{
    const int r = threadIdx.x;
    if (r > n) return;
    int t = 0;
    if (r == n) t = ntiles;
    else if (r > 0) {
        const long long want = nsub_total * r / n;
        int lo = 0, hi = ntiles;                                       // first tile whose first sub-block is at or beyond `want`
        while (lo < hi) { const int mid = (lo + hi) >> 1; if ((long long)tiles[mid].soff < want) lo = mid + 1; else hi = mid; }
        t = lo;
    }
    sp->tb[r] = t;
    // (runs shrink towards the end of a share only where a share is long enough for that to matter: small systems keep their run length)
    if (r == 0) { sp->n = n; sp->max_items_per_strip = 0; sp->taper = (ntiles / n >= 8 * XT_TAPER) ? XT_TAPER : 0; }
}
__device__ __forceinline__ int xt_run_len(int t, int t1, int kc, const XSplit *sp)
{
    int r = 0;
    while (r + 1 < sp->n && t >= sp->tb[r + 1]) ++r;
    const int end = sp->tb[r + 1], d = end - 1 - t;
    const int sh = sp->taper ? d / sp->taper : 5;
    int len = sh >= 5 ? kc : min(kc, 1 << sh);
    return max(1, min(len, min(end, t1) - t));
}
// MODE 0: items per strip (+ the largest count); MODE 1: write them at ioff[w]
// rec_shift = 2 (one GPU): the run count of every strip is padded to a multiple of four with empty runs, so that the four waves of a
// workgroup always hold runs of ONE strip and can leave one combined record of column sums (a quarter of the records the row kernel folds)
template <int MODE>
__global__ void k_xt_items(int nK, int nW, int kc, int ntiles, const int *__restrict__ toff, const int *__restrict__ ioff, const XTile *__restrict__ tiles,
                           XSplit *sp, int *__restrict__ nitem_w, XItem *__restrict__ items, int rec_shift)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w == 0 && !MODE) nitem_w[2 * (nW + 2)] = rec_shift;             // read by the row kernels (xt_row_block_runs)
    if (w >= nW) return;
    const int t0 = toff[(long long)w * nK], t1 = (w + 1 < nW) ? toff[(long long)(w + 1) * nK] : ntiles;
    int o = MODE ? ioff[w] : 0, c = 0;
    const int m = (1 << rec_shift) - 1;
    for (int t = t0; t < t1;) {
        const int len = xt_run_len(t, t1, kc, sp);
        if (MODE) { const XTile f = tiles[t]; XItem it; it.t0 = t; it.t1 = t + len; it.w = w; it.c = rec_shift ? 1 : 0; it.k0 = f.k; it.mask0 = f.mask; it.soff0 = f.soff; it.pad = (o + c) >> rec_shift; items[o + c] = it; }
        t += len; ++c;
        // the group of four is completed with empty runs at the end of the strip and where a rank's share ends inside it: the empty runs
        // carry the index of the LAST tile before them (t0 = t1: no tile), so that they sort into the share they complete
        bool cut = t >= t1;
        if (!cut) for (int r = 1; r < sp->n; ++r) cut |= sp->tb[r] == t;
        if (cut) {
            const int cpad = (c + m) & ~m;
            for (; c < cpad; ++c)
                if (MODE) { XItem it; it.t0 = t - 1; it.t1 = t - 1; it.w = w; it.c = 1; it.k0 = 0; it.mask0 = 0; it.soff0 = 0; it.pad = (o + c) >> rec_shift; items[o + c] = it; }
        }
    }
    if (!MODE) { nitem_w[w] = c; atomicMax(&sp->max_items_per_strip, c); }
}
// first item of every rank's share (items are in tile order and never cross a share boundary) + what the host needs of the boundary tiles
__global__ void k_xt_rank_items(const int *__restrict__ nitems_dev, int ntiles, long long nsub_total, const XItem *__restrict__ items,
                                const XTile *__restrict__ tiles, XSplit *sp)
{
    const int r = threadIdx.x;
    const int nitems = *nitems_dev;
    if (r == 0) sp->nitems = nitems;
    if (r > sp->n) return;
    const int tb = sp->tb[r];
    int lo = 0, hi = nitems;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (items[mid].t0 < tb) lo = mid + 1; else hi = mid; }
    if (r == sp->n) lo = nitems;                                      // the empty runs that pad the last strip (t0 = ntiles) belong to the last share
    sp->item_lo[r] = lo;
    sp->soff[r] = tb < ntiles ? tiles[tb].soff : (int)nsub_total;
    sp->w_first[r] = tb < ntiles ? tiles[tb].w : 0;
    sp->w_last[r] = tb > 0 ? tiles[tb - 1].w : 0;
}

// ---- fill: values of the stored sub-blocks --------------------------------------------------------------------------------
// One workgroup per tile; thread (q = tid / 32, c = tid % 32) owns column 32 q + c of the tile and walks its 32 rows.
// Element (row r, column 32 q + c) of sub-block slot sl sits at ((sl * 32 + r) * 32 + c): a sub-block is 8 consecutive
// 1-KiB wave loads of the apply kernel.
__global__ __launch_bounds__(XT_NT) void k_xt_fill(XParams P, int ns, const XTile *__restrict__ tiles, int sub_base, SNodes S, TCacheView TC,
                                                   double *__restrict__ tval, unsigned long long *__restrict__ nnz_upper)
{
    __shared__ double rx[XT_R], ry[XT_R], rz[XT_R], rcb[XT_R];
    __shared__ int rf[XT_R], rslot[XT_R], rslotA[XT_R], rmr[XT_R];
    const XTile td = tiles[blockIdx.x];
    const int tid = threadIdx.x;
    if (tid < XT_R) {
        const int s = XT_R * td.k + tid;           // < ns_pad: the arrays are padded
        rx[tid] = S.x[s]; ry[tid] = S.y[s]; rz[tid] = S.z[s]; rcb[tid] = S.cb[s]; rf[tid] = S.flag[s]; rslot[tid] = S.slot[s]; rslotA[tid] = S.slotA[s]; rmr[tid] = S.mr[s];
    }
    __syncthreads();
    const int q = tid >> 5, c = tid & 31;
    int cnt = 0;
    if ((td.mask >> q) & 1u) {
        const int sl = __popc(td.mask & ((1u << q) - 1u));
        double *dst = tval + ((size_t)(td.soff - sub_base) + sl) * XT_SUB + c;
        const int sc = XT_C * td.w + XT_SBW * q + c;
        const double cx = S.x[sc], cy = S.y[sc], cz = S.z[sc], ccb = S.cb[sc];
        const int cf = S.flag[sc], cslot = S.slot[sc], cslotA = S.slotA[sc], cmr = S.mr[sc];
        const double prefac = -(sqrt(2 * P.m_e) / DKMC_HBAR) * (2.0 / 3.0);
        for (int r = 0; r < XT_R; ++r) {
            const int s = XT_R * td.k + r;
            double v = 0.0;
            if (sc > s && cf && rf[r])
                v = xt_tvalue(P, prefac, TC, rx[r], ry[r], rz[r], rcb[r], rf[r], rslot[r], rslotA[r], rmr[r], cx, cy, cz, ccb, cf, cslot, cslotA, cmr);
            dst[r * XT_SBW] = v;
            cnt += v != 0.0;
        }
    }
    cnt = wave_sum_all_i(cnt);
    if ((tid & 63) == 0 && cnt) atomicAdd(nnz_upper, (unsigned long long)cnt);
}

// ---- apply: one pass over the tiles (+ the sparse rows in the same launch) -----------------------------------------------
// OP 0: matrix-vector product; OP 1: dissipated power (host formula current_solver.cpp:288-357 on the pairs of T).
template <int OP>
__device__ __forceinline__ void xt_acc(const dbl2 v, const double pcx, const double pcy, const double pr, double &ra, double &cax, double &cay, bool vpos)
{
    if (OP == 0) { ra += v.x * pcx + v.y * pcy; cax += v.x * pr; cay += v.y * pr; }
    else {
        // entry x between row node i (potential pr) and column node j (pcx / pcy): ical = x (m_i - m_j); the pair heats the
        // node the current flows INTO: row i when ical has the sign opposite to Vd, column j otherwise; amount x (m_i - m_j)^2
        const double dx = pr - pcx, dy = pr - pcy;
        const double ix = v.x * dx, iy = v.y * dy;
        const double ex = ix * dx, ey = iy * dy;
        const bool rx = vpos ? (ix < 0) : (ix > 0), ry = vpos ? (iy < 0) : (iy > 0);
        ra += (rx ? ex : 0.0) + (ry ? ey : 0.0);
        cax += rx ? 0.0 : ex; cay += ry ? 0.0 : ey;
    }
}

// One work item.  Lane (rr = lane / 16, cc = lane % 16) owns rows 4 j + rr (j = 0..7) and the column pair 2 cc, 2 cc + 1 of every
// sub-block.  Per sub-block: 8 loads of 1 KiB per wave (two sub-blocks in flight), row sums accumulate in registers over the
// tile, the two column sums are combined over rr and added to the wave's 256 column accumulators in LDS (lcol).
template <int OP, int NTL>
__device__ __forceinline__ void xt_tile_role(const XItem it, const XTile *__restrict__ tiles, int sub_base, const double *__restrict__ tval,
                                             const double *__restrict__ vS, int nW, int ns_pad, double *__restrict__ rowpart,
                                             double *__restrict__ colpart, bool vpos, double *lcol, double *lq)
{
    // it.c != 0: the four waves of this workgroup hold runs of one strip and write ONE record (lcol = this wave's 512 doubles of
    // lcol_all[4][512]; the caller guarantees that all four waves get here: the item count is a multiple of four)
    const int lane = threadIdx.x & 63, cc = lane & 15, rr = lane >> 4;
    XTile td; td.k = it.k0; td.w = it.w; td.mask = it.mask0; td.soff = it.soff0;
#define XT_LD(dst, slot) _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) dst[j_] = NTL ? __builtin_nontemporal_load(base + (size_t)(8 * (slot) + j_) * 64) : base[(size_t)(8 * (slot) + j_) * 64];
    // the first 8 KiB of the first tile are requested before anything else: the set-up below (LDS, vector entries) runs behind them
    dbl2 va[8];
    bool have_first = false;
    if (td.mask == 0xffu) {
        const dbl2 *base = reinterpret_cast<const dbl2 *>(tval + (size_t)(td.soff - sub_base) * XT_SUB) + lane;
        XT_LD(va, 0)
        have_first = true;
    }
    // the window's 256 vector entries and the 256 column accumulators live in wave-private LDS: their reads are counted by
    // lgkmcnt, so waiting for them never drains the tile stream (vmcnt)
    {
        const dbl2 *src = reinterpret_cast<const dbl2 *>(vS + (size_t)it.w * XT_C) + 2 * lane;
        const dbl2 q0 = src[0], q1 = src[1];
        dbl2 z; z.x = 0.0; z.y = 0.0;
        reinterpret_cast<dbl2 *>(lcol)[2 * lane] = z; reinterpret_cast<dbl2 *>(lcol)[2 * lane + 1] = z;
        reinterpret_cast<dbl2 *>(lq)[2 * lane] = q0; reinterpret_cast<dbl2 *>(lq)[2 * lane + 1] = q1;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
#define XT_SUBBLOCK(vv, q)                                                                                                  \
    {                                                                                                                       \
        const dbl2 pc = *reinterpret_cast<const dbl2 *>(lq + XT_SBW * (q) + 2 * cc);                                        \
        double cax = 0.0, cay = 0.0;                                                                                        \
        _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) xt_acc<OP>(vv[j_], pc.x, pc.y, pr[j_], ra[j_], cax, cay, vpos);    \
        cax += __shfl_xor(cax, 16, WAVE); cay += __shfl_xor(cay, 16, WAVE);                                                 \
        cax += __shfl_xor(cax, 32, WAVE); cay += __shfl_xor(cay, 32, WAVE);                                                 \
        if (rr == 0) { dbl2 *l_ = reinterpret_cast<dbl2 *>(lcol + XT_SBW * (q) + 2 * cc); dbl2 o_ = *l_; o_.x += cax; o_.y += cay; *l_ = o_; } \
    }
#pragma unroll 1
    for (int t = it.t0; t < it.t1; ++t) {
        XTile nxt = td;
        if (t + 1 < it.t1) nxt = tiles[t + 1];                       // descriptor of the next tile: in flight behind this tile's stream
        const double *vr = vS + (size_t)td.k * XT_R + rr;
        double pr[8], ra[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { pr[j] = vr[4 * j]; ra[j] = 0.0; }
        const dbl2 *base = reinterpret_cast<const dbl2 *>(tval + (size_t)(td.soff - sub_base) * XT_SUB) + lane;
        if (td.mask == 0xffu) {
            // full tile: 64 KiB contiguous, two sub-blocks (16 KiB) in flight per wave
            dbl2 vb[8];
            if (!have_first) { XT_LD(va, 0) }
            have_first = false;
#pragma unroll 1
            for (int h = 0; h < 3; ++h) {                            // a real loop: the register budget stays at two sub-blocks
                XT_LD(vb, 2 * h + 1)
                XT_SUBBLOCK(va, 2 * h)
                XT_LD(va, 2 * h + 2)
                XT_SUBBLOCK(vb, 2 * h + 1)
            }
            XT_LD(vb, 7)
            XT_SUBBLOCK(va, 6)
            XT_SUBBLOCK(vb, 7)
        } else {
            unsigned mm = td.mask; int sl = 0;                     // partial tile (a few per cent of the storage): one sub-block at a time
#pragma unroll 1
            while (mm) {
                const int q = __ffs(mm) - 1; mm &= mm - 1;
                XT_LD(va, sl)
                XT_SUBBLOCK(va, q)
                ++sl;
            }
        }
        // 8 row sums per lane -> 32 row sums of the tile: butterfly over the 16 lanes that share rr (bits 3, 2, 1 of the lane
        // select which half survives, bit 0 completes the sum); fixed order
#pragma unroll
        for (int half = 4, bit = 8; half >= 1; half >>= 1, bit >>= 1) {
            const bool up = (lane & bit) != 0;
#pragma unroll
            for (int j = 0; j < half; ++j) {
                const double keep = up ? ra[j + half] : ra[j];
                const double send = up ? ra[j] : ra[j + half];
                ra[j] = keep + __shfl_xor(send, bit, WAVE);
            }
        }
        const double rs = ra[0] + __shfl_xor(ra[0], 1, WAVE);
        const int j = ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
        if (!(lane & 1)) rowpart[((size_t)td.k * nW + td.w) * XT_R + 4 * j + rr] = rs;
        td = nxt;
    }
#undef XT_LD
#undef XT_SUBBLOCK
    if (it.c == 0) {
        if (rr == 0) {
            double *cp = colpart + (size_t)it.pad * XT_C + 2 * cc;        // one 256-entry record per run, runs of a strip consecutive
#pragma unroll
            for (int q = 0; q < 8; ++q) *reinterpret_cast<dbl2 *>(cp + XT_SBW * q) = *reinterpret_cast<const dbl2 *>(lcol + XT_SBW * q + 2 * cc);
        }
        return;
    }
    // one record per workgroup: wave v adds the four waves' column sums of sub-blocks 2v, 2v + 1 in a fixed order
    __syncthreads();
    if (lane < 32) {
        const int wv = (int)(threadIdx.x >> 6);
        const double *all = lcol - (size_t)wv * (2 * XT_C);                // lcol[0][..] of the workgroup
        const int off = XT_SBW * (2 * wv + (lane >> 4)) + 2 * (lane & 15);
        const dbl2 a = *reinterpret_cast<const dbl2 *>(all + off), b = *reinterpret_cast<const dbl2 *>(all + 2 * XT_C + off);
        const dbl2 c2 = *reinterpret_cast<const dbl2 *>(all + 4 * XT_C + off), d = *reinterpret_cast<const dbl2 *>(all + 6 * XT_C + off);
        dbl2 o; o.x = (a.x + b.x) + (c2.x + d.x); o.y = (a.y + b.y) + (c2.y + d.y);
        *reinterpret_cast<dbl2 *>(colpart + (size_t)it.pad * XT_C + off) = o;
    }
}

#ifndef XT_RPG_FUSED
#define XT_RPG_FUSED 4           // atom rows per 8-lane group of the neighbour role inside k_xt_apply (85 k sites, same box: 2 -> 26.0, 4 -> 24.2, 8 -> 27.5 us)
#endif
// The neighbour part Xs of the product.  bid < nsb: atom rows, 8 lanes per row; bid = nsb, nsb + 1: the two driver rows (one
// workgroup each).
template <int RPG>
__device__ __forceinline__ void xt_neigh_roles(int bid, int nsb, int Nsub, const xrp_t *__restrict__ rp, const int *__restrict__ ci,
                                                 const double *__restrict__ val, const double *__restrict__ q, const double *__restrict__ sc,
                                                 const int *__restrict__ nsrank, const XCtrl *ctrl, double *__restrict__ t, double *red)
{
    if (bid < nsb) {
        // 8 lanes per row, RPG rows per lane group.  The first 32 entries of all RPG rows (every entry of an ordinary atom row) are
        // fetched together, each level of the dependent chain -- row pointers -> values / columns -> q -- issued for all rows
        // before the first use.  RPG = 4 next to the tile role, where the launch runs at 2 waves per SIMD and the loads in flight
        // per lane are what hides the latency (85 k sites: 27.8 -> 27.5 us); RPG = 1 in the stand-alone kernel, where 42 VGPRs give
        // full occupancy instead (9.4e5 sites: 38 us against 50 us with RPG = 4).
        const int g = threadIdx.x >> 3, l = threadIdx.x & 7;
        const int base = 2 + bid * (XT_NT / 8) * RPG + g;
        xrp_t p0[RPG], p1[RPG]; int sr[RPG]; double scale[RPG];
#pragma unroll
        for (int j = 0; j < RPG; ++j) {
            const int row = base + j * (XT_NT / 8);
            const bool ok = row < Nsub;
            p0[j] = ok ? rp[row] : 0; p1[j] = ok ? rp[row + 1] : 0; sr[j] = ok ? nsrank[row] : 0; scale[j] = ok ? sc[row] : 0.0;
        }
        if (ctrl->done) return;
        double v[RPG][4]; int c[RPG][4];
#pragma unroll
        for (int j = 0; j < RPG; ++j)
#pragma unroll
            for (int u = 0; u < 4; ++u) { const xrp_t p = p0[j] + l + 8 * u; const bool ok = p < p1[j]; v[j][u] = ok ? val[p] : 0.0; c[j][u] = ok ? ci[p] : 0; }
        double x[RPG][4];
#pragma unroll
        for (int j = 0; j < RPG; ++j)
#pragma unroll
            for (int u = 0; u < 4; ++u) x[j][u] = q[c[j][u]];
#pragma unroll
        for (int j = 0; j < RPG; ++j) {
            const int row = base + j * (XT_NT / 8);
            double s = (v[j][0] * x[j][0] + v[j][1] * x[j][1]) + (v[j][2] * x[j][2] + v[j][3] * x[j][3]);
            for (xrp_t pb = p0[j] + l + 32; pb < p1[j]; pb += 32) {            // rows with more than 32 entries
                double vv[4]; int cc[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const xrp_t p = pb + 8 * u; const bool ok = p < p1[j]; vv[u] = ok ? val[p] : 0.0; cc[u] = ok ? ci[p] : 0; }
                double xx[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) xx[u] = q[cc[u]];
                s += (vv[0] * xx[0] + vv[1] * xx[1]) + (vv[2] * xx[2] + vv[3] * xx[3]);
            }
            s = group_sum<8>(s);
            if (l == 0 && row < Nsub) t[row] = sr[j] < 0 ? scale[j] * s : s;
        }
        return;
    }
    if (ctrl->done) return;
    const int row = bid - nsb;                                      // 0 or 1
    const xrp_t p0 = rp[row], p1 = rp[row + 1];
    double s0 = 0.0, s1 = 0.0;
    xrp_t p = p0 + threadIdx.x;
    for (; p + XT_NT < p1; p += 2 * XT_NT) { s0 += val[p] * q[ci[p]]; s1 += val[p + XT_NT] * q[ci[p + XT_NT]]; }
    if (p < p1) s0 += val[p] * q[ci[p]];
    const double s = block_sum_all<XT_NT>(s0 + s1, red);
    if (threadIdx.x == 0) t[row] = sc[row] * s;
}
// Blocks 0, 1: the two driver rows (thousands of entries each: the longest dependent chain of the launch starts first).  Blocks
// [2, 2 + ntb): tile work items (one per wave).  The rest: the atom rows of Xs, 8 lanes per row, one row per group (no barrier,
// no partial sums: the p.t dot product is formed by the row kernel).  Non-S rows are finished here (scaled); S rows leave their sparse sum in t for the row kernel.
// NTL: non-temporal loads of the tile stream once a sweep no longer fits the 256 MiB Infinity Cache (cg.hip: 45 vs 53 us at
// 240 MB with the default policy, 478 vs 456 us at 1.86 GB).
template <int NTL>
__global__ __launch_bounds__(XT_NT) void k_xt_apply(int nitems, const XItem *__restrict__ items, const XTile *__restrict__ tiles, int sub_base,
                                                    const double *__restrict__ tval, const double *__restrict__ qS, int nW, int ns_pad,
                                                    double *__restrict__ rowpart, double *__restrict__ colpart, const XCtrl *ctrl,
                                                    int ntb, int nsb, int Nsub, const xrp_t *__restrict__ rp, const int *__restrict__ ci,
                                                    const double *__restrict__ val, const double *__restrict__ q, const double *__restrict__ sc,
                                                    const int *__restrict__ nsrank, double *__restrict__ t, int vb0)
{
    __shared__ double red[XT_NT / 64];
    __shared__ __attribute__((aligned(16))) double lcol[XT_NT / 64][2 * XT_C];
    // vb0: role offset of the launch's first workgroup.  0: the whole product in one launch.  A sharded solve launches the tile
    // roles alone (vb0 = 2, ntb workgroups) and the neighbour part as k_xt_neigh on a second stream, beside the exchange.
    const int vb = (int)blockIdx.x + vb0;
    // Workgroup -> role.  0, 1: the driver rows.  NTL = 0 (the sweep fits the Infinity Cache; the launch is ramp- and latency-bound):
    // tile and neighbour workgroups ALTERNATE in groups of 8 (8 consecutive workgroups land on the 8 XCDs, so every XCD sees both
    // kinds) while both kinds last, the rest of the longer list behind -- the neighbour rows, a latency chain at 2 waves per SIMD,
    // then run beside the tile stream from the first microsecond instead of in the tail of the launch (85 k sites: 27.2 -> 24.4 us;
    // all neighbour workgroups first: 25.5 us).
    // NTL = 1 (the tiles stream from HBM): all tile workgroups first; there the tile waves alone saturate HBM and every slot a
    // neighbour workgroup holds early costs tile bytes in flight (9.4e5 sites: 2.67 ms against 2.86 ms interleaved).
    int tile_idx = -1, nb_idx = -1;
    if (vb >= 2) {
        const int i = vb - 2, nmix = NTL ? 0 : (min(ntb, nsb) >> 3) << 3;
        if (i < 2 * nmix) { const int grp = i >> 3, idx = ((grp >> 1) << 3) + (i & 7); if (grp & 1) nb_idx = idx; else tile_idx = idx; }
        else { const int j = i - 2 * nmix; if (j < ntb - nmix) tile_idx = nmix + j; else nb_idx = nmix + j - (ntb - nmix); }
    }
    if (tile_idx >= 0) {
        const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        const int item = tile_idx * (XT_NT / 64) + wv;
        const XItem it = items[min(item, nitems - 1)];              // in flight together with the stop flag
        if (ctrl->done || item >= nitems) return;                   // (the flag is set only by the last kernel of an iteration: uniform over the launch)
        xt_tile_role<0, NTL>(it, tiles, sub_base, tval, qS, nW, ns_pad, rowpart, colpart, true, lcol[wv], lcol[wv] + XT_C);
        return;
    }
    xt_neigh_roles<XT_RPG_FUSED>(vb < 2 ? nsb + vb : nb_idx, nsb, Nsub, rp, ci, val, q, sc, nsrank, ctrl, t, red);
}
// The neighbour part alone (sharded solve: second stream, beside the exchange).  Same role bodies as in k_xt_apply, but compiled
// without the tile role's registers and LDS: twice the resident waves for what is a chain of three dependent latencies per row
// (DESIGN.md section 7).
__global__ __launch_bounds__(XT_NT) void k_xt_neigh(int nsb, int Nsub, const xrp_t *__restrict__ rp, const int *__restrict__ ci, const double *__restrict__ val,
                                                    const double *__restrict__ q, const double *__restrict__ sc, const int *__restrict__ nsrank,
                                                    const XCtrl *ctrl, double *__restrict__ t)
{
    __shared__ double red[XT_NT / 64];
    const int vb = (int)blockIdx.x;
    xt_neigh_roles<1>(vb < 2 ? nsb + vb : vb - 2, nsb, Nsub, rp, ci, val, q, sc, nsrank, ctrl, t, red);
}
// tiles only (diagonal pass with q = 1, power pass with q = m)
template <int OP>
__global__ __launch_bounds__(XT_NT) void k_xt_tiles_only(int nitems, const XItem *__restrict__ items, const XTile *__restrict__ tiles, int sub_base,
                                                         const double *__restrict__ tval, const double *__restrict__ vS, int nW, int ns_pad,
                                                         double *__restrict__ rowpart, double *__restrict__ colpart, int vpos)
{
    __shared__ __attribute__((aligned(16))) double lcol[XT_NT / 64][2 * XT_C];
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int item = (int)blockIdx.x * (XT_NT / 64) + wv;
    if (item >= nitems) return;
    xt_tile_role<OP, 1>(items[item], tiles, sub_base, tval, vS, nW, ns_pad, rowpart, colpart, vpos != 0, lcol[wv], lcol[wv] + XT_C);
}

// ---- row kernel: per S-row, row partials (ascending window) + column partials (ascending run) ------------------------------
// One workgroup per row block: 8 slices x 32 S-rows, slice sl adds every 8th window and every 8th run, the slices are combined
// in a fixed order through LDS.  MODE 0: t = s_i (sparse sum + tile sums) for the S rows, then the workgroup adds p.t, r.t and t.t
// over its share of ALL rows (the non-S rows were finished by the apply kernel) and writes one partial of each (XT_PSTRIDE
// apart).  MODE 1: xout[s] = tile sums
// only (this rank's share in the sharded solve; the diagonal and power passes).
// [w_lo, w_hi): windows that can hold partial sums in this launch -- all of them on one GPU; in a sharded solve the windows of this
// rank's tiles (every other cell of the partial arrays is zero on this rank: not reading it saves 7/8 of this kernel at 8 ranks)
// records of column sums of row block k's strip: count and first index.  nitem_w: [runs per strip | their exclusive scan | rec_shift]
// (xt_build_items); with rec_shift = 2 four runs share a record
__device__ __forceinline__ int xt_row_block_runs(int k, const int *__restrict__ nitem_w, int w_lo, int w_hi, int nW)
{
    const int wk = k / (XT_C / XT_R);
    return (wk >= w_lo && wk < w_hi) ? nitem_w[wk] >> nitem_w[2 * (nW + 2)] : 0;
}
__device__ __forceinline__ int xt_row_block_cbase(int k, int nW, const int *__restrict__ nitem_w)
{
    return nitem_w[nW + 2 + k / (XT_C / XT_R)] >> nitem_w[2 * (nW + 2)];
}
// wr = wrange[k], nc = xt_row_block_runs(k, ...): fetched by the caller, with whatever else starts its chain
__device__ __forceinline__ double xt_row_block_sum(int k, int nW, int cbase, int2 wr, int nc,
                                                   const double *__restrict__ rowpart, const double *__restrict__ colpart, double (*sl_sum)[XT_R],
                                                   int w_lo, int w_hi)
{
    const int ns_pad = XT_C;                                          // distance between the records of consecutive runs of the strip
    const int r = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int s = XT_R * k + r;
    wr.x = max(wr.x, w_lo); wr.y = min(wr.y, w_hi);
    if (wr.x >= wr.y && nc == 0) return 0.0;                          // nothing of this row block on this rank (uniform over the workgroup)
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    const double *rpp = rowpart + (size_t)k * nW * XT_R + r;
    const double *cpp = colpart + (size_t)cbase * XT_C + (s - XT_C * (k / (XT_C / XT_R)));
    int w = wr.x + sl, c = sl;
    // independent loads in flight per list: 16 while the list is long (the strips at the tapered end of a share hold thousands of
    // single-tile runs: 300 terms per slice, whose latency rounds are what this kernel costs at 9.4e5 sites), then 8, 4, 1.  Term j of a
    // slice always goes to accumulator j % 4, whatever the unrolling: the sum is the same bits.
    for (; c + 120 < nc; c += 128) {
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = cpp[(size_t)(c + 8 * u) * ns_pad];
#pragma unroll
        for (int u = 0; u < 16; u += 4) { a0 += v[u]; a1 += v[u + 1]; a2 += v[u + 2]; a3 += v[u + 3]; }
    }
    for (; c + 56 < nc; c += 64) {                                    // tapered shares end in strips of single-tile items: up to nK terms
        const double v0 = cpp[(size_t)c * ns_pad], v1 = cpp[(size_t)(c + 8) * ns_pad], v2 = cpp[(size_t)(c + 16) * ns_pad], v3 = cpp[(size_t)(c + 24) * ns_pad];
        const double v4 = cpp[(size_t)(c + 32) * ns_pad], v5 = cpp[(size_t)(c + 40) * ns_pad], v6 = cpp[(size_t)(c + 48) * ns_pad], v7 = cpp[(size_t)(c + 56) * ns_pad];
        a0 += v0; a1 += v1; a2 += v2; a3 += v3; a0 += v4; a1 += v5; a2 += v6; a3 += v7;
    }
    for (; c + 24 < nc; c += 32) { a0 += cpp[(size_t)c * ns_pad]; a1 += cpp[(size_t)(c + 8) * ns_pad]; a2 += cpp[(size_t)(c + 16) * ns_pad]; a3 += cpp[(size_t)(c + 24) * ns_pad]; }
    for (; c < nc; c += 8) a0 += cpp[(size_t)c * ns_pad];
    for (; w + 24 < wr.y; w += 32) { a0 += rpp[(size_t)w * XT_R]; a1 += rpp[(size_t)(w + 8) * XT_R]; a2 += rpp[(size_t)(w + 16) * XT_R]; a3 += rpp[(size_t)(w + 24) * XT_R]; }
    for (; w < wr.y; w += 8) a1 += rpp[(size_t)w * XT_R];
    const double mine = (a0 + a1) + (a2 + a3);
    __syncthreads();                                                 // previous round's readers are done with sl_sum
    sl_sum[sl][r] = mine;
    __syncthreads();
    double tot = 0.0;
    if (sl == 0) {
#pragma unroll
        for (int u = 0; u < 8; ++u) tot += sl_sum[u][r];
    }
    return tot;                                                      // valid in threads 0..31
}
#define XT_PSTRIDE 1024          // distance between the p.t, r.t and t.t partial arrays
// the three dot products of an iteration from one pass over p, r, t: p.t gives alpha; r.t and t.t give r'.r' = r.r + 2 alpha r.t +
// alpha^2 t.t for r' = r + alpha t WITHOUT a second global reduction, so that the vector update and the new direction fit in
// one kernel (k_xt_step).  The identity is exact for the r' actually formed (it does not assume conjugacy).
// block sums of NV values with one LDS round (two barriers), fixed order; red4: [XT_NT / 64][4] doubles
template <int NV>
__device__ __forceinline__ void xt_block_sums(double (&v)[NV], double (*red4)[4])
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int u = 0; u < NV; ++u) v[u] = wave_sum(v[u]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int u = 0; u < NV; ++u) red4[w][u] = v[u];
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NV; ++u) { double s = 0.0; for (int i = 0; i < XT_NT / 64; ++i) s += red4[i][u]; v[u] = s; }
}
__device__ __forceinline__ void xt_dots_write(double a_pt, double a_rt, double a_tt, double (*red4)[4], double *part)
{
    double v[3] = {a_pt, a_rt, a_tt};
    xt_block_sums<3>(v, red4);
    if (threadIdx.x == 0) { part[blockIdx.x] = v[0]; part[XT_PSTRIDE + blockIdx.x] = v[1]; part[2 * XT_PSTRIDE + blockIdx.x] = v[2]; }
}
template <int MODE>
__global__ __launch_bounds__(XT_NT) void k_xt_rows(int ns, int nK, int nW, int ns_pad, const int2 *__restrict__ wrange, const int *__restrict__ nitem_w,
                                                   const double *__restrict__ rowpart, const double *__restrict__ colpart,
                                                   const int *__restrict__ srow, const double *__restrict__ sS, const double *__restrict__ pvec,
                                                   double *__restrict__ t, double *__restrict__ part, const XCtrl *ctrl, double *__restrict__ xout,
                                                   int m, const int *__restrict__ nsrank, int flag_rank0, const double *__restrict__ rvec,
                                                   int w_lo, int w_hi)
{
    __shared__ double red[XT_NT / 64][4];
    __shared__ double sl_sum[8][XT_R];
    __shared__ int sdone;
    double acc = 0.0, arr = 0.0, att = 0.0;
    // Everything that does not depend on the partial sums is fetched first -- the S rows' own entries (row index, scale, sparse
    // sum, p, r) and this workgroup's share of the non-S rows for the dot products -- so that the dependent chain of the launch is
    // flag -> partial sums -> store instead of flag -> partial sums -> row index -> vectors -> store.
    const int chunk = (m + (int)gridDim.x - 1) / (int)gridDim.x;
    const int i0 = blockIdx.x * chunk, i1 = min(m, i0 + chunk);
    const int ifirst = i0 + (int)threadIdx.x;
    double fp = 0.0, ft = 0.0, fr = 0.0; int fs = 0;
    if (MODE == 0 && ifirst < i1) { fs = nsrank[ifirst]; fp = pvec[ifirst]; ft = t[ifirst]; fr = rvec[ifirst]; }
    int row0 = -1; double s0 = 0.0, t0 = 0.0, p0 = 0.0, r0 = 0.0;
    {
        const int s = XT_R * (int)blockIdx.x + (int)threadIdx.x;
        if (MODE == 0 && (int)blockIdx.x < nK && threadIdx.x < XT_R && s < ns) { row0 = srow[s]; s0 = sS[s]; t0 = t[row0]; p0 = pvec[row0]; r0 = rvec[row0]; }
    }
    // ... and the extent of the first row block's partial lists: one dependent level less behind the flag
    int2 wr0 = make_int2(0, 0); int nc0 = 0;
    int cb0 = 0;
    if ((int)blockIdx.x < nK) { wr0 = wrange[blockIdx.x]; nc0 = xt_row_block_runs(blockIdx.x, nitem_w, w_lo, w_hi, nW); cb0 = xt_row_block_cbase(blockIdx.x, nW, nitem_w); }
    if (ctrl) {
        if (threadIdx.x == 0) sdone = ctrl->done;
        __syncthreads();
        if (sdone) return;
    }
    for (int k = blockIdx.x; k < nK; k += gridDim.x) {
        const bool own = k == (int)blockIdx.x;
        const double sum = xt_row_block_sum(k, nW, own ? cb0 : xt_row_block_cbase(k, nW, nitem_w), own ? wr0 : wrange[k], own ? nc0 : xt_row_block_runs(k, nitem_w, w_lo, w_hi, nW), rowpart, colpart, sl_sum, w_lo, w_hi);
        const int s = XT_R * k + (int)threadIdx.x;
        if (threadIdx.x < XT_R && s < ns) {
            if (MODE == 1) xout[s] = sum;
            else if (k == (int)blockIdx.x) { const double tv = s0 * (t0 + sum); t[row0] = tv; acc += p0 * tv; arr += r0 * tv; att += tv * tv; }
            else { const int row = srow[s]; const double tv = sS[s] * (t[row] + sum); t[row] = tv; acc += pvec[row] * tv; arr += rvec[row] * tv; att += tv * tv; }
        }
    }
    if (MODE == 1) {
        // the stop decision (rank 0's) and the abort word (any rank's) ride in the all-reduce, slots ns and ns + 1
        if (ctrl && blockIdx.x == 0 && threadIdx.x == 0) { xout[ns] = (flag_rank0 && ctrl->done_local) ? 1.0 : 0.0; xout[ns + 1] = ctrl->abort_local ? 1.0 : 0.0; }
        return;
    }
    // the non-S rows of this workgroup's share of the vector
    if (ifirst < i1 && fs < 0) { acc += fp * ft; arr += fr * ft; att += ft * ft; }
    for (int i = ifirst + XT_NT; i < i1; i += XT_NT) if (nsrank[i] < 0) { const double tv = t[i]; acc += pvec[i] * tv; arr += rvec[i] * tv; att += tv * tv; }
    xt_dots_write(acc, arr, att, red, part);
}
// sharded solve, after the all-reduce of xout: finish the S rows, then the same partials as MODE 0
__global__ __launch_bounds__(XT_NT) void k_xt_rows_apply(int ns, int nK, const double *__restrict__ xbuf, const int *__restrict__ srow,
                                                         const double *__restrict__ sS, const double *__restrict__ pvec, double *__restrict__ t,
                                                         double *__restrict__ part, XCtrl *ctrl, int m, const int *__restrict__ nsrank,
                                                         const double *__restrict__ rvec)
{
    __shared__ double red[XT_NT / 64][4];
    __shared__ int sdone;
    if (threadIdx.x == 0) sdone = ctrl->done || xbuf[ns] != 0.0 || xbuf[ns + 1] != 0.0;     // xbuf[ns]: rank 0's stop decision; xbuf[ns + 1]: a rank aborted -- the same on every rank
    __syncthreads();
    if (sdone) { if (blockIdx.x == 0 && threadIdx.x == 0) { ctrl->done = 1; if (xbuf[ns + 1] != 0.0) ctrl->aborted = 1; } return; }
    double acc = 0.0, arr = 0.0, att = 0.0;
    for (int k = blockIdx.x; k < nK; k += gridDim.x) {
        const int s = XT_R * k + (int)threadIdx.x;
        if (threadIdx.x < XT_R && s < ns) { const int row = srow[s]; const double tv = sS[s] * (t[row] + xbuf[s]); t[row] = tv; acc += pvec[row] * tv; arr += rvec[row] * tv; att += tv * tv; }
    }
    const int chunk = (m + (int)gridDim.x - 1) / (int)gridDim.x;
    const int i0 = blockIdx.x * chunk, i1 = min(m, i0 + chunk);
    for (int i = i0 + threadIdx.x; i < i1; i += XT_NT) if (nsrank[i] < 0) { const double tv = t[i]; acc += pvec[i] * tv; arr += rvec[i] * tv; att += tv * tv; }
    xt_dots_write(acc, arr, att, red, part);
}

// ---- vector kernels of the CG (sign convention and stop tests of solve_sparse_CG_Jacobi, iterative_solvers_gpu.cu:349-459) ----
// diag -= tile row sums (S rows); s = 1/sqrt(diag); b *= s; y /= s; q = s * y (= the caller's y); compact copies over S
__global__ void k_xt_scale_init(int Nsub, const xrp_t *__restrict__ diag_pos, double *__restrict__ val, const int *__restrict__ nsrank,
                                const double *__restrict__ tsum, double *__restrict__ sc, double *__restrict__ b, double *__restrict__ y,
                                double *__restrict__ q, double *__restrict__ sS, double *__restrict__ qS)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Nsub) return;
    const int sr = nsrank[i];
    double d = val[diag_pos[i]];
    if (sr >= 0) { d = d + -tsum[sr]; val[diag_pos[i]] = d; }       // calc_diagonal_X_gpu: diag += -(sum of the row's off-diagonals)
    const double s = 1.0 / sqrt(d);
    sc[i] = s;
    b[i] = b[i] * s;
    const double ys = y[i] * 1 / s;
    y[i] = ys;
    const double qv = s * ys;
    q[i] = qv;
    if (sr >= 0) { sS[sr] = s; qS[sr] = qv; }
}
// r = t - b ; p = -r ; q = s p ; partial r.r
__global__ __launch_bounds__(XT_NT) void k_xt_resid_init(int m, const double *__restrict__ t, const double *__restrict__ b, double *__restrict__ r,
                                                         double *__restrict__ p, const double *__restrict__ sc, double *__restrict__ q,
                                                         const int *__restrict__ nsrank, double *__restrict__ qS, double *__restrict__ part)
{
    __shared__ double red[XT_NT / 64];
    double acc = 0.0;
    for (int i = blockIdx.x * XT_NT + threadIdx.x; i < m; i += gridDim.x * XT_NT) {
        const double rv = -b[i] + t[i];
        r[i] = rv; p[i] = -rv; acc += rv * rv;
        const double qv = sc[i] * -rv;
        q[i] = qv;
        const int sr = nsrank[i];
        if (sr >= 0) qS[sr] = qv;
    }
    const double tot = block_sum_all<XT_NT>(acc, red);
    if (threadIdx.x == 0) part[blockIdx.x] = tot;
}
__global__ __launch_bounds__(XT_NT) void k_xt_check0(const double *part, int npart, XCtrl *ctrl, double tol2)
{
    __shared__ double red[XT_NT / 64];
    double s = 0.0;
    for (int i = threadIdx.x; i < npart; i += XT_NT) s += part[i];
    const double rr = block_sum_all<XT_NT>(s, red);
    if (threadIdx.x == 0) { ctrl->rr[0] = rr; ctrl->rr[1] = rr; ctrl->iters = 0; const int d = !(sqrt(rr) > tol2); if (ctrl->sharded) ctrl->done_local = d; else ctrl->done = d; }
}
// One kernel per iteration for everything after the matrix-vector product:
//   alpha = r.r / p.t ; y += alpha p ; r' = r + alpha t ; beta = r'.r' / r.r ; p = beta p - r' ; q = s p ; stop test on r'.r'
// (solve_sparse_CG_Jacobi, iterative_solvers_gpu.cu:424-455).  r.r is the DIRECT sum of squares of the current r (partials of the
// previous launch of this kernel, double-buffered); r'.r' -- needed for beta before r' exists everywhere -- is formed as
// r.r + 2 alpha r.t + alpha^2 t.t from the partials of the row kernel (exact for the r' formed here; rounding ~1e-16 r.r, the
// same order as regrouping a sum).  The reference's separate update / direction passes need two global reductions for this.
__global__ __launch_bounds__(XT_NT) void k_xt_step(int m, int it, const double *__restrict__ part3, int npart, const double *__restrict__ part_rr_in,
                                                   int npart_rr, double *__restrict__ part_rr_out, double *__restrict__ p,
                                                   const double *__restrict__ t, double *__restrict__ y, double *__restrict__ r,
                                                   const double *__restrict__ sc, double *__restrict__ q, const int *__restrict__ nsrank,
                                                   double *__restrict__ qS, XCtrl *ctrl, double tol2)
{
    __shared__ double red[XT_NT / 64][4];
    __shared__ int sdone;
    // `done` is a stop word stamped with the iteration: 0 = keep going, d > 0 = kernels of iterations >= d - 1 must not run.  This
    // kernel is the one launch that both reads and (workgroup 0, below) writes it: the value it writes, it + 2, does not stop
    // iteration `it`, so a workgroup scheduled after workgroup 0 has published still updates its slice of y and r (a plain 0/1 flag
    // let such a workgroup skip y += alpha p on the converging iteration: a partially updated, run-dependent solution).
    if (threadIdx.x == 0) { const int d = ctrl->done; sdone = d != 0 && it + 1 >= d; }
    // the first four elements of this thread are fetched before the partial sums are reduced: their latency hides behind the
    // reduction chain (one element batch covers the whole vector unless the grid is capped)
    const int stride = gridDim.x * XT_NT, i0 = blockIdx.x * XT_NT + threadIdx.x;
    double pv[4], tv[4], yv[4], rv[4], sv[4]; int srk[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * stride;
        const bool ok = i < m;
        pv[u] = ok ? p[i] : 0.0; tv[u] = ok ? t[i] : 0.0; yv[u] = ok ? y[i] : 0.0; rv[u] = ok ? r[i] : 0.0; sv[u] = ok ? sc[i] : 0.0;
        srk[u] = ok ? nsrank[i] : -1;
    }
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    for (int i = threadIdx.x; i < npart; i += XT_NT) { a[0] += part3[i]; a[1] += part3[XT_PSTRIDE + i]; a[2] += part3[2 * XT_PSTRIDE + i]; }
    for (int i = threadIdx.x; i < npart_rr; i += XT_NT) a[3] += part_rr_in[i];
    xt_block_sums<4>(a, red);
    if (sdone) return;
    const double pAp = a[0], rt = a[1], tt = a[2], rr = a[3];
    const double alpha = rr / pAp;
    const double rr_new = rr + 2.0 * alpha * rt + alpha * alpha * tt;
    const double beta = rr_new / rr;
    double acc = 0.0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * stride;
        if (i < m) {
            y[i] = yv[u] + alpha * pv[u];
            const double rn = rv[u] + alpha * tv[u];
            r[i] = rn;
            acc += rn * rn;
            const double pn = pv[u] * beta - rn;
            p[i] = pn;
            const double qv = sv[u] * pn;
            q[i] = qv;
            if (srk[u] >= 0) qS[srk[u]] = qv;
        }
    }
    for (int i = i0 + 4 * stride; i < m; i += stride) {
        const double pw = p[i];
        y[i] += alpha * pw;
        const double rn = r[i] + alpha * t[i];
        r[i] = rn;
        acc += rn * rn;
        const double pn = pw * beta - rn;
        p[i] = pn;
        const double qv = sc[i] * pn;
        q[i] = qv;
        const int sr = nsrank[i];
        if (sr >= 0) qS[sr] = qv;
    }
    double av[1] = {acc};
    xt_block_sums<1>(av, red);
    if (threadIdx.x == 0) part_rr_out[blockIdx.x] = av[0];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        ctrl->rr[(it + 1) & 1] = rr_new;
        ctrl->iters = it + 1;
        if (!(rr_new > tol2)) { if (ctrl->sharded) ctrl->done_local = 1; else ctrl->done = it + 2; }
    }
}
__global__ void k_xt_set_sharded(XCtrl *ctrl) { ctrl->sharded = 1; }
// test aid (tests/test_gpu_parity.py; no counterpart in the reference): the stamped stop word of k_xt_step, deterministically.  Runs ONE launch of
// iteration `it` over m elements (p = t = r = 1, y = 0, partial sums such that alpha = 1 and r'.r' > tol^2) with ctrl->done preset to
// `done_word`, and counts the elements of y the launch updated.  0 and it + 2 (the value workgroup 0 of the SAME launch publishes when the
// iteration converges) must update every element -- whichever workgroups see the word; it + 1 and below (a stop published by an
// earlier iteration) must update none.
__global__ void k_xt_dbg_fill(int m, double *p, double *t, double *y, double *r, double *sc, int *nsrank)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) { p[i] = 1.0; t[i] = 1.0; y[i] = 0.0; r[i] = 1.0; sc[i] = 1.0; nsrank[i] = -1; }
}
__global__ void k_xt_dbg_count(int m, const double *y, int *n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m && y[i] != 0.0) atomicAdd(n, 1);
}
extern "C" int dkmc_debug_step_stop_word(int m, int it, int done_word, int *updated, int *done_after)
{
    Engine &e = eng(); hipStream_t st = e.stream;
    if (m < 1 || it < 0 || !updated) return dkmc_fail(13, "debug_step_stop_word: bad arguments", __FILE__, __LINE__);
    double *buf = nullptr; int *ibuf = nullptr;
    const size_t nd = (size_t)6 * m + 3 * XT_PSTRIDE + 2 * 512 + 8;
    HIPCHK(hipMalloc((void **)&buf, nd * 8));
    HIPCHK(hipMalloc((void **)&ibuf, ((size_t)m + 4) * 4 + sizeof(XCtrl)));
    HIPCHK(hipMemsetAsync(buf, 0, nd * 8, st));
    HIPCHK(hipMemsetAsync(ibuf, 0, ((size_t)m + 4) * 4 + sizeof(XCtrl), st));
    double *p = buf, *t = p + m, *y = t + m, *r = y + m, *sc = r + m, *q = sc + m, *part3 = q + m, *prr = part3 + 3 * XT_PSTRIDE, *qS = prr + 2 * 512;
    int *nsrank = ibuf, *cnt = ibuf + m;
    XCtrl *ctrl = reinterpret_cast<XCtrl *>(ibuf + m + 4);
    const int gv = xt_grid((m + XT_NT - 1) / XT_NT, 4, 512);
    hipLaunchKernelGGL(k_xt_dbg_fill, dim3((m + 255) / 256), dim3(256), 0, st, m, p, t, y, r, sc, nsrank);
    // p.t = m, r.t = -m / 2, t.t = m, r.r = m: alpha = 1, r'.r' = m: no convergence at any tolerance below m
    const double h_part[3] = {(double)m, -0.5 * (double)m, (double)m}, h_rr = (double)m;
    for (int k = 0; k < 3; ++k) HIPCHK(hipMemcpyAsync(part3 + k * XT_PSTRIDE, &h_part[k], 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(prr + 512 * (it & 1), &h_rr, 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(&ctrl->done, &done_word, 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_xt_step, dim3(gv), dim3(XT_NT), 0, st, m, it, (const double *)part3, 1, (const double *)(prr + 512 * (it & 1)), 1,
                       prr + 512 * ((it + 1) & 1), p, (const double *)t, y, r, (const double *)sc, q, (const int *)nsrank, qS, ctrl, 1e-30);
    hipLaunchKernelGGL(k_xt_dbg_count, dim3((m + 255) / 256), dim3(256), 0, st, m, (const double *)y, cnt);
    int h_cnt = 0, h_done = 0;
    HIPCHK(hipMemcpyAsync(&h_cnt, cnt, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&h_done, &ctrl->done, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    (void)hipFree(buf); (void)hipFree(ibuf);
    KCHK();
    *updated = h_cnt;
    if (done_after) *done_after = h_done;
    return e.err_code;
}

// a rank whose host side failed between two collectives: its contribution to the next all-reduce says so (whatever else it holds)
__global__ void k_xt_abort_word(XCtrl *ctrl, double *xbuf, int ns) { ctrl->abort_local = 1; xbuf[ns + 1] = 1.0; }
// test aid: make this rank fail once, in the assembly (phase 1) or on the host side of CG iteration `iteration` (phase 2)
static int g_fault_phase = 0, g_fault_iter = 0;
extern int g_xtb_fault_iter;                 // xtb.hip: phase 3 = host side of a block-CG iteration of a sharded solve
extern "C" void dkmc_debug_inject_fault(int phase, int iteration) { if (phase == 3) { g_xtb_fault_iter = iteration; return; } g_fault_phase = phase; g_fault_iter = iteration; }
// q = s y (full length and compact over S): the product's input when a solve continues from an iterate another loop left in y
__global__ void k_xt_requeue(int m, const double *__restrict__ y, const double *__restrict__ sc, const int *__restrict__ nsrank, double *__restrict__ q, double *__restrict__ qS)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const double qv = sc[i] * y[i];
    q[i] = qv;
    const int sr = nsrank[i];
    if (sr >= 0) qS[sr] = qv;
}
__global__ void k_xt_vec_mul(int m, double *__restrict__ y, const double *__restrict__ s)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) y[i] = y[i] * s[i];
}
__global__ void k_xt_fill_f64(double *p, long long n, double v)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
__global__ void k_xt_node_srank(int Nsub, const int *__restrict__ srank, int *__restrict__ node_srank)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < Nsub) node_srank[i] = i < 2 ? -1 : srank[i - 2];
}
__global__ void k_xt_gather_S(int ns, const int *__restrict__ srow, const double *__restrict__ v, double *__restrict__ vS)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < ns) vS[s] = v[srow[s]];
}
// dissipated power of every atom row: sparse part (8 lanes per row) + the tile sums of S rows; host formula on X's pattern
// (current_solver.cpp:299-357; see SURVEY B8/B9 for the index slips of the CUDA kernels this replaces)
__global__ __launch_bounds__(XT_NT) void k_xt_power_rows(int Nsub, const xrp_t *__restrict__ rp, const int *__restrict__ ci, const double *__restrict__ val,
                                                         const double *__restrict__ m, double Vd, const int *__restrict__ nsrank,
                                                         const double *__restrict__ psumS, const int *__restrict__ aflag,
                                                         const int *__restrict__ atom_site, double alpha, double *__restrict__ site_power)
{
    const int g = threadIdx.x >> 3, l = threadIdx.x & 7;
    const int i = 2 + blockIdx.x * (XT_NT / 8) + g;
    if (i >= Nsub) return;
    const double mi = m[i];
    double p = 0.0;
    for (xrp_t q = rp[i] + l; q < rp[i + 1]; q += 8) {
        const int c = ci[q];
        if (c < 2 || c == i) continue;
        const double ical = val[q] * (mi - m[c]);
        double v = 0.0;
        if ((ical < 0 && Vd > 0) || (ical > 0 && Vd < 0)) v = -ical;
        p += v * (m[c] - mi);
    }
    p = group_sum<8>(p);
    const int a = i - 2;
    if (l == 0 && !(aflag[a] & AF_METAL)) {
        const int sr = nsrank[i];
        if (sr >= 0) p += psumS[sr];
        site_power[atom_site[a]] = -1 * alpha * p;
    }
}

// ---- host side -----------------------------------------------------------------------------------------------------------------
XTState g_xt;
XTBuffers g_xb;


// one pass over this rank's tiles with the vector vS (compact over S), tile sums of every S-row into out[ns] (all ranks)
template <int OP>
static int xt_tile_sums(const double *vS, double *out, int vpos)
{
    Engine &e = eng(); hipStream_t st = e.stream; const XTState &X = g_xt;
    if (X.item_n > 0)
        hipLaunchKernelGGL((k_xt_tiles_only<OP>), dim3((X.item_n + 3) / 4), dim3(XT_NT), 0, st, X.item_n, (const XItem *)g_xb.items + X.item_lo,
                           (const XTile *)g_xb.tiles, (int)X.sub_base, (const double *)g_xb.tval, vS, X.nW, X.ns_pad, g_xb.rowpart, g_xb.colpart, vpos);
    hipLaunchKernelGGL((k_xt_rows<1>), dim3(std::max(X.nK, 1)), dim3(XT_NT), 0, st, X.ns, X.nK, X.nW, X.ns_pad, (const int2 *)g_xb.wrange,
                       (const int *)g_xb.nitem_w, (const double *)g_xb.rowpart, (const double *)g_xb.colpart, (const int *)nullptr,
                       (const double *)nullptr, (const double *)nullptr, (double *)nullptr, (double *)nullptr, (const XCtrl *)nullptr, out,
                       0, (const int *)nullptr, 0, (const double *)nullptr, X.w_lo, X.w_hi);
    KCHK();
    if (comm_attached()) {                                    // complete the sums, then rank 0's bits for everyone (diagonal, power)
        int rc = comm_allreduce_sum_f64(out, (size_t)X.ns); if (rc) return rc;
        rc = comm_bcast0_f64(out, (size_t)X.ns); if (rc) return rc;
    }
    return 0;
}

int xt_tile_sums_mv(const double *vS, double *out) { return xt_tile_sums<0>(vS, out, 1); }      // xtb.hip (test aid)

// Builds the work items for an nranks-way split and reports rank `me`'s share.  Slots: where the item arrays go (the resident ones of
// an assembly, or temporary ones of dkmc_xt_time_share).
static int xt_build_items(int nK, int nW, int kc, int ntiles, long long nsub_total, const int *toff, const XTile *tiles, int nranks, int me,
                          int slot_nitemw, int slot_items, int slot_split, XShare *out, int rec_shift = 0)
{
    Engine &e = eng(); hipStream_t st = e.stream;
    if (nranks > XT_MAXRANKS) return dkmc_fail(46, "update_power: more ranks than XT_MAXRANKS", __FILE__, __LINE__);
    int *nitem_w = (int *)scratch(slot_nitemw, (size_t)(nW + 4) * 4 * 3);
    XSplit *sp = (XSplit *)scratch(slot_split, sizeof(XSplit));
    if (!nitem_w || !sp) return e.err_code;
    int *ioff = nitem_w + nW + 2;
    XSplit h{};
    out->nitem_w = nitem_w; out->items = nullptr; out->nitems = 0; out->maxchunk = 1; out->rec_shift = rec_shift;
    out->item_lo = 0; out->item_n = 0; out->tile_lo = 0; out->tile_n = 0; out->sub_base = 0; out->sub_n = 0; out->w_lo = 0; out->w_hi = 0;
    if (ntiles <= 0) { HIPCHK(hipMemsetAsync(nitem_w, 0, (size_t)(nW + 4) * 4 * 3, st)); return 0; }
    hipLaunchKernelGGL(k_xt_split, dim3(1), dim3(XT_MAXRANKS + 1), 0, st, ntiles, nsub_total, tiles, nranks, sp);
    hipLaunchKernelGGL((k_xt_items<0>), dim3((nW + 255) / 256), dim3(256), 0, st, nK, nW, kc, ntiles, toff, (const int *)nullptr, tiles, sp, nitem_w, (XItem *)nullptr, rec_shift);
    int rc = dkmc_exclusive_scan_i32(nitem_w, ioff, nW, ioff + nW); if (rc) return rc;
    XItem *items = (XItem *)scratch(slot_items, (size_t)(ntiles + 4 * (size_t)(nW + nranks) + 4) * sizeof(XItem));      // an item holds at least one tile (+ up to 3 empty runs per strip and per share end): no need to wait for the count
    if (!items) return e.err_code;
    hipLaunchKernelGGL((k_xt_items<1>), dim3((nW + 255) / 256), dim3(256), 0, st, nK, nW, kc, ntiles, toff, (const int *)ioff, tiles, sp, nitem_w, items, rec_shift);
    hipLaunchKernelGGL(k_xt_rank_items, dim3(1), dim3(XT_MAXRANKS + 1), 0, st, (const int *)(ioff + nW), ntiles, nsub_total, (const XItem *)items, tiles, sp);
    HIPCHK(hipMemcpyAsync(&h, sp, sizeof(XSplit), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    const int nitems = h.nitems;
    out->items = items; out->nitems = nitems; out->maxchunk = std::max(1, h.max_items_per_strip);
    out->item_lo = h.item_lo[me]; out->item_n = h.item_lo[me + 1] - h.item_lo[me];
    out->tile_lo = h.tb[me]; out->tile_n = h.tb[me + 1] - h.tb[me];
    out->sub_base = h.soff[me]; out->sub_n = h.soff[me + 1] - h.soff[me];
    out->w_lo = h.w_first[me]; out->w_hi = out->tile_n > 0 ? h.w_last[me + 1] + 1 : out->w_lo;
    return 0;
}

// Second stream of the sharded solve: the neighbour part of the product (replicated on every rank, ~0.3 GB at 9.4e5 sites) runs
// here, beside the tile pass, the partial row sums and the all-reduce on the engine's stream; the two meet again before the rows
// are finished.  Events from a ring: a wait refers to the record that preceded it, the host may run a whole batch ahead.
#define XT_SIDE_RING 64
struct XSide { hipStream_t st = nullptr; hipEvent_t a[XT_SIDE_RING], b[XT_SIDE_RING]; int device = -1; unsigned seq = 0; bool ready = false; };
static XSide g_side;
static int xt_side_init()
{
    XSide &S = g_side; const int dev = eng().device;
    if (S.ready && S.device == dev) return 0;
    if (S.ready) { (void)hipStreamDestroy(S.st); for (int i = 0; i < XT_SIDE_RING; ++i) { (void)hipEventDestroy(S.a[i]); (void)hipEventDestroy(S.b[i]); } S.ready = false; }
    HIPCHK(hipStreamCreateWithFlags(&S.st, hipStreamNonBlocking));
    for (int i = 0; i < XT_SIDE_RING; ++i) {
        HIPCHK(hipEventCreateWithFlags(&S.a[i], hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&S.b[i], hipEventDisableTiming));
    }
    S.device = dev; S.seq = 0; S.ready = true;
    return 0;
}

// Assemble Xs + tiles and solve X m = rhs with the Jacobi-scaled CG.  aneigh/ancnt/aflag/srank/S/atom_site: current.hip steps 1-2.
int xt_assemble_and_solve(dkmc_gpubuf *buf, const XParams &P, int ns, const SEntry *S, const int *aneigh, const int *ancnt, const int *aflag,
                          const int *srank, const int *atom_site, double *rhs, double *y, int *iters_out, double *rr_out, double *yaux, int yaux_valid)
{
    TCacheView TC{};                     // brought up to date for this rank's share inside assemble() (tc_prepare_tiled, current.hip)
    Engine &e = eng(); hipStream_t st = e.stream;
    XTState &X = g_xt; X.valid = false;
    const int Na = P.Na, Nsub = Na + 1;
    X.Nsub = Nsub; X.ns = ns;
    const int ns_pad = ((ns + XT_C - 1) / XT_C) * XT_C + XT_C;          // one window of slack: tiles of the last row block read pS[32k + ...] up to ns_pad
    X.ns_pad = ns_pad;
    const int nK = (ns + XT_R - 1) / XT_R, nW = (ns + XT_C - 1) / XT_C;
    X.nK = nK; X.nW = nW;
    const long long ncell = (long long)nK * nW;
    if (ncell > 0x7ffffff0ll) return dkmc_fail(47, "update_power: too many tile cells", __FILE__, __LINE__);

    // Everything up to the first collective is LOCAL work (pattern, census, tile list, this rank's share, storage, fill): in a sharded
    // solve a failure here (an allocation that does not fit, a launch error) must not leave the peers blocked in the collective that
    // follows -- the ranks agree on the outcome of this phase first (comm_agree) and leave together.
    const int nbr = (Nsub + 255) / 256, m = Nsub;
    const bool sharded = comm_attached() != 0;
    const int nr = sharded ? comm_nranks() : 1, me = sharded ? comm_rank() : 0;
    int rc = 0, ntiles = 0;
    long long xs_nnz = 0;
    int *cnt = nullptr, *nsrank = nullptr, *col = nullptr, *si = nullptr, *srow = nullptr, *nitem_w = nullptr;
    xrp_t *rp = nullptr, *dpos = nullptr;
    double *val = nullptr, *sd = nullptr, *tval = nullptr, *rowpart = nullptr, *colpart = nullptr, *sc = nullptr, *r = nullptr, *p = nullptr, *t = nullptr,
           *q = nullptr, *vS = nullptr, *part = nullptr, *qS = nullptr, *sS = nullptr, *xS = nullptr, *part_pt = nullptr, *part_rr = nullptr;
    XTile *tiles = nullptr; int2 *wrange = nullptr; XItem *items = nullptr; unsigned long long *d_cnt = nullptr; XCtrl *ctrl = nullptr;
    SNodes SN{};
    double *xbuf = nullptr;
    auto assemble = [&]() -> int {
        if (g_fault_phase == 1) { g_fault_phase = 0; return dkmc_fail(90, "injected fault (assembly of X)", __FILE__, __LINE__); }
        if (sharded) { xbuf = (double *)scratch(S_CG_XCHG, ((size_t)ns * (e.x_block > 1 ? 16 : 1) + 2) * (e.x_block > 1 ? comm_nranks() : 1) * 8); if (!xbuf) return e.err_code; if (ns > 0) { rc = xt_side_init(); if (rc) return rc; } }
        // ---- sparse part Xs: neighbour pattern of every row + values (same kernels as the CSR path, all rows) ----
        cnt = (int *)scratch(S_X_CNT, (size_t)(Nsub + 4) * 4);
        rp = (xrp_t *)scratch(S_X_ROWPTR, (size_t)(Nsub + 4) * sizeof(xrp_t));
        dpos = (xrp_t *)scratch(S_XT_DPOS, (size_t)(Nsub + 4) * sizeof(xrp_t));
        nsrank = (int *)scratch(S_X_SCB, (size_t)(Nsub + 4) * 4);
        if (!cnt || !rp || !dpos || !nsrank) return e.err_code;
        hipLaunchKernelGGL((k_xpat_plain<0>), dim3(nbr), dim3(256), 0, st, P, (const int *)nullptr, aneigh, ancnt, cnt, (const xrp_t *)nullptr, (int *)nullptr);
        rc = dkmc_exclusive_scan_i32_i64(cnt, rp, Nsub, rp + Nsub); if (rc) return rc;
        HIPCHK(hipMemcpyAsync(&xs_nnz, rp + Nsub, sizeof(long long), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (xs_nnz <= 0) return dkmc_fail(10, "update_power: empty X", __FILE__, __LINE__);
        X.xs_nnz = xs_nnz;
        col = (int *)scratch(S_X_COL, (size_t)xs_nnz * 4);
        val = (double *)scratch(S_X_DATA, (size_t)xs_nnz * 8);
        if (!col || !val) return e.err_code;
        hipLaunchKernelGGL((k_xpat_plain<1>), dim3(nbr), dim3(256), 0, st, P, (const int *)nullptr, aneigh, ancnt, cnt, (const xrp_t *)rp, col);
        hipLaunchKernelGGL((k_xval<16>), dim3((Nsub + 15) / 16), dim3(256), 0, st, P, Nsub, (const SEntry *)nullptr, 0, (const int *)nullptr, (const xrp_t *)rp,
                           (const int *)col, (const double *)buf->atom_x, (const double *)buf->atom_y, (const double *)buf->atom_z, aflag,
                           (const double *)buf->atom_CB_edge, val, TCacheView{}, atom_site, dpos);       // (neighbour entries only: no tunnelling integral)
        hipLaunchKernelGGL(k_xt_node_srank, dim3(nbr), dim3(256), 0, st, Nsub, srank, nsrank);
        KCHK();
        g_xb.rp = rp; g_xb.dpos = dpos; g_xb.ci = col; g_xb.val = val; g_xb.nsrank = nsrank;
        g_xb.ax = buf->atom_x; g_xb.ay = buf->atom_y; g_xb.az = buf->atom_z;

        // ---- S in solver order ----
        sd = (double *)scratch(S_XT_SNODE_D, (size_t)ns_pad * 4 * 8);
        si = (int *)scratch(S_XT_SNODE_I, (size_t)ns_pad * 5 * 4);
        if (!sd || !si) return e.err_code;
        SN.x = sd; SN.y = sd + ns_pad; SN.z = sd + 2 * (size_t)ns_pad; SN.cb = sd + 3 * (size_t)ns_pad;
        SN.flag = si; SN.slot = si + ns_pad; SN.mr = si + 2 * (size_t)ns_pad; SN.slotA = si + 4 * (size_t)ns_pad;
        srow = si + 3 * (size_t)ns_pad;
        hipLaunchKernelGGL(k_xt_snodes, dim3((ns_pad + 255) / 256), dim3(256), 0, st, ns, ns_pad, S, (const double *)buf->atom_x, (const double *)buf->atom_y,
                           (const double *)buf->atom_z, sd, sd + ns_pad, sd + 2 * (size_t)ns_pad, sd + 3 * (size_t)ns_pad, si, srow);
        g_xb.S = SN; g_xb.srow = srow;

        // ---- census -> tile list -> work items ----
        X.ntiles = 0; X.nitems = 0; X.nsub_total = 0; X.t_upper = 0; X.kc = 1;
        unsigned *cmask = nullptr; int *is_tile = nullptr, *nsubc = nullptr, *toff = nullptr, *soff = nullptr;
        if (ncell > 0) {
            cmask = (unsigned *)scratch(S_XT_CMASK, (size_t)(ncell + 4) * 4);
            is_tile = (int *)scratch(S_XT_ISTILE, (size_t)(ncell + 4) * 4);
            nsubc = (int *)scratch(S_XT_NSUBC, (size_t)(ncell + 4) * 4);
            toff = (int *)scratch(S_XT_TOFF, (size_t)(ncell + 4) * 4);
            soff = (int *)scratch(S_XT_SOFF, (size_t)(ncell + 4) * 4);
            if (!cmask || !is_tile || !nsubc || !toff || !soff) return e.err_code;
            hipLaunchKernelGGL(k_xt_census, dim3((unsigned)((ncell + 3) / 4)), dim3(XT_NT), 0, st, P, ns, nK, nW, SN, cmask);
            hipLaunchKernelGGL(k_xt_cell_counts, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, st, ncell, (const unsigned *)cmask, is_tile, nsubc);
            rc = dkmc_exclusive_scan_i32(is_tile, toff, (int)ncell, toff + ncell); if (rc) return rc;
            rc = dkmc_exclusive_scan_i32(nsubc, soff, (int)ncell, soff + ncell); if (rc) return rc;
            int h2[2] = {0, 0};
            HIPCHK(hipMemcpyAsync(&h2[0], toff + ncell, sizeof(int), hipMemcpyDeviceToHost, st));
            HIPCHK(hipMemcpyAsync(&h2[1], soff + ncell, sizeof(int), hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            X.ntiles = h2[0]; X.nsub_total = h2[1];
            if (h2[1] < 0) return dkmc_fail(47, "update_power: more than 2^31 sub-blocks", __FILE__, __LINE__);
        }
        ntiles = X.ntiles;
        // tiles per work item: ~4 k items per GPU.  Measured at 234 975 sites (19 372 tiles): 2 / 4 / 8 / 16 tiles per item -> 198 / 188 / 206 /
        // 222 us per launch (more items: ~2 us of start-up chain per wave round; fewer: the last waves stream alone, latency-bound)
        X.kc = std::max(1, std::min(XT_MAXKC, ntiles / nr / 4096));
        // the block-CG's tile kernel runs ONE workgroup of four waves per CU: a small system (85 k sites: 1 868 tiles) is one wave round of
        // two-tile runs (234 workgroups) instead of 1.9 rounds of single tiles (36 -> 2x us per sweep)
        if (e.x_block > 1 && ntiles <= 2048 * nr) X.kc = std::max(1, std::min(XT_MAXKC, (ntiles + 1024 * nr - 1) / (1024 * nr)));
        tiles = (XTile *)scratch(S_XT_TILES, (size_t)(ntiles + 1) * sizeof(XTile));
        wrange = (int2 *)scratch(S_XT_WRANGE, (size_t)(nK + 4) * sizeof(int2));
        if (!tiles || !wrange) return e.err_code;
        if (ncell > 0) {
            hipLaunchKernelGGL(k_xt_tile_list, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, st, nK, ncell, (const unsigned *)cmask, (const int *)toff, (const int *)soff, tiles);
            hipLaunchKernelGGL(k_xt_wrange, dim3((nK + 255) / 256), dim3(256), 0, st, nK, nW, (const unsigned *)cmask, wrange);
        }
        XShare sh{};
        if (e.x_items_kc > 0) X.kc = e.x_items_kc;
        rc = xt_build_items(nK, nW, X.kc, ntiles, X.nsub_total, toff, tiles, nr, me, S_XT_NITEMW, S_XT_ITEMS, S_XT_SPLIT, &sh, 2); if (rc) return rc;
        if ((sh.item_lo | sh.item_n) & 3) return dkmc_fail(48, "update_power: run list not padded to groups of four", __FILE__, __LINE__);
        X.nitems = sh.nitems; X.maxchunk = sh.maxchunk; X.rec_shift = sh.rec_shift;
        items = sh.items; nitem_w = sh.nitem_w;
        KCHK();
        g_xb.tiles = tiles; g_xb.items = items; g_xb.wrange = wrange; g_xb.nitem_w = nitem_w; g_xb.cmask = cmask; g_xb.toff = toff;

        // ---- this rank's share of the work items (contiguous in the strip-major tile list, balanced by stored bytes) ----
        X.item_lo = sh.item_lo; X.item_n = sh.item_n; X.tile_lo = sh.tile_lo; X.tile_n = sh.tile_n; X.sub_base = sh.sub_base; X.sub_n = sh.sub_n;
        X.w_lo = sh.w_lo; X.w_hi = sh.w_hi;
        e.stats.comm_ranks = sharded ? comm_nranks() : 0;
        e.stats.comm_count_per_rank = sharded ? ns + 2 : 0;
        e.stats.comm_local_segments = X.item_n;

        // ---- coefficient cache, brought up to date for what THIS rank's tiles read (its column windows; everything on one GPU) ----
        rc = tc_prepare_tiled(P, buf, ns, S, aflag, srank, atom_site, sharded ? XT_C * X.w_lo : 0, sharded ? XT_C * X.w_hi : 0x7fffffff, sharded ? 1 : 0, &TC);
        if (rc) return rc;
        hipLaunchKernelGGL(k_xt_snode_slots, dim3((ns_pad + 255) / 256), dim3(256), 0, st, ns, ns_pad, S, atom_site, TC, si + ns_pad, si + 4 * (size_t)ns_pad, si + 2 * (size_t)ns_pad);

        // ---- storage + fill ----
        tval = (double *)scratch(S_XT_TVAL, (size_t)(X.sub_n + 4) * XT_SUB * 8);       // (slack: the block-CG product requests two sub-blocks beyond a full tile)
        rowpart = (double *)scratch(S_XT_ROWPART, (size_t)(ncell + 1) * XT_R * 8);
        colpart = (double *)scratch(S_XT_COLPART, (size_t)(X.nitems + 1) * XT_C * 8);       // one 256-entry record per run (32 MB at 9.4e5 sites; [run position][S rank] took 1.8 GB)
        d_cnt = (unsigned long long *)scratch(S_XT_CNT, 16);
        if (!tval || !rowpart || !colpart || !d_cnt) return e.err_code;
        HIPCHK(hipMemsetAsync(rowpart, 0, (size_t)(ncell + 1) * XT_R * 8, st));
        HIPCHK(hipMemsetAsync(colpart, 0, (size_t)(X.nitems + 1) * XT_C * 8, st));
        HIPCHK(hipMemsetAsync(d_cnt, 0, 16, st));
        if (X.tile_n > 0)
            hipLaunchKernelGGL(k_xt_fill, dim3(X.tile_n), dim3(XT_NT), 0, st, P, ns, (const XTile *)tiles + X.tile_lo, (int)X.sub_base, SN, TC, tval, d_cnt);
        KCHK();
        g_xb.tval = tval; g_xb.rowpart = rowpart; g_xb.colpart = colpart;
        X.valid = true;

        // ---- vectors ----
        sc = (double *)scratch(S_CG_S, (size_t)m * 8); r = (double *)scratch(S_CG_R, (size_t)m * 8);
        p = (double *)scratch(S_CG_P, (size_t)m * 8); t = (double *)scratch(S_CG_T, (size_t)m * 8);
        q = (double *)scratch(S_XT_Q, (size_t)m * 8);
        vS = (double *)scratch(S_CG_PS, (size_t)ns_pad * 3 * 8);       // qS | sS | scratch of the tile-sum passes (padding stays zero)
        part = (double *)scratch(S_CG_PART, (size_t)3 * 8192 * 8);
        ctrl = (XCtrl *)scratch(S_CG_CTRL, sizeof(XCtrl));
        if (!sc || !r || !p || !t || !q || !vS || !part || !ctrl) return e.err_code;
        qS = vS; sS = vS + ns_pad; xS = vS + 2 * (size_t)ns_pad;
        HIPCHK(hipMemsetAsync(vS, 0, (size_t)ns_pad * 3 * 8, st));
        part_pt = part; part_rr = part + 4096;          // p.t | r.t | t.t partials (XT_PSTRIDE apart); r.r partials, double-buffered (512 apart)
        return 0;
    };
    rc = assemble();
    if (sharded) rc = comm_agree(rc, "assembly of X");
    if (rc) return rc;

    const double tol2 = e.cg_tol * e.cg_tol;

    // ---- diagonal: -(row sums of T), one pass with the vector of ones ----
    if (ns > 0) {
        hipLaunchKernelGGL(k_xt_fill_f64, dim3((ns + 255) / 256), dim3(256), 0, st, qS, (long long)ns, 1.0);
        rc = xt_tile_sums<0>(qS, xS, 1); if (rc) return rc;
    }
    hipLaunchKernelGGL(k_xt_scale_init, dim3(nbr), dim3(256), 0, st, m, (const xrp_t *)dpos, val, (const int *)nsrank, (const double *)xS, sc, rhs, y, q, sS, qS);
    KCHK();

    // ---- launch shapes ----
    const int ntb = (X.item_n + 3) / 4;
    const int nsb = (std::max(m - 2, 1) + XT_NT / 8 * XT_RPG_FUSED - 1) / (XT_NT / 8 * XT_RPG_FUSED);     // neighbour role inside k_xt_apply
    const int nsb1 = (std::max(m - 2, 1) + XT_NT / 8 - 1) / (XT_NT / 8);                                    // k_xt_neigh
    const int n2b = xt_grid(std::max(std::max(nK, (m + 4095) / 4096), 1), 1, 1024);    // row kernel: S row blocks + its share of the p.t dot
    const int gv = xt_grid(m, XT_NT * 4, 256);
    const int np_pt = n2b;
    const bool nt_loads = (size_t)X.sub_n * XT_SUB * 8 > ((size_t)200 << 20);

    const bool split = sharded && ns > 0;                 // tile pass + exchange on the engine's stream, neighbour part on the side stream
    const bool seq_neigh = !split && (size_t)X.sub_n * XT_SUB * 8 > ((size_t)2 << 30);
    e.stats.xt_split_launch = (split || seq_neigh) ? 1 : 0;  // the timed apply launch carries the tiles only
    int cur_it = -1, local_fail = 0;        // iteration being enqueued; first host-side failure of this rank inside the loop
    auto matvec = [&](hipEvent_t e0, hipEvent_t e1, hipEvent_t e2, hipEvent_t e3, hipEvent_t ec) -> int {
#define XT_APPLY_ARGS(NI, NTB, NSB) NI, (const XItem *)items + X.item_lo, (const XTile *)tiles, (int)X.sub_base, (const double *)tval, (const double *)qS, nW, ns_pad, \
                      rowpart, colpart, (const XCtrl *)ctrl, NTB, NSB, m, (const xrp_t *)rp, (const int *)col, (const double *)val, (const double *)q, \
                      (const double *)sc, (const int *)nsrank, t
        if (split) {
            XSide &S = g_side; const int sl = (int)(S.seq++ % XT_SIDE_RING);
            // Host-side failures between two collectives must not leave the peers in the all-reduce: the local part runs in `local`;
            // if it fails, this rank still joins the all-reduce, with the abort word set, and every rank leaves the loop together.
            auto local = [&]() -> int {
            if (g_fault_phase == 2 && cur_it >= g_fault_iter) { g_fault_phase = 0; return dkmc_fail(91, "injected fault (CG iteration)", __FILE__, __LINE__); }
            if (ntb > 0) {
                if (nt_loads) hipExtLaunchKernelGGL((k_xt_apply<1>), dim3(ntb), dim3(XT_NT), 0, st, e0, e1, 0, XT_APPLY_ARGS(X.item_n, ntb, 0), 2);
                else hipExtLaunchKernelGGL((k_xt_apply<0>), dim3(ntb), dim3(XT_NT), 0, st, e0, e1, 0, XT_APPLY_ARGS(X.item_n, ntb, 0), 2);
            } else if (e0) { HIPCHK(hipEventRecord(e0, st)); HIPCHK(hipEventRecord(e1, st)); }     // a rank without tiles: the sampled interval is empty
            // the neighbour part starts when the tile pass has drained (it would only share the HBM stream with it before) and runs
            // beside the partial row sums and the exchange; q, the stop flag and the previous readers of t are behind this event too
            HIPCHK(hipEventRecord(S.a[sl], st));
            HIPCHK(hipStreamWaitEvent(S.st, S.a[sl], 0));
            hipLaunchKernelGGL(k_xt_neigh, dim3(nsb1 + 2), dim3(XT_NT), 0, S.st, nsb1, m, (const xrp_t *)rp, (const int *)col, (const double *)val, (const double *)q,
                               (const double *)sc, (const int *)nsrank, (const XCtrl *)ctrl, t);
            HIPCHK(hipEventRecord(S.b[sl], S.st));
            hipLaunchKernelGGL((k_xt_rows<1>), dim3(std::max(nK, 1)), dim3(XT_NT), 0, st, ns, nK, nW, ns_pad, (const int2 *)wrange, (const int *)nitem_w, (const double *)rowpart,
                               (const double *)colpart, (const int *)srow, (const double *)sS, (const double *)p, t, part_pt, (const XCtrl *)ctrl, xbuf,
                               m, (const int *)nsrank, comm_rank() == 0 ? 1 : 0, (const double *)r, X.w_lo, X.w_hi);   // no partial arrays here: one workgroup per row block
            KCHK();
            return 0;
            };
            if (int lrc = local()) {
                if (!local_fail) local_fail = lrc;
                (void)hipGetLastError();
                hipLaunchKernelGGL(k_xt_abort_word, dim3(1), dim3(1), 0, st, ctrl, xbuf, ns);
            }
            if (int rcx = comm_allreduce_sum_f64(xbuf, (size_t)ns + 2)) return rcx;          // |S| row sums + rank 0's stop decision + the abort word
            if (ec) HIPCHK(hipEventRecord(ec, st));
            HIPCHK(hipStreamWaitEvent(st, S.b[sl], 0));                                       // the neighbour sums are in t
            hipExtLaunchKernelGGL(k_xt_rows_apply, dim3(n2b), dim3(XT_NT), 0, st, e2, e3, 0, ns, nK, (const double *)xbuf, (const int *)srow, (const double *)sS,
                                  (const double *)p, t, part_pt, ctrl, m, (const int *)nsrank, (const double *)r);
            return 0;
        }
        if (seq_neigh) {
            // multi-GB sweeps: the neighbour rows leave the tile launch altogether (there they are a 65-190 us tail at 2 waves per SIMD);
            // tile pass, then the neighbour part as its own full-occupancy kernel (38-42 us at 9.4e5 sites): 13.82 / 13.49 -> 13.39 s per
            // step at 9.4e5 sites; even at 2.35e5 sites (1 GB), a loss at 1.5e5 (0.4 GB), where the extra launch costs more than the tail
            if (ntb > 0) hipExtLaunchKernelGGL((k_xt_apply<1>), dim3(ntb), dim3(XT_NT), 0, st, e0, e1, 0, XT_APPLY_ARGS(X.item_n, ntb, 0), 2);
            else if (e0) { HIPCHK(hipEventRecord(e0, st)); HIPCHK(hipEventRecord(e1, st)); }
            hipLaunchKernelGGL(k_xt_neigh, dim3(nsb1 + 2), dim3(XT_NT), 0, st, nsb1, m, (const xrp_t *)rp, (const int *)col, (const double *)val, (const double *)q,
                               (const double *)sc, (const int *)nsrank, (const XCtrl *)ctrl, t);
        } else if (nt_loads) hipExtLaunchKernelGGL((k_xt_apply<1>), dim3(ntb + nsb + 2), dim3(XT_NT), 0, st, e0, e1, 0, XT_APPLY_ARGS(X.item_n, ntb, nsb), 0);
        else hipExtLaunchKernelGGL((k_xt_apply<0>), dim3(ntb + nsb + 2), dim3(XT_NT), 0, st, e0, e1, 0, XT_APPLY_ARGS(X.item_n, ntb, nsb), 0);
#undef XT_APPLY_ARGS
        hipExtLaunchKernelGGL((k_xt_rows<0>), dim3(n2b), dim3(XT_NT), 0, st, e2, e3, 0, ns, nK, nW, ns_pad, (const int2 *)wrange, (const int *)nitem_w,
                              (const double *)rowpart, (const double *)colpart, (const int *)srow, (const double *)sS, (const double *)p, t,
                              part_pt, (const XCtrl *)ctrl, (double *)nullptr, m, (const int *)nsrank, 0, (const double *)r, 0, nW);
        return 0;
    };

    // ---- block-CG (dkmc_set_x_block(s > 1), one GPU): xtb.hip.  Column 0 = this system; on a loss of definiteness of its s x s systems the
    // single-vector loop below continues from the last good iterate ----
    XCtrl h{};
    bool solved = false;
    double prof_long_ms = 0.0, prof_short_ms = 0.0, prof_comm_ms = 0.0; int prof_long_n = 0, prof_short_n = 0, prof_comm_n = 0;
    e.stats.xb_width = 1; e.stats.xb_fallback = 0;
    if (e.x_block > 1 && ns > 0) {
        XtbArgs B{};
        B.m = m; B.ns = ns; B.ns_pad = ns_pad; B.nK = nK; B.nW = nW; B.s = e.x_block;
        B.items = (const XItem *)items + X.item_lo; B.item_n = X.item_n; B.tiles = tiles; B.sub_base = (int)X.sub_base; B.tval = tval;
        B.wrange = wrange; B.nitem_w = nitem_w; B.nrecords = X.nitems >> X.rec_shift;
        B.srow = srow; B.sS = sS; B.nsrank = nsrank; B.rp = rp; B.ci = col; B.val = val; B.sc = sc; B.ax = buf->atom_x; B.ay = buf->atom_y; B.az = buf->atom_z; B.b = rhs; B.y = y;
        B.yaux = yaux; B.yaux_valid = yaux_valid;
        B.ctrl = ctrl; B.tol2 = tol2; B.nt_loads = nt_loads; B.sharded = sharded; B.w_lo = X.w_lo; B.w_hi = X.w_hi;
        int bi = 0; double brr = 0.0;
        rc = xtb_cg(B, &bi, &brr);
        e.stats.xb_width = e.x_block;
        if (rc == 0) { solved = true; h.iters = bi; h.rr[bi & 1] = brr; if (sharded && !(e.x_slab && comm_nranks() > 1)) e.stats.comm_count_per_rank = (long long)ns * (4 * ((e.x_block + 3) / 4)) + 2; }      // (slab-distributed loop: set there)
        else if (rc != DKMC_XTB_BREAKDOWN) return rc;
        else {
            e.stats.xb_fallback = 1;
            hipLaunchKernelGGL(k_xt_requeue, dim3(nbr), dim3(256), 0, st, m, (const double *)y, (const double *)sc, (const int *)nsrank, q, qS);
        }
    }
    const bool prof = e.profiling != 0;
    if (!solved) {
    // ---- r = A y - b, p = -r ----
    HIPCHK(hipMemsetAsync(ctrl, 0, sizeof(XCtrl), st));
    if (sharded && ns > 0) hipLaunchKernelGGL(k_xt_set_sharded, dim3(1), dim3(1), 0, st, ctrl);
    HIPCHK(hipMemsetAsync(p, 0, (size_t)m * 8, st));
    rc = matvec(nullptr, nullptr, nullptr, nullptr, nullptr); if (rc) return rc;
    hipLaunchKernelGGL(k_xt_resid_init, dim3(gv), dim3(XT_NT), 0, st, m, (const double *)t, (const double *)rhs, r, p, (const double *)sc, q, (const int *)nsrank, qS, part_rr);
    hipLaunchKernelGGL(k_xt_check0, dim3(1), dim3(XT_NT), 0, st, (const double *)part_rr, gv, ctrl, tol2);
    KCHK();

    // ---- optional kernel profile: start/stop events of sampled apply launches (bench.py roofline) ----
    static hipEvent_t evs[4 * 64], evc[64 / XT_PROF_STRIDE]; static bool evs_ready = false;
    if (prof && !evs_ready) { for (auto &ev : evs) HIPCHK(hipEventCreate(&ev)); for (auto &ev : evc) HIPCHK(hipEventCreate(&ev)); evs_ready = true; }

    // ---- iterations in batches; the host polls the control block between batches (batch plan: see cg.hip) ----
    int it = 0, launched = 0, batch = 8;
    if (e.x_iter_hint > 24) batch = e.x_iter_hint - 8;
    int loop_rc = 0;
    for (;;) {
        HIPCHK(hipMemcpyAsync(&h, ctrl, sizeof(XCtrl), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (prof && launched) {
            for (int b = 0; b < launched && b < 64; b += XT_PROF_STRIDE) {
                if (it - launched + b >= h.iters) break;
                float ms = 0.f;
                HIPCHK(hipEventElapsedTime(&ms, evs[4 * b], evs[4 * b + 1])); prof_long_ms += ms; ++prof_long_n;
                HIPCHK(hipEventElapsedTime(&ms, evs[4 * b + 2], evs[4 * b + 3])); prof_short_ms += ms; ++prof_short_n;
                if (sharded && ns > 0) { HIPCHK(hipEventElapsedTime(&ms, evs[4 * b + 1], evc[b / XT_PROF_STRIDE])); prof_comm_ms += ms; ++prof_comm_n; }
            }
        }
        if (h.done || loop_rc) break;
        if (it >= 200000) { dkmc_fail(4, "CG: no convergence after 200000 iterations", __FILE__, __LINE__); break; }
        for (int b = 0; b < batch; ++b, ++it) {
            cur_it = it;
            const bool pb = prof && b < 64 && (b % XT_PROF_STRIDE == 0);
            loop_rc = matvec(pb ? evs[4 * b] : nullptr, pb ? evs[4 * b + 1] : nullptr, pb ? evs[4 * b + 2] : nullptr, pb ? evs[4 * b + 3] : nullptr,
                             pb ? evc[b / XT_PROF_STRIDE] : nullptr);
            if (loop_rc) break;
            hipLaunchKernelGGL(k_xt_step, dim3(gv), dim3(XT_NT), 0, st, m, it, (const double *)part_pt, np_pt, (const double *)(part_rr + 512 * (it & 1)), gv,
                               part_rr + 512 * ((it + 1) & 1), p, (const double *)t, y, r, (const double *)sc, q, (const int *)nsrank, qS, ctrl, tol2);
        }
        launched = batch;
        KCHK();
        if (e.x_iter_hint > 24) batch = 8; else if (batch < 64) batch *= 2;
    }
    if (loop_rc) return loop_rc;
    if (local_fail) return local_fail;                                                     // this rank failed between two collectives: the peers were told (abort word)
    if (h.aborted) return dkmc_fail(46, "a peer rank aborted the sharded current solve", __FILE__, __LINE__);
    }
    hipLaunchKernelGGL(k_xt_vec_mul, dim3(nbr), dim3(256), 0, st, m, y, (const double *)sc);
    KCHK();
    // the solution every later phase starts from: rank 0's bits (the block-CG adds the ranks' slots in rank order on every rank: same bits already)
    if (sharded && !solved) { rc = comm_bcast0_f64(y, (size_t)m); if (rc) return rc; }
    HIPCHK(hipMemcpyAsync(&X.t_upper, d_cnt, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    e.x_iter_hint = h.iters;
    if (iters_out) *iters_out = h.iters;
    if (rr_out) *rr_out = h.rr[h.iters & 1];
    // ---- statistics ----
    long long t_upper = (long long)X.t_upper;
    if (sharded) {              // each rank counted its own tiles
        double *cntbuf = (double *)scratch(S_XT_CNT, 16);
        double hv = (double)t_upper;
        HIPCHK(hipMemcpy(cntbuf, &hv, 8, hipMemcpyHostToDevice));
        if (int rcx = comm_allreduce_sum_f64(cntbuf, 1)) return rcx;
        HIPCHK(hipStreamSynchronize(st));
        HIPCHK(hipMemcpy(&hv, cntbuf, 8, hipMemcpyDeviceToHost));
        t_upper = (long long)(hv + 0.5);
    }
    e.stats.X_nnz = xs_nnz + 2 * t_upper;
    e.stats.spmv_tiles = ntiles; e.stats.spmv_tile_entries = t_upper;
    e.stats.xt_subblocks = X.nsub_total; e.stats.xt_local_subblocks = X.sub_n; e.stats.xt_items = X.nitems; e.stats.xt_kc = X.kc; e.stats.xt_records = X.nitems >> X.rec_shift;
    e.stats.xt_sparse_nnz = xs_nnz; e.stats.xt_ns = ns;
    e.stats.spmv_segments = 0; e.stats.spmv_segment_entries = 0;
    e.stats.spmv_long_rows = ns; e.stats.spmv_short_rows = m - ns; e.stats.spmv_long_nnz = 2 * t_upper; e.stats.spmv_short_nnz = xs_nnz;
    if (prof && !solved) {
        e.stats.spmv_long_ms = prof_long_ms; e.stats.spmv_short_ms = prof_short_ms;
        e.stats.spmv_long_launches = prof_long_n; e.stats.spmv_short_launches = prof_short_n;
        e.stats.comm_ms = prof_comm_ms; e.stats.comm_launches = prof_comm_n;
    }
    return e.err_code;
}

// dissipated power from the solved (G0-scaled, shifted) node potentials m
int xt_power(dkmc_gpubuf *buf, const XParams &P, const int *aflag, const int *atom_site, const double *m, double Vd, double alpha)
{
    Engine &e = eng(); hipStream_t st = e.stream; const XTState &X = g_xt;
    if (!X.valid) return dkmc_fail(13, "xt_power: no assembled X", __FILE__, __LINE__);
    double *vS = (double *)e.buf[S_CG_PS];
    double *mS = vS, *pS = vS + 2 * (size_t)X.ns_pad;
    if (X.ns > 0) {
        hipLaunchKernelGGL(k_xt_gather_S, dim3((X.ns + 255) / 256), dim3(256), 0, st, X.ns, (const int *)g_xb.srow, m, mS);
        int rc = xt_tile_sums<1>(mS, pS, Vd > 0 ? 1 : 0); if (rc) return rc;
    }
    hipLaunchKernelGGL(k_xt_power_rows, dim3((X.Nsub + 31) / 32), dim3(XT_NT), 0, st, X.Nsub, (const xrp_t *)g_xb.rp, (const int *)g_xb.ci, (const double *)g_xb.val,
                       m, Vd, (const int *)g_xb.nsrank, (const double *)pS, aflag, atom_site, alpha, buf->site_power);
    KCHK();
    (void)P;
    return 0;
}

// ---- measurement aid (bench.py: strong-scaling model; no counterpart in the reference) -------------------------------------------
// Times, on the X left resident by the last single-GPU solve, what ONE rank of an nranks-way sharded solve would run per CG
// iteration: the apply kernel over that rank's share of the work items (built with the work-item size an nranks run uses) plus the
// neighbour part, and the three kernels around the exchange (partial row sums, finish, vector step).  The exchange itself (one
// all-reduce of |S| + 1 doubles) cannot be measured on one GPU.  Vectors are scratch: the numbers computed here are discarded.
// Measurement aid: how the resident X fills its tiles -- hist[c] = tiles with c of their 8 sub-blocks present (c = 0 .. 8), hist[9] = tiles that sit in a
// chain of full tiles of length >= 2 inside a run (what the product kernel streams with sub-blocks in flight), hist[10] = runs
extern "C" int dkmc_xt_tile_census(long long *hist)
{
    Engine &e = eng(); const XTState &X = g_xt;
    if (!X.valid || X.ntiles <= 0 || !hist) return dkmc_fail(13, "xt_tile_census: needs the X of a solve", __FILE__, __LINE__);
    std::vector<XTile> t((size_t)X.ntiles);
    HIPCHK(hipStreamSynchronize(e.stream));
    HIPCHK(hipMemcpy(t.data(), g_xb.tiles, (size_t)X.ntiles * sizeof(XTile), hipMemcpyDeviceToHost));
    for (int c = 0; c < 11; ++c) hist[c] = 0;
    for (const XTile &x : t) hist[__builtin_popcount(x.mask & 0xffu)]++;
    std::vector<XItem> it((size_t)X.nitems);
    HIPCHK(hipMemcpy(it.data(), g_xb.items, (size_t)X.nitems * sizeof(XItem), hipMemcpyDeviceToHost));
    for (const XItem &r : it) {
        if (r.t1 <= r.t0) continue;
        hist[10]++;
        int run = 0;
        for (int q = r.t0; q <= r.t1; ++q) {
            const bool full = q < r.t1 && (t[(size_t)q].mask & 0xffu) == 0xffu;
            if (full) ++run; else { if (run >= 2) hist[9] += run; run = 0; }
        }
    }
    return 0;
}
extern "C" int dkmc_xt_time_share(int nranks, int rank, int reps, double *apply_us, double *side_us, int *items_out, long long *subblocks_out)
{
    Engine &e = eng(); hipStream_t st = e.stream; const XTState &X = g_xt;
    if (!X.valid || comm_attached() || X.nitems <= 0 || X.tile_n != X.ntiles) return dkmc_fail(13, "xt_time_share: needs the X of a single-GPU solve", __FILE__, __LINE__);
    if (nranks < 1 || rank < 0 || rank >= nranks || reps < 1) return dkmc_fail(13, "xt_time_share: bad arguments", __FILE__, __LINE__);
    const int nK = X.nK, nW = X.nW, ns = X.ns, ns_pad = X.ns_pad, m = X.Nsub, ntiles = X.ntiles;
    const int kc = std::max(1, std::min(XT_MAXKC, ntiles / nranks / 4096));
    XShare sh{};
    int rc = xt_build_items(nK, nW, kc, ntiles, X.nsub_total, (const int *)g_xb.toff, (const XTile *)g_xb.tiles, nranks, rank,
                            S_XT_T_NITEMW, S_XT_T_ITEMS, S_XT_T_SPLIT, &sh, 2);
    if (!rc && ((sh.item_lo | sh.item_n) & 3)) rc = dkmc_fail(48, "xt_time_share: run list not padded to groups of four", __FILE__, __LINE__);
    if (rc) return rc;
    const int item_n = sh.item_n, i0 = sh.item_lo;
    const XItem *items = sh.items; const int *nitem_w = sh.nitem_w;
    double *colpart = (double *)scratch(S_XT_T_COLPART, (size_t)(sh.nitems + 1) * XT_C * 8);
    if (!colpart) return e.err_code;
    HIPCHK(hipMemsetAsync(colpart, 0, (size_t)(sh.nitems + 1) * XT_C * 8, st));
    if (items_out) *items_out = item_n;
    if (subblocks_out) *subblocks_out = sh.sub_n;
    // scratch vectors of the last solve (their contents do not matter) + a private y and control block
    double *sc = (double *)e.buf[S_CG_S], *r = (double *)e.buf[S_CG_R], *p = (double *)e.buf[S_CG_P], *t = (double *)e.buf[S_CG_T], *q = (double *)e.buf[S_XT_Q];
    double *vS = (double *)e.buf[S_CG_PS], *part = (double *)e.buf[S_CG_PART];
    double *ytmp = (double *)scratch(S_MISC3, (size_t)(m + ns + 8) * 8);
    XCtrl *ctrl = (XCtrl *)scratch(S_MISC2, 256);
    if (!ytmp || !ctrl || !sc || !r || !p || !t || !q || !vS || !part) return e.err_code;
    double *xbuf = ytmp + m;
    double *qS = vS, *sS = vS + ns_pad, *part_pt = part, *part_rr = part + 4096;
    HIPCHK(hipMemsetAsync(ctrl, 0, sizeof(XCtrl), st));
    HIPCHK(hipMemsetAsync(ytmp, 0, (size_t)(m + ns + 8) * 8, st));
    const int ntb = (item_n + 3) / 4, nsb = (std::max(m - 2, 1) + XT_NT / 8 * XT_RPG_FUSED - 1) / (XT_NT / 8 * XT_RPG_FUSED), nsb1 = (std::max(m - 2, 1) + XT_NT / 8 - 1) / (XT_NT / 8);
    const int n2b = xt_grid(std::max(std::max(nK, (m + 4095) / 4096), 1), 1, 1024), gv = xt_grid(m, XT_NT * 4, 256);
    const bool nt_loads = (size_t)sh.sub_n * XT_SUB * 8 > ((size_t)200 << 20);
    // one GPU: the whole product in one launch.  nranks > 1: the tile pass alone, as the sharded solve launches it; the neighbour part
    // (which that solve runs on a second stream beside the exchange) is timed on its own as the fourth side kernel.
    auto apply = [&](int part) {
#define XT_TS_ARGS(NI, NTB, NSB) NI, (const XItem *)items + i0, (const XTile *)g_xb.tiles, 0, (const double *)g_xb.tval, (const double *)qS, nW, ns_pad, \
                   g_xb.rowpart, colpart, (const XCtrl *)ctrl, NTB, NSB, m, (const xrp_t *)g_xb.rp, (const int *)g_xb.ci, (const double *)g_xb.val, \
                   (const double *)q, (const double *)sc, (const int *)g_xb.nsrank, t
        if (part == 2) hipLaunchKernelGGL(k_xt_neigh, dim3(nsb1 + 2), dim3(XT_NT), 0, st, nsb1, m, (const xrp_t *)g_xb.rp, (const int *)g_xb.ci, (const double *)g_xb.val,
                                          (const double *)q, (const double *)sc, (const int *)g_xb.nsrank, (const XCtrl *)ctrl, t);
        else if (part == 1) {
            if (ntb <= 0) return;
            if (nt_loads) hipLaunchKernelGGL((k_xt_apply<1>), dim3(ntb), dim3(XT_NT), 0, st, XT_TS_ARGS(item_n, ntb, 0), 2);
            else hipLaunchKernelGGL((k_xt_apply<0>), dim3(ntb), dim3(XT_NT), 0, st, XT_TS_ARGS(item_n, ntb, 0), 2);
        } else {
            if (nt_loads) hipLaunchKernelGGL((k_xt_apply<1>), dim3(ntb + nsb + 2), dim3(XT_NT), 0, st, XT_TS_ARGS(item_n, ntb, nsb), 0);
            else hipLaunchKernelGGL((k_xt_apply<0>), dim3(ntb + nsb + 2), dim3(XT_NT), 0, st, XT_TS_ARGS(item_n, ntb, nsb), 0);
        }
#undef XT_TS_ARGS
    };
    const bool seq_neigh = (size_t)sh.sub_n * XT_SUB * 8 > ((size_t)2 << 30);     // one GPU, multi-GB sweep: the solve runs tile pass + k_xt_neigh too
    const int apply_part = (nranks > 1 || seq_neigh) ? 1 : 0;
    auto side = [&](int which, int it) {
        if (which == 0)
            hipLaunchKernelGGL((k_xt_rows<1>), dim3(std::max(nK, 1)), dim3(XT_NT), 0, st, ns, nK, nW, ns_pad, (const int2 *)g_xb.wrange, (const int *)nitem_w, (const double *)g_xb.rowpart,
                               (const double *)colpart, (const int *)g_xb.srow, (const double *)sS, (const double *)p, t, part_pt, (const XCtrl *)ctrl, xbuf,
                               m, (const int *)g_xb.nsrank, 0, (const double *)r, sh.w_lo, sh.w_hi);
        else if (which == 1)
            hipLaunchKernelGGL(k_xt_rows_apply, dim3(n2b), dim3(XT_NT), 0, st, ns, nK, (const double *)xbuf, (const int *)g_xb.srow, (const double *)sS,
                               (const double *)p, t, part_pt, ctrl, m, (const int *)g_xb.nsrank, (const double *)r);
        else
            hipLaunchKernelGGL(k_xt_step, dim3(gv), dim3(XT_NT), 0, st, m, it, (const double *)part_pt, n2b, (const double *)(part_rr + 512 * (it & 1)), gv,
                               part_rr + 512 * ((it + 1) & 1), p, (const double *)t, ytmp, r, (const double *)sc, q, (const int *)g_xb.nsrank, qS, ctrl, -1.0);
    };
    apply(apply_part); for (int w = 0; w < 3; ++w) side(w, 0);    // warm-up
    hipEvent_t ev[10];
    for (auto &x : ev) HIPCHK(hipEventCreate(&x));
    // each kernel back to back with itself: in a solve the apply kernel and the exchange sit between them, so none of the three
    // finds its operands in L2 from the previous one either
    for (int w = 0; w < 5; ++w) {
        HIPCHK(hipMemsetAsync(ctrl, 0, sizeof(XCtrl), st));
        HIPCHK(hipEventRecord(ev[2 * w], st));
        if (w < 4 || apply_part == 1)
            for (int k = 0; k < reps; ++k) { if (w == 0) apply(apply_part); else if (w == 4) apply(2); else side(w - 1, k); }
        HIPCHK(hipEventRecord(ev[2 * w + 1], st));
    }
    HIPCHK(hipEventSynchronize(ev[9]));
    for (int w = 0; w < 5; ++w) {
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, ev[2 * w], ev[2 * w + 1]));
        const double us = (double)ms * 1e3 / reps;
        if (w == 0) { if (apply_us) *apply_us = us; } else if (side_us) side_us[w - 1] = (w == 4 && apply_part == 0) ? 0.0 : us;
    }
    for (auto &x : ev) (void)hipEventDestroy(x);
    KCHK();
    // the vectors of the last solve are garbage now: the next solve rebuilds all of them
    return e.err_code;
}

// ---- test aid (tests/test_dist_sharded.py; no counterpart in the reference) -----------------------------------------------------
// Emulates, on the X left resident by the last single-GPU solve, the tile pass of an nranks-way sharded matrix-vector product: for
// every rank in turn it builds that rank's work items exactly as a sharded assembly does, runs the tile kernel over them into
// zeroed partial arrays and the partial-row-sum kernel restricted to the rank's windows, and adds the ranks' results.  Reports the
// largest deviation from the one-GPU tile pass over the same vector, and how many work items / sub-blocks the shares hold in total
// (each must be covered exactly once).
__global__ void k_xt_test_vector(int ns, double *__restrict__ v)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < ns) v[s] = 0.5 + (double)(((unsigned)s * 2654435761u) >> 16) / 65536.0;
}
__global__ void k_xt_test_accumulate(int ns, double *__restrict__ acc, const double *__restrict__ x)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < ns) acc[s] += x[s];
}
extern "C" int dkmc_xt_check_shares(int nranks, double *max_abs_diff, double *max_abs, long long *subblocks_sum, long long *items_sum, int *items_total)
{
    Engine &e = eng(); hipStream_t st = e.stream; const XTState &X = g_xt;
    if (!X.valid || comm_attached() || X.tile_n != X.ntiles) return dkmc_fail(13, "xt_check_shares: needs the X of a single-GPU solve", __FILE__, __LINE__);
    if (nranks < 1 || nranks > XT_MAXRANKS) return dkmc_fail(13, "xt_check_shares: bad arguments", __FILE__, __LINE__);
    const int nK = X.nK, nW = X.nW, ns = X.ns, ns_pad = X.ns_pad, ntiles = X.ntiles;
    if (max_abs_diff) *max_abs_diff = 0.0;
    if (max_abs) *max_abs = 0.0;
    if (subblocks_sum) *subblocks_sum = 0;
    if (items_sum) *items_sum = 0;
    if (items_total) *items_total = 0;
    if (ns <= 0 || ntiles <= 0) return 0;
    double *buf = (double *)scratch(S_XT_T_MISC, (size_t)4 * ns_pad * 8);
    if (!buf) return e.err_code;
    double *v = buf, *full = buf + ns_pad, *acc = buf + 2 * (size_t)ns_pad, *tmp = buf + 3 * (size_t)ns_pad;
    const int gb = (ns + 255) / 256;
    HIPCHK(hipMemsetAsync(buf, 0, (size_t)4 * ns_pad * 8, st));
    hipLaunchKernelGGL(k_xt_test_vector, dim3(gb), dim3(256), 0, st, ns, v);
    int rc = xt_tile_sums<0>(v, full, 0); if (rc) return rc;
    const size_t rowpart_bytes = (size_t)((long long)nK * nW + 1) * XT_R * 8;
    const int kc = std::max(1, std::min(XT_MAXKC, ntiles / nranks / 4096));
    long long sb_sum = 0, it_sum = 0;
    for (int r = 0; r < nranks; ++r) {
        XShare sh{};
        rc = xt_build_items(nK, nW, kc, ntiles, X.nsub_total, (const int *)g_xb.toff, (const XTile *)g_xb.tiles, nranks, r,
                            S_XT_T_NITEMW, S_XT_T_ITEMS, S_XT_T_SPLIT, &sh, 2);
        if (rc) return rc;
        if ((sh.item_lo | sh.item_n) & 3) return dkmc_fail(48, "xt_check_shares: run list not padded to groups of four", __FILE__, __LINE__);
        if (items_total) *items_total = sh.nitems;
        sb_sum += sh.sub_n; it_sum += sh.item_n;
        double *colpart = (double *)scratch(S_XT_T_COLPART, (size_t)(sh.nitems + 1) * XT_C * 8);
        if (!colpart) return e.err_code;
        HIPCHK(hipMemsetAsync(colpart, 0, (size_t)(sh.nitems + 1) * XT_C * 8, st));      // what a rank's assembly does: cells of tiles it does not own stay zero
        HIPCHK(hipMemsetAsync(g_xb.rowpart, 0, rowpart_bytes, st));
        if (sh.item_n > 0)
            hipLaunchKernelGGL((k_xt_tiles_only<0>), dim3((sh.item_n + 3) / 4), dim3(XT_NT), 0, st, sh.item_n, (const XItem *)sh.items + sh.item_lo,
                               (const XTile *)g_xb.tiles, 0, (const double *)g_xb.tval, (const double *)v, nW, ns_pad, g_xb.rowpart, colpart, 0);
        hipLaunchKernelGGL((k_xt_rows<1>), dim3(std::max(nK, 1)), dim3(XT_NT), 0, st, ns, nK, nW, ns_pad, (const int2 *)g_xb.wrange,
                           (const int *)sh.nitem_w, (const double *)g_xb.rowpart, (const double *)colpart, (const int *)nullptr,
                           (const double *)nullptr, (const double *)nullptr, (double *)nullptr, (double *)nullptr, (const XCtrl *)nullptr, tmp,
                           0, (const int *)nullptr, 0, (const double *)nullptr, sh.w_lo, sh.w_hi);
        hipLaunchKernelGGL(k_xt_test_accumulate, dim3(gb), dim3(256), 0, st, ns, acc, (const double *)tmp);
        KCHK();
    }
    HIPCHK(hipMemsetAsync(g_xb.rowpart, 0, rowpart_bytes, st));       // the resident layout again: every cell with a tile is rewritten by each apply
    std::vector<double> hf(ns), ha(ns);
    HIPCHK(hipMemcpyAsync(hf.data(), full, (size_t)ns * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(ha.data(), acc, (size_t)ns * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    double md = 0.0, ma = 0.0;
    for (int i = 0; i < ns; ++i) { md = std::max(md, fabs(hf[i] - ha[i])); ma = std::max(ma, fabs(hf[i])); }
    if (max_abs_diff) *max_abs_diff = md;
    if (max_abs) *max_abs = ma;
    if (subblocks_sum) *subblocks_sum = sb_sum;
    if (items_sum) *items_sum = it_sum;
    return e.err_code;
}


// ---- test / measurement aid: the slab-distributed block-CG (xtb_slab.inc) with nranks VIRTUAL ranks on ONE GPU -----------------------------------
// On the X left resident by the last single-GPU solve: (1) the block-CG of width `width` as one GPU runs it, from a zero start, to `tol`;
// (2) the same system by the slab-distributed loop with nranks virtual ranks -- shares of the tiles built as a sharded assembly builds them, rows
// owned by lateral slabs, every exchange a device copy.  The emulation itself fails (error 13) if the virtual ranks leave the loop at
// different sweeps or end with different bits.  Out: largest deviation of the two solutions relative to the largest entry, both sweep counts,
// the mean kernel times of virtual rank time_rank over sweeps 2 ... 25 (time_rank < 0: none), the doubles a rank receives per sweep in the three
// exchanges (largest over the ranks).  Scratch vectors of the last solve are overwritten; delivered results are not.
// sweep_cap > 0 (measurement run): the distributed loop stops after that many sweeps and the one-GPU solve is skipped (rel_diff = -1).
extern "C" int dkmc_xtb_emulate_slabs(int nranks, int width, double tol, int time_rank, int sweep_cap, double *rel_diff, int *iters_slab, int *iters_ref,
                                      double *times_us /* [8] */, long long *xdoubles /* [3] */, int *rows_min_max /* [2] */)
{
    Engine &e = eng(); hipStream_t st = e.stream; const XTState &X = g_xt;
    if (!X.valid || comm_attached() || X.tile_n != X.ntiles || X.ns <= 0 || !g_xb.ay) return dkmc_fail(13, "xtb_emulate_slabs: needs the X of a single-GPU solve", __FILE__, __LINE__);
    if (nranks < 1 || nranks > 32) return dkmc_fail(13, "xtb_emulate_slabs: 1 ... 32 ranks", __FILE__, __LINE__);
    const int m = X.Nsub, ns = X.ns, s = std::max(2, std::min(width, 16));
    double *sc = (double *)e.buf[S_CG_S], *sS = (double *)e.buf[S_CG_PS], *rhs = (double *)e.buf[S_X_RHS];
    double *yb = (double *)scratch(S_XTB_EMU_Y, (size_t)(m + 8) * 8 * 2);
    XCtrl *ctrl = (XCtrl *)scratch(S_XTB_EMU_CTRL, 256);
    if (!sc || !sS || !rhs || !yb || !ctrl) return e.err_code ? e.err_code : dkmc_fail(13, "xtb_emulate_slabs: no solver state", __FILE__, __LINE__);
    sS += X.ns_pad;
    XtbArgs B{};
    B.m = m; B.ns = ns; B.ns_pad = X.ns_pad; B.nK = X.nK; B.nW = X.nW; B.s = s;
    B.items = (const XItem *)g_xb.items + X.item_lo; B.item_n = X.item_n; B.tiles = g_xb.tiles; B.sub_base = (int)X.sub_base; B.tval = g_xb.tval;
    B.wrange = g_xb.wrange; B.nitem_w = g_xb.nitem_w; B.nrecords = X.nitems >> X.rec_shift;
    B.srow = g_xb.srow; B.sS = sS; B.nsrank = g_xb.nsrank; B.rp = g_xb.rp; B.ci = g_xb.ci; B.val = g_xb.val; B.sc = sc;
    B.ax = g_xb.ax; B.ay = g_xb.ay; B.az = g_xb.az; B.b = rhs; B.ctrl = ctrl; B.tol2 = tol * tol;
    B.nt_loads = (size_t)X.sub_n * XT_SUB * 8 > ((size_t)200 << 20); B.sharded = false; B.w_lo = X.w_lo; B.w_hi = X.w_hi;
    const int hint = e.x_iter_hint;
    // (1) one GPU
    double *yref = yb, *yslab = yb + m + 8;
    HIPCHK(hipMemsetAsync(yb, 0, (size_t)(m + 8) * 8 * 2, st));
    B.y = yref;
    int it_ref = 0, it_slab = 0; double rr = 0.0;
    int rc = 0;
    // (the one-GPU loop WITHOUT the split polynomial preconditioner: the slab-distributed loop has none, and the two are compared sweep for sweep)
    if (sweep_cap <= 0) { const int pd0 = e.x_poly; e.x_poly = 0; rc = xtb_cg(B, &it_ref, &rr); e.x_poly = pd0; if (rc && rc != DKMC_XTB_BREAKDOWN) return rc; }
    // (2) nranks virtual ranks: the shares of an nranks-way assembly
    const int ntiles = X.ntiles;
    int kc = std::max(1, std::min(XT_MAXKC, ntiles / nranks / 4096));
    if (ntiles <= 2048 * nranks) kc = std::max(1, std::min(XT_MAXKC, (ntiles + 1024 * nranks - 1) / (1024 * nranks)));
    std::vector<XShare> sh((size_t)nranks);
    for (int r = 0; r < nranks; ++r) {
        rc = xt_build_items(X.nK, X.nW, kc, ntiles, X.nsub_total, (const int *)g_xb.toff, (const XTile *)g_xb.tiles, nranks, r, S_XT_T_NITEMW, S_XT_T_ITEMS, S_XT_T_SPLIT, &sh[r], 2);
        if (rc) return rc;
        if ((sh[r].item_lo | sh[r].item_n) & 3) return dkmc_fail(48, "xtb_emulate_slabs: run list not padded to groups of four", __FILE__, __LINE__);
    }
    B.y = yslab;
    rc = xtb_cg_slab_emulate(B, nranks, sh.data(), time_rank, sweep_cap, &it_slab, &rr, times_us, xdoubles);
    e.x_iter_hint = hint;
    if (rc && rc != DKMC_XTB_BREAKDOWN) return rc;
    std::vector<double> h0((size_t)m), h1((size_t)m);
    HIPCHK(hipMemcpyAsync(h0.data(), yref, (size_t)m * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(h1.data(), yslab, (size_t)m * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    double md = 0.0, ma = 0.0;
    for (int i = 0; i < m; ++i) { md = std::max(md, fabs(h0[i] - h1[i])); ma = std::max(ma, fabs(h0[i])); }
    if (rel_diff) *rel_diff = sweep_cap > 0 ? -1.0 : (ma > 0.0 ? md / ma : md);
    if (iters_slab) *iters_slab = it_slab;
    if (iters_ref) *iters_ref = it_ref;
    if (rows_min_max) {          // balance of the slabs: rows of the smallest and the largest one (the table the solver built)
        std::vector<int> tab((size_t)nranks);
        HIPCHK(hipMemcpy(tab.data(), (int *)e.buf[S_XTB_SLAB_TAB] + 4096 + 32 + 2, (size_t)nranks * 4, hipMemcpyDeviceToHost));
        rows_min_max[0] = *std::min_element(tab.begin(), tab.end()); rows_min_max[1] = *std::max_element(tab.begin(), tab.end());
    }
    return e.err_code;
}

const xrp_t *xt_xs_rp() { return g_xb.rp; }
const int *xt_xs_col() { return g_xb.ci; }
const double *xt_xs_val() { return g_xb.val; }
bool xt_valid() { return g_xt.valid; }

// CSR of the whole X (inspection, small systems, one rank): Xs merged with both triangles of the tiles, rows column-sorted
int xt_export_csr(int *rows_out, long long *nnz_out, int *h_rp, int *h_col, double *h_data)
{
    Engine &e = eng(); const XTState &X = g_xt;
    if (!X.valid) return dkmc_fail(13, "get_last_X: no assembled X", __FILE__, __LINE__);
    if (comm_attached()) return dkmc_fail(13, "get_last_X: not available on a sharded X", __FILE__, __LINE__);
    HIPCHK(hipStreamSynchronize(e.stream));
    const long long nnz = X.xs_nnz + 2 * (long long)X.t_upper;
    if (rows_out) *rows_out = X.Nsub;
    if (nnz_out) *nnz_out = nnz;
    if (!h_rp && !h_col && !h_data) return 0;
    if (nnz > 2147483647LL || X.nsub_total > (1ll << 18)) return dkmc_fail(12, "get_last_X: X is too large for the int32 CSR copy of this call", __FILE__, __LINE__);
    std::vector<xrp_t> rp((size_t)X.Nsub + 1); std::vector<int> ci((size_t)X.xs_nnz); std::vector<double> va((size_t)X.xs_nnz);
    HIPCHK(hipMemcpy(rp.data(), g_xb.rp, rp.size() * sizeof(xrp_t), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(ci.data(), g_xb.ci, ci.size() * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(va.data(), g_xb.val, va.size() * 8, hipMemcpyDeviceToHost));
    std::vector<XTile> tl((size_t)X.ntiles); std::vector<double> tv((size_t)X.nsub_total * XT_SUB); std::vector<int> srow((size_t)X.ns_pad);
    if (X.ntiles) HIPCHK(hipMemcpy(tl.data(), g_xb.tiles, tl.size() * sizeof(XTile), hipMemcpyDeviceToHost));
    if (X.nsub_total) HIPCHK(hipMemcpy(tv.data(), g_xb.tval, tv.size() * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(srow.data(), g_xb.srow, srow.size() * 4, hipMemcpyDeviceToHost));
    std::vector<std::vector<std::pair<int, double>>> rowsv((size_t)X.Nsub);
    for (int i = 0; i < X.Nsub; ++i) for (xrp_t p = rp[i]; p < rp[i + 1]; ++p) rowsv[i].push_back({ci[p], va[p]});
    for (const XTile &t : tl) {
        int sl = 0;
        for (int q = 0; q < 8; ++q) {
            if (!((t.mask >> q) & 1u)) continue;
            const double *sb = tv.data() + ((size_t)t.soff + sl) * XT_SUB; ++sl;
            for (int r = 0; r < XT_R; ++r) for (int c = 0; c < XT_SBW; ++c) {
                const double v = sb[r * XT_SBW + c];
                if (v == 0.0) continue;
                const int s = XT_R * t.k + r, sc = XT_C * t.w + XT_SBW * q + c;
                rowsv[srow[s]].push_back({srow[sc], v}); rowsv[srow[sc]].push_back({srow[s], v});
            }
        }
    }
    long long pos = 0;
    for (int i = 0; i < X.Nsub; ++i) {
        std::sort(rowsv[i].begin(), rowsv[i].end(), [](const std::pair<int, double> &a, const std::pair<int, double> &b) { return a.first < b.first; });
        if (h_rp) h_rp[i] = (int)pos;
        for (auto &pr : rowsv[i]) { if (h_col) h_col[pos] = pr.first; if (h_data) h_data[pos] = pr.second; ++pos; }
    }
    if (h_rp) h_rp[X.Nsub] = (int)pos;
    if (pos != nnz) return dkmc_fail(13, "get_last_X: entry count of the tiles does not match the fill count", __FILE__, __LINE__);
    return 0;
}
