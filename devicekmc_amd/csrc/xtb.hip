// xtb.hip -- block-CG of the current solve on the tiled X, with the tile x panel product on the matrix cores (dkmc_set_x_block(s), s = 2 ... 16).
//
// What it replaces: solve_sparse_CG_Jacobi (iterative_solvers_gpu.cu:309-480) as update_power_gpu_sparse calls it
// (current_solver_gpu.cu:963-994), and -- for the tunnelling block -- the dense tile product of the reference's unfinished split path
// (add_submatrix_product, iterative_solvers_gpu.cu:634-652; solve_sparse_CG_splitmatrix :656-821), which applies the tile as a GEMV.
// X changes every superstep and has ONE physical right-hand side, so there is nothing to batch -- but a block-CG whose other columns are
// AUXILIARY right-hand sides (fixed-seed pseudo-random, zero start) searches the block Krylov space of all s columns and deflates the
// outlying part of X's spectrum (loop_G = 1e12, high_G = 1e5, low_G = 1e-8 nodes): on the oracle's X of the 85 071-site device the
// reference's stop test on the physical column is met after 666 / 208 / 133 / 95 sweeps at s = 1 / 4 / 8 / 16 (tools/blockcg_proto.py).
// One sweep streams every stored 32 x 32 sub-block ONCE (8 KiB, both triangles from one read, as the single-vector kernel does) and
// multiplies it and its transpose into 32 x 16 panels: 4 s flops per 8 bytes -- at s = 16 a contraction of 8 flop/B, which is what
// north_star reserves the matrix cores for.
//
// Algorithm (scaled system A = S X S, sign convention of the reference: R = A Y - B, first direction -R):
//     T = A P                                            one sweep
//     c = -(P'T)^-1 P'R ;  Y += P c ;  R+ = R + T c
//     beta = (P'T)^-1 (T'R + T'T c)                      = (P'T)^-1 T'R+ : makes the new directions A-conjugate to P
//     P+ = (-R+ + P beta) W                              W = inverse transposed Cholesky factor of the Gram matrix of (-R+ + P beta):
//                                                        directions stay orthonormal, the s x s systems well conditioned
// Every s x s matrix (P'T, P'R, T'R, T'T, R'R) comes from ONE pass over the panels (k_xtb_rows, MFMA with the rows as the contraction
// index); R+'R+ and the Gram matrix of the new directions follow from them algebraically (exact for the R+ actually formed, the
// identity the single-vector loop of xt.hip uses for r'.r'), so an iteration has no second global reduction:
//     k_xtb_apply (+ k_xtb_neigh)   T = A P: tiles on the matrix cores; neighbour part Xs as CSR x panel
//     k_xtb_rows                    partial sums -> S rows of T; partial Gram matrices
//     k_xtb_gred + k_xtb_small      Gram matrices in a fixed order; ONE wave: the s x s algebra, stop test on column 0 (||r||^2 <= tol^2)
//     k_xtb_step                    Y, R, P, Q = S P as 16-row x 16 x 16 MFMA products
// Column 0 carries the reference's right-hand side and start vector; the result is its solution, to the same stop test.  The iterate
// sequence is NOT the reference's (s = 1 keeps that: xt.hip).
//
// Tile x panel product (k_xtb_apply), per 32 x 32 sub-block (instruction and lane maps: see the kernel):
//   column sums  Yc[col][v] += sum_row T[row][col] Qr[row][v]:  the 8 wave loads of the stream (lane (cc, rr) holds rows 4 j + rr, columns
//                2 cc, 2 cc + 1) ARE the A operands (tile row on the k index): no data movement.
//   row sums     Yr[row][v] += sum_col T[row][col] Qc[col][v]:  needs the sub-block with the COLUMN on the k index: one round trip through
//                wave-private LDS (8 ds_write_b128 into an XOR-swizzled image, 8 ds_read_b128, both conflict-free).
// Column sums stay in accumulator registers over the run of tiles (one 256-column strip), the four waves of a workgroup leave one
// combined record; row sums (32 x s) go to a grid-indexed array per tile.
// Only column 0 of Y is kept (a vector): the auxiliary solutions are never needed.  The neighbour part gathers P and the scaling, so no
// scaled full-length copy of P exists; the compact scaled copy over S (QS) feeds the tiles.
#include "xtiles.h"
#include "slab.h"
#include <hip/hip_ext.h>
#include <algorithm>
#include <functional>
#include <vector>

#define XB_SP 16                      // vectors per panel row (padded block width)
#define XB_MAXPOLY 16                 // largest degree of the split polynomial preconditioner (dkmc_set_x_poly)
#define XB_DSPLIT 8                   // workgroups per driver row of Xs
#define XB_NG 6                       // Gram matrices per pass: P'T, P'R, T'R, T'T, R'R, P'P
typedef double dbl4 __attribute__((ext_vector_type(4)));
#define XB_MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)
#define XB_MFMA4(a, b, c) __builtin_amdgcn_mfma_f64_4x4x4f64((a), (b), (c), 0, 0, 0)

// auxiliary right-hand sides: a hash of (row, column) -> uniform in [-1, 1); the same on every rank and in tools/blockcg_proto.py
__device__ __forceinline__ double xtb_aux(int row, int col)
{
    unsigned long long h = (unsigned long long)row * 0x9E3779B97F4A7C15ull + (unsigned long long)col * 0xC2B2AE3D27D4EB4Full + 12345ull;
    h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    return (double)(h >> 11) * (1.0 / 4503599627370496.0) - 1.0;
}
// column v of the scaled right-hand side panel at a row: the physical one, an auxiliary one, or padding.
// Auxiliary columns are a free choice (only their block Krylov space matters).  aux != nullptr: the SMOOTH set -- column v is
// cos(kx pi xi) cos(ky pi eta) cos(kz pi zeta) / s_row, (xi, eta, zeta) the atom's position scaled to the unit box and (kx, ky, kz) the v-th
// lowest mode of the Laplacian of the device's bounding box (k_xtb_modes: adapts to the aspect ratio -- along x for a narrow stack, lateral
// for a wide one).  The neighbour part of X is a graph Laplacian: its low modes are smooth, and right-hand sides rich in them deflate that end
// of the spectrum from the first sweeps (oracle's X, tools/blockcg_proto.py, x modes: 6.4e3 rows 48 -> 31 sweeps, 5.8e4 rows 95 -> 76).
// Smooth systems also CONVERGE sooner than the physical column; close to a converged tolerance (1e-10 and below) their residual columns
// vanish and the s x s systems lose rank, so below 1e-8 the hash set stays (dkmc_set_x_aux).
struct XbAux { double lo[3], hi[3]; int k[16][3]; unsigned long long mm[6]; int hs, pad_; };      // hs: columns 1 ... hs - 1 take the smooth set, hs ... s - 1 the hash set
__device__ __forceinline__ double xtb_rhs(const double *__restrict__ b, int row, int v, int s, const XbAux *__restrict__ aux = nullptr,
                                          const double *__restrict__ ax = nullptr, const double *__restrict__ ay = nullptr, const double *__restrict__ az = nullptr,
                                          const double *__restrict__ sc = nullptr)
{
    if (v == 0) return b[row];
    if (v >= s) return 0.0;
    if (!aux || v >= aux->hs) return xtb_aux(row, v);
    // rows 0 / 1: the ground-side and the source-side driver node, at the two ends of x, in the middle of the cross-section
    double u[3];
    if (row < 2) { u[0] = row == 0 ? 1.0 : 0.0; u[1] = 0.5; u[2] = 0.5; }
    else {
        const double c[3] = {ax[row - 2], ay[row - 2], az[row - 2]};
#pragma unroll
        for (int d = 0; d < 3; ++d) u[d] = aux->hi[d] > aux->lo[d] ? (c[d] - aux->lo[d]) / (aux->hi[d] - aux->lo[d]) : 0.0;
    }
    return cospi((double)aux->k[v][0] * u[0]) * cospi((double)aux->k[v][1] * u[1]) * cospi((double)aux->k[v][2] * u[2]) / sc[row];
}
// bounding box of the atoms (order-preserving integer image of c + 2^20 > 0: the result does not depend on the order of the atomics)
__global__ __launch_bounds__(256) void k_xtb_box(int na, const double *__restrict__ ax, const double *__restrict__ ay, const double *__restrict__ az, XbAux *aux)
{
    // wave minima / maxima first: one atomic per wave and bound instead of one per atom (6 x 5.8e4 atomics on six words cost 1 ms at 85 k sites)
    unsigned long long lo[3] = {~0ull, ~0ull, ~0ull}, hi[3] = {0ull, 0ull, 0ull};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < na; i += gridDim.x * blockDim.x) {
        const double c[3] = {ax[i], ay[i], az[i]};
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const unsigned long long u = (unsigned long long)__double_as_longlong(c[d] + 1048576.0);
            lo[d] = u < lo[d] ? u : lo[d]; hi[d] = u > hi[d] ? u : hi[d];
        }
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long l2 = __shfl_xor(lo[d], off, WAVE), h2 = __shfl_xor(hi[d], off, WAVE);
            lo[d] = l2 < lo[d] ? l2 : lo[d]; hi[d] = h2 > hi[d] ? h2 : hi[d];
        }
        if ((threadIdx.x & 63) == 0) { atomicMin(&aux->mm[2 * d], lo[d]); atomicMax(&aux->mm[2 * d + 1], hi[d]); }
    }
}
// the s - 1 lowest non-constant modes of the box: (kx / Lx)^2 + (ky / Ly)^2 + (kz / Lz)^2 ascending over kx, ky, kz < 8, ties with the larger
// kx (then ky) first; x_only: (v, 0, 0).  One workgroup of 512: thread = candidate, its rank by counting.
__global__ __launch_bounds__(512) void k_xtb_modes(XbAux *aux, int x_only)
{
    __shared__ double ev[512];
    const int code = threadIdx.x, kx = code >> 6, ky = (code >> 3) & 7, kz = code & 7;         // code = kx * 64 + ky * 8 + kz
    double L[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const double lo = __longlong_as_double((long long)aux->mm[2 * d]) - 1048576.0, hi = __longlong_as_double((long long)aux->mm[2 * d + 1]) - 1048576.0;
        if (code == 0) { aux->lo[d] = lo; aux->hi[d] = hi; }
        L[d] = hi - lo;
        if (!(L[d] > 1e-6)) L[d] = 1e-6;
    }
    ev[code] = (kx / L[0]) * (kx / L[0]) + (ky / L[1]) * (ky / L[1]) + (kz / L[2]) * (kz / L[2]);
    __syncthreads();
    if (code == 0) { aux->k[0][0] = aux->k[0][1] = aux->k[0][2] = 0; return; }
    if (x_only) { if (code < 16) { aux->k[code][0] = code; aux->k[code][1] = 0; aux->k[code][2] = 0; } return; }
    const double mine = ev[code];
    int rank = 0;
    for (int c = 1; c < 512; ++c) rank += (ev[c] < mine || (ev[c] == mine && c > code)) ? 1 : 0;
    if (rank < 15) { aux->k[rank + 1][0] = kx; aux->k[rank + 1][1] = ky; aux->k[rank + 1][2] = kz; }
}
// position of (S rank r, vector v) in the compact panel QS: rows pairwise interleaved, [r / 2][v][r % 2] -- two consecutive rows of one
// vector are one 16-byte read (two k-steps of the row product).  (The LDS copy of a strip's window adds a swizzle: k_xtb_apply.)
__device__ __forceinline__ size_t xtb_qs_pos(int r, int v) { return (size_t)(r >> 1) * 32 + 2 * v + (r & 1); }

// ---- tiles x panel -----------------------------------------------------------------------------------------------------------------
// One run of tiles per wave, the four runs of a workgroup in one strip (the run list is padded to groups of four per strip: xt.hip).
// v_mfma_f64_4x4x4_4b_f64 (four independent 4 x 4 x 4 products per instruction; lane maps probed on gfx950, tools/probe_mfma_f64_4x4x4.hip:
// A[i][k] of block n: lane i + 4 n + 16 k; B[k][j]: lane j + 4 n + 16 k; D[i][j]: lane j + 4 n + 16 i) issues every 18 cycles -- 68 TFLOP/s
// measured chip-wide, twice the 16x16x4 form (tools/bench_mfma_f64.hip: 140 cycles, 34 TFLOP/s).  The four blocks of an instruction are
// four groups of tile rows (row sums) or columns (column sums) against the SAME four vectors, so the panel operand is replicated over the
// blocks and the cost scales with NG = ceil(s / 4) vector groups: 32 NG instructions per 32 x 32 sub-block.
template <int NTL, int NG, int variant = 0>
__global__ __launch_bounds__(XT_NT) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k_xtb_apply(int nitems, const XItem *__restrict__ items, const XTile *__restrict__ tiles, int sub_base, const double *__restrict__ tval,
                 const double *__restrict__ QS, int nW, double *__restrict__ rowpartB, double *__restrict__ colpartB, const XCtrl *ctrl)
{
    // variant 0 = the product kernel.  Every other value is the ROUND-4 FORM of the loop (stages issued in bursts, conditional loads at the tile end,
    // panel rows loaded directly), kept for same-box comparisons (dkmc_set_x_apply_form(1) = variant 8, same results as 0) and as the carrier of the
    // measurement variants of dkmc_xtb_time_apply: 1 = no matrix instructions (stream + LDS traffic only), 2 = the tile stream is not re-read
    // (matrix instructions + LDS traffic only), 3 = operand stages of one k-pair, 4 = no LDS traffic (the row sums replaced by as many products on
    // the loaded registers), 7 = 2 and 4 together (the matrix instructions alone); 10 = the PRODUCT form without re-reading the tile stream; the
    // results of 1, 2, 4, 7 and 10 are meaningless
    constexpr int so = 4 * NG;                                                 // vectors per row of the partial-sum arrays
    constexpr bool PF = variant == 0 || variant == 10 || variant == 12;        // product form of the loop (12, measurement: partial tiles skipped)
    __shared__ __attribute__((aligned(16))) double qc[XT_C * XB_SP];          // the strip's 256 panel rows in QS order (32 KiB)
    __shared__ __attribute__((aligned(16))) double ts[4 * 2 * XT_SUB];        // per wave: two sub-block images (2 x 8 KiB)
    __shared__ __attribute__((aligned(16))) double brs[PF ? 4 * XT_R * XB_SP : 2];   // per wave: the next tile's 32 panel rows (4 KiB)
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63, cc = lane & 15, rr = lane >> 4, jv = lane & 3, blk = cc >> 2;
    const int item = (int)blockIdx.x * 4 + wv;
    const XItem it = items[min(item, nitems - 1)];
    if (ctrl->done) return;                                                    // uniform over the launch
    {   // the window of the strip, 16-byte chunks (vector v of a row pair) in QS order except for a swizzle of the vector index by
        // 4 (row pair % 4): the four row pairs a wave reads at once then sit on different LDS banks
        const dbl2 *src = reinterpret_cast<const dbl2 *>(QS + (size_t)it.w * XT_C * XB_SP);
        dbl2 *dst = reinterpret_cast<dbl2 *>(qc);
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int ch = threadIdx.x + 256 * u; dst[ch ^ (((ch >> 4) & 3) << 2)] = src[ch]; }
    }
    __syncthreads();
    double *tsw = ts + (size_t)wv * 2 * XT_SUB;
    double *brw = brs + (PF ? (size_t)wv * XT_R * XB_SP : 0);
    double Yc[8][2][NG];
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int g = 0; g < NG; ++g) Yc[q][e][g] = 0.0;
    // LDS offsets (doubles) of this lane.  Sub-block image: g(r, c) = 32 r + (c ^ ((r & 15) << 1)), written as the loads arrive (lane
    // (cc, rr) holds rows 4 j + rr, columns 2 cc, 2 cc + 1), read with the row on the lane (row 16 mb + cc, columns 8 kk + 2 rr, + 1)
    int woff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { const int r = 4 * j + rr; woff[j] = 32 * r + ((2 * cc) ^ ((r & 15) << 1)); }
    const int roff = 32 * cc;                                                  // + 512 mb + ((8 kk + 2 rr) ^ (cc << 1))
    // panel operands: vector 4 g + jv of panel row rho sits at xtb_qs_pos(rho, 4 g + jv); for rho = 32 q + 8 kk + 2 rr (+ 1): one 16-byte read
    int qoff[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) qoff[g] = rr * 32 + ((2 * (4 * g + jv)) ^ (rr << 3));      // + (16 q + 4 kk) * 32
#define XB_LD(dst, slot) if ((variant != 2 && variant != 7 && variant != 10) || (slot) < 2) { _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) dst[j_] = NTL ? __builtin_nontemporal_load(base + (size_t)(8 * (slot) + j_) * 64) : base[(size_t)(8 * (slot) + j_) * 64]; }
    // panel operands of the column sums: rows 4 j + rr of a tile's 32 panel rows, vectors 4 g + jv (the same in all four blocks)
#define XB_LDBR(k_)                                                                                                            \
    {                                                                                                                           \
        const double *qr_ = QS + (size_t)(k_) * XT_R * XB_SP;                                                                   \
        _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) {                                                                      \
            const int rho_ = 4 * j_ + rr, r32_ = rho_ >> 1;                                                                     \
            _Pragma("unroll") for (int g_ = 0; g_ < NG; ++g_) br[j_][g_] = qr_[r32_ * 32 + 2 * (4 * g_ + jv) + (rho_ & 1)];      \
        }                                                                                                                       \
    }
    // the same rows by way of LDS -- four coalesced 16-byte loads per lane EARLY in the tile (LDBN), a plain copy into brw (WBN), read back in br's
    // lane map once the tile's last column sums are issued (RDBRH).  vmcnt retires in order: loaded at the END of a tile (round-4 form) these rows sat
    // behind the prefetched sub-blocks and the next tile's first stage drained the queue
#define XB_LDBN(k_) { const dbl2 *qn_ = reinterpret_cast<const dbl2 *>(QS + (size_t)(k_) * XT_R * XB_SP) + lane; _Pragma("unroll") for (int u_ = 0; u_ < 4; ++u_) bn[u_] = qn_[64 * u_]; }
#define XB_WBN() { dbl2 *bw_ = reinterpret_cast<dbl2 *>(brw) + lane; _Pragma("unroll") for (int u_ = 0; u_ < 4; ++u_) bw_[64 * u_] = bn[u_]; }
    // One sub-block = four stages of 8 NG matrix instructions each, software-pipelined by hand: hipcc issues an LDS read right in front of
    // the instructions that use it (measured: one exposed LDS latency per 4 matrix instructions, 3.5 ms per sweep at 9.4e5 sites where the
    // instructions alone take 1.9), so the operands of the row sums are requested one stage ahead and sched_barriers keep the order:
    //   image -> LDS | read operands of k-pairs 0, 1 | COLUMN sums, loads 0-3 | read operands of k-pairs 2, 3 | ROW sums, k-pairs 0, 1 |
    //   COLUMN sums, loads 4-7 | ROW sums, k-pairs 2, 3
    // Column sums come straight from the loaded registers (they are the A operands with the tile ROW on the k index); row sums from the
    // image read back with the tile COLUMN on the k index.
    struct RowOps { dbl2 a0[2], a1[2], bc[2][NG]; };
#define XB_RDROW(R, q, bufi, kk0)                                                                                              \
    {                                                                                                                           \
        const double *img_ = tsw + (bufi) * XT_SUB;                                                                             \
        _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_) {                                                                      \
            const int sw_ = (8 * ((kk0) + h_) + 2 * rr) ^ (cc << 1);                                                            \
            R.a0[h_] = *reinterpret_cast<const dbl2 *>(img_ + roff + sw_);                                                      \
            R.a1[h_] = *reinterpret_cast<const dbl2 *>(img_ + roff + 512 + sw_);                                                \
            _Pragma("unroll") for (int g_ = 0; g_ < NG; ++g_)                                                                   \
                R.bc[h_][g_] = *reinterpret_cast<const dbl2 *>(qc + qoff[g_] + (16 * (q) + 4 * ((kk0) + h_)) * 32);             \
        }                                                                                                                       \
    }
#define XB_COLH(vv, q, j0)                                                                                                     \
    if (variant != 1) {                                                                                                         \
        _Pragma("unroll") for (int j_ = (j0); j_ < (j0) + 4; ++j_)                                                              \
            _Pragma("unroll") for (int g_ = 0; g_ < NG; ++g_) {                                                                 \
                Yc[q][0][g_] = XB_MFMA4(vv[j_].x, br[j_][g_], Yc[q][0][g_]);                                                    \
                Yc[q][1][g_] = XB_MFMA4(vv[j_].y, br[j_][g_], Yc[q][1][g_]);                                                    \
            }                                                                                                                   \
    }
#define XB_ROWH(R)                                                                                                             \
    _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_)                                                                            \
        _Pragma("unroll") for (int g_ = 0; g_ < NG; ++g_) {                                                                     \
            if (variant != 1) {                                                                                                 \
                Yr[0][g_] = XB_MFMA4(R.a0[h_].x, R.bc[h_][g_].x, Yr[0][g_]); Yr[1][g_] = XB_MFMA4(R.a1[h_].x, R.bc[h_][g_].x, Yr[1][g_]); \
                Yr[0][g_] = XB_MFMA4(R.a0[h_].y, R.bc[h_][g_].y, Yr[0][g_]); Yr[1][g_] = XB_MFMA4(R.a1[h_].y, R.bc[h_][g_].y, Yr[1][g_]); \
            } else { Yr[0][g_] += R.a0[h_].x + R.bc[h_][g_].x; Yr[1][g_] += R.a1[h_].y + R.bc[h_][g_].y; }                       \
        }
    // variant 3 (measurement): stages of ONE k-pair (16 matrix instructions at s = 16) instead of two: half the staging registers
    struct RowOps1 { dbl2 a0, a1, bc[NG]; };
#define XB_RD1(R, q, bufi, kk_)                                                                                                \
    {                                                                                                                           \
        const double *img_ = tsw + (bufi) * XT_SUB;                                                                             \
        const int sw_ = (8 * (kk_) + 2 * rr) ^ (cc << 1);                                                                       \
        R.a0 = *reinterpret_cast<const dbl2 *>(img_ + roff + sw_);                                                              \
        R.a1 = *reinterpret_cast<const dbl2 *>(img_ + roff + 512 + sw_);                                                        \
        _Pragma("unroll") for (int g_ = 0; g_ < NG; ++g_) R.bc[g_] = *reinterpret_cast<const dbl2 *>(qc + qoff[g_] + (16 * (q) + 4 * (kk_)) * 32); \
    }
#define XB_ROW1(R)                                                                                                             \
    _Pragma("unroll") for (int g_ = 0; g_ < NG; ++g_) {                                                                         \
        Yr[0][g_] = XB_MFMA4(R.a0.x, R.bc[g_].x, Yr[0][g_]); Yr[1][g_] = XB_MFMA4(R.a1.x, R.bc[g_].x, Yr[1][g_]);               \
        Yr[0][g_] = XB_MFMA4(R.a0.y, R.bc[g_].y, Yr[0][g_]); Yr[1][g_] = XB_MFMA4(R.a1.y, R.bc[g_].y, Yr[1][g_]);               \
    }
#define XB_COLQ(vv, q, j0)                                                                                                     \
    _Pragma("unroll") for (int j_ = (j0); j_ < (j0) + 2; ++j_)                                                                  \
        _Pragma("unroll") for (int g_ = 0; g_ < NG; ++g_) {                                                                     \
            Yc[q][0][g_] = XB_MFMA4(vv[j_].x, br[j_][g_], Yc[q][0][g_]);                                                        \
            Yc[q][1][g_] = XB_MFMA4(vv[j_].y, br[j_][g_], Yc[q][1][g_]);                                                        \
        }
#define XB_SB() __builtin_amdgcn_sched_barrier(0);
#define XB_COL(vv, q, bufi)                                                                                                    \
    {                                                                                                                           \
        double *img_ = tsw + (bufi) * XT_SUB;                                                                                   \
        if (variant != 4 && variant != 7) { _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) *reinterpret_cast<dbl2 *>(img_ + woff[j_]) = vv[j_]; } \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                                                                  \
        __builtin_amdgcn_wave_barrier();                                                                                        \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                                                                  \
        if (variant == 4 || variant == 7) {                                                                                     \
            XB_COLH(vv, q, 0) XB_COLH(vv, q, 4)                                                                                 \
            _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_)                                                                    \
                _Pragma("unroll") for (int g_ = 0; g_ < NG; ++g_) {                                                             \
                    Yr[0][g_] = XB_MFMA4(vv[j_].x, br[j_][g_], Yr[0][g_]); Yr[1][g_] = XB_MFMA4(vv[j_].y, br[j_][g_], Yr[1][g_]); \
                }                                                                                                               \
        } else if (variant == 3) {                                                                                              \
            XB_RD1(Ra, q, bufi, 0) XB_SB() XB_COLQ(vv, q, 0) XB_SB() XB_RD1(Rb, q, bufi, 1) XB_SB() XB_ROW1(Ra) XB_SB()         \
            XB_COLQ(vv, q, 2) XB_SB() XB_RD1(Ra, q, bufi, 2) XB_SB() XB_ROW1(Rb) XB_SB() XB_COLQ(vv, q, 4) XB_SB()              \
            XB_RD1(Rb, q, bufi, 3) XB_SB() XB_ROW1(Ra) XB_SB() XB_COLQ(vv, q, 6) XB_SB()                                        \
        } else {                                                                                                                \
        XB_RDROW(R0, q, bufi, 0)                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                                      \
        XB_COLH(vv, q, 0)                                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                                      \
        XB_RDROW(R1, q, bufi, 2)                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                                      \
        XB_COLH(vv, q, 4)                                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                                      \
        }                                                                                                                       \
    }
#define XB_ROW(q, bufi)                                                                                                        \
    {                                                                                                                           \
        if (variant == 4 || variant == 7) { }                                                                                   \
        else if (variant == 3) { XB_ROW1(Rb) XB_SB() }                                                                          \
        else { XB_ROWH(R0) __builtin_amdgcn_sched_barrier(0); XB_ROWH(R1) __builtin_amdgcn_sched_barrier(0); }                  \
    }
    // PRODUCT FORM (variant 0): the same four stages of 8 NG matrix instructions, but nothing is issued in a burst BETWEEN stages: the LDS reads of the row operands,
    // the image write of the next sub-block and the stream loads ride INSIDE a stage, spread over its matrix instructions by sched_group_barriers
    // (an instruction of another pipe issues in the shadow of a matrix instruction; a burst of 12-20 LDS instructions leaves the matrix pipe idle):
    //   S1: read row operands k 0,1 of q     + COLUMN sums loads 0-3        S3: load sub-block q + 2 into the free set + ROW sums k 0,1
    //   S2: read row operands k 2,3 of q     + COLUMN sums loads 4-7        S4: write the image of sub-block q + 1   + ROW sums k 2,3
#define XB_G(mask_, n_) __builtin_amdgcn_sched_group_barrier(mask_, n_, 0);
#define XB_GREP(cnt_, body_) _Pragma("unroll") for (int gi_ = 0; gi_ < (cnt_); ++gi_) { body_ }
#define XB_WIMG(vv, bufi)                                                                                                      \
    {                                                                                                                           \
        double *img_ = tsw + (bufi) * XT_SUB;                                                                                   \
        _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) *reinterpret_cast<dbl2 *>(img_ + woff[j_]) = vv[j_];                    \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                                                                  \
        __builtin_amdgcn_wave_barrier();                                                                                        \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                                                                  \
    }
#define XB_RDBRH(j0)                                                                                                           \
    _Pragma("unroll") for (int j_ = (j0); j_ < (j0) + 4; ++j_) {                                                                \
        const int rho_ = 4 * j_ + rr; const int r32_ = rho_ >> 1;                                                                          \
        _Pragma("unroll") for (int g_ = 0; g_ < NG; ++g_) br[j_][g_] = brw[r32_ * 32 + 2 * (4 * g_ + jv) + (rho_ & 1)];         \
    }
    // groups: 0x008 matrix instruction, 0x020 VMEM read, 0x100 LDS read, 0x200 LDS write.  MF = matrix instructions of a stage (8 NG)
#define XB6_S1(vv, q) { XB_RDROW(R0, q, (q) & 1, 0) XB_COLH(vv, q, 0) XB_GREP(2 + NG, XB_G(0x008, 2) XB_G(0x100, 2)) XB_G(0x008, 8 * NG - 2 * (2 + NG)) } XB_SB()
#define XB6_S2(vv, q) { XB_RDROW(R1, q, (q) & 1, 2) XB_COLH(vv, q, 4) XB_GREP(2 + NG, XB_G(0x008, 2) XB_G(0x100, 2)) XB_G(0x008, 8 * NG - 2 * (2 + NG)) } XB_SB()
#define XB6_S3(LOAD, EXTRA, NEX) { LOAD EXTRA XB_ROWH(R0) XB_GREP(8 + (NEX), XB_G(0x008, 1) XB_G(0x020, 1)) XB_G(0x008, 8 * NG - 8 - (NEX)) } XB_SB()
#define XB6_S4(vo, qn, EXTRA, NEX) { XB_WIMG(vo, (qn) & 1) EXTRA XB_ROWH(R1) XB_GREP(8 + (NEX), XB_G(0x008, 1) XB_G(0x200, 1)) XB_G(0x008, 8 * NG - 8 - (NEX)) } XB_SB()
#define XB6_SUB(vv, vo, q, LOAD, E3, N3, E4, N4) XB6_S1(vv, q) XB6_S2(vv, q) XB6_S3(LOAD, E3, N3) XB6_S4(vo, (q) + 1, E4, N4)
#define XB_SUBBLOCK(vv, q, bufi) XB_COL(vv, q, bufi) XB_ROW(q, bufi)
    // The stream is pipelined ACROSS tiles: the first two sub-blocks of the next tile of the run (contiguous in the store) and its panel
    // rows are requested inside the last sub-block of this one -- one wave per SIMD holds two sub-blocks in flight, and a tile that
    // starts with a cold load exposes a full HBM latency per 8 phases of ~1.2 us (measured: 3.6 against 2.9 ms per sweep at 9.4e5 sites).
    XTile td; td.k = it.k0; td.w = it.w; td.mask = it.mask0; td.soff = it.soff0;
    double br[8][NG];
    RowOps R0, R1;
    RowOps1 Ra, Rb;
    if (it.t0 < it.t1) XB_LDBR(td.k)
    // row sums of a tile: D[i][j] of block n of Yr[mb][g]: row 16 mb + 4 n + i (i = rr, n = blk), vector 4 g + jv
#define XB_ROWSUMS()                                                                                                           \
    {                                                                                                                           \
        double *rp_ = rowpartB + (((size_t)td.k * nW + td.w) * XT_R + 4 * blk + rr) * so + jv;                                  \
        _Pragma("unroll") for (int g_ = 0; g_ < NG; ++g_) { rp_[4 * g_] = Yr[0][g_]; rp_[(size_t)16 * so + 4 * g_] = Yr[1][g_]; } \
    }
    int t = it.t0;
#pragma unroll 1
    while (t < it.t1) {
        if (td.mask != 0xffu) {
            // partial tile (a few per cent of the storage): one sub-block at a time
            double Yr[2][NG];
#pragma unroll
            for (int g = 0; g < NG; ++g) { Yr[0][g] = 0.0; Yr[1][g] = 0.0; }
            const dbl2 *base = reinterpret_cast<const dbl2 *>(tval + (size_t)(td.soff - sub_base) * XT_SUB) + lane;
            int sl = 0;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (variant != 12 && ((td.mask >> q) & 1u)) {
                    dbl2 vp[8];
                    XB_LD(vp, sl)
                    XB_SUBBLOCK(vp, q, q & 1)
                    ++sl;
                }
            }
            XB_ROWSUMS()
            ++t;
            if (t < it.t1) { td = tiles[t]; XB_LDBR(td.k) }
            continue;
        }
        // a chain of full tiles (64 KiB each, contiguous in the store): two sub-blocks in flight, primed once per chain and carried across
        // its tiles (a ring of four was measured: no faster, the registers are better spent on the operand pipeline above)
        dbl2 va[8], vb[8];
        {
            const dbl2 *base = reinterpret_cast<const dbl2 *>(tval + (size_t)(td.soff - sub_base) * XT_SUB) + lane;
            XB_LD(va, 0)
            XB_LD(vb, 1)
            if (PF) XB_WIMG(va, 0)
        }
        bool chain;
        dbl2 bn[4];
#pragma unroll 1
        do {
            XTile nxt = td;
            const bool more = t + 1 < it.t1;
            if (PF) nxt = tiles[min(t + 1, it.t1 - 1)];            // the tile itself at the end of the run
            else if (more) nxt = tiles[t + 1];
            chain = more && nxt.mask == 0xffu;
            double Yr[2][NG];
#pragma unroll
            for (int g = 0; g < NG; ++g) { Yr[0][g] = 0.0; Yr[1][g] = 0.0; }
            const dbl2 *base = reinterpret_cast<const dbl2 *>(tval + (size_t)(td.soff - sub_base) * XT_SUB) + lane;
            // (the stream registers of a sub-block are free once its image is written and its COLUMN sums are issued -- the row sums read the image --:
            // the loads of sub-block n + 2 go out there, a sub-block and a half ahead of their use instead of one: at 1.1 us of matrix work per
            // sub-block the loaded HBM latency of ~1.8 us was exposed on every sub-block)
            if (PF) {
                // No load under a condition in the body of a chain: hipcc's vmcnt counts must assume a conditional load was NOT issued, and the wait
                // for an older one then waits for it too (round-4 form: a full drain in sub-block 7 and at every tile start).  Past the end of a chain
                // the two slots re-read the first KiB of this tile (stride 0: cache hits, discarded)
                const dbl2 *base1 = base;
                const size_t lstr = chain ? 64 : 0;
#define XB_LDN(dst, slot) if (variant != 10) { _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) dst[j_] = NTL ? __builtin_nontemporal_load(base1 + (size_t)(8 * (slot) + j_) * lstr) : base1[(size_t)(8 * (slot) + j_) * lstr]; }
                XB6_SUB(va, vb, 0, XB_LD(va, 2), , 0, , 0)
                XB6_SUB(vb, va, 1, XB_LD(vb, 3), XB_LDBN(nxt.k), 4, , 0)
                XB6_SUB(va, vb, 2, XB_LD(va, 4), , 0, , 0)
                XB6_SUB(vb, va, 3, XB_LD(vb, 5), , 0, , 0)
                XB6_SUB(va, vb, 4, XB_LD(va, 6), , 0, , 0)
                XB6_SUB(vb, va, 5, XB_LD(vb, 7), , 0, XB_WBN(), 4)
                XB6_SUB(va, vb, 6, XB_LDN(va, 8), , 0, , 0)
                // the tile's last column sums are issued after S2: the next tile's panel rows come back from LDS inside S3 and S4
                XB6_S1(vb, 7) XB6_S2(vb, 7)
                { XB_LDN(vb, 9) XB_RDBRH(0) XB_ROWH(R0)
                  XB_GREP(4 * NG, XB_G(0x008, 1) XB_G(0x100, 1)) XB_GREP(8, XB_G(0x008, 1) XB_G(0x020, 1)) XB_G(0x008, 4 * NG - 8) } XB_SB()
                { XB_WIMG(va, 0) XB_RDBRH(4) XB_ROWH(R1)
                  XB_GREP(8, XB_G(0x008, 1) XB_G(0x200, 1)) XB_GREP(4 * NG, XB_G(0x008, 1) XB_G(0x100, 1)) XB_G(0x008, 4 * NG - 8) } XB_SB()
#undef XB_LDN
            } else {
                // round-4 form
                XB_COL(va, 0, 0) XB_LD(va, 2) XB_ROW(0, 0)
                XB_COL(vb, 1, 1) XB_LD(vb, 3) XB_ROW(1, 1)
                XB_COL(va, 2, 0) XB_LD(va, 4) XB_ROW(2, 0)
                XB_COL(vb, 3, 1) XB_LD(vb, 5) XB_ROW(3, 1)
                XB_COL(va, 4, 0) XB_LD(va, 6) XB_ROW(4, 0)
                XB_COL(vb, 5, 1) XB_LD(vb, 7) XB_ROW(5, 1)
                XB_COL(va, 6, 0)
                if (chain) { XB_LD(va, 8) }                                 // the next tile's first sub-block
                XB_ROW(6, 0)
                XB_COL(vb, 7, 1)
                if (more) XB_LDBR(nxt.k)                                    // this tile's column sums are issued: the panel registers are free
                if (chain) { XB_LD(vb, 9) }
                XB_ROW(7, 1)
            }
            XB_ROWSUMS()
            td = nxt;
            ++t;
        } while (chain);
    }
#undef XB_ROWSUMS
#undef XB_LD
#undef XB_LDBR
#undef XB_LDBN
#undef XB_WBN
#undef XB_COL
#undef XB_ROW
#undef XB_RD1
#undef XB_ROW1
#undef XB_COLQ
#undef XB_SB
#undef XB_RDROW
#undef XB_COLH
#undef XB_ROWH
#undef XB_SUBBLOCK
#undef XB_G
#undef XB_GREP
#undef XB_WIMG
#undef XB_RDBRH
#undef XB6_S1
#undef XB6_S2
#undef XB6_S3
#undef XB6_S4
#undef XB6_SUB
    // one record of column sums per workgroup: the four waves' accumulators are added in a fixed order through LDS, four sub-block
    // positions per round.  Yc[q][e][g] of lane (block n = blk, i = rr, j = jv) is column 32 q + 8 n + 2 i + e, vector 4 g + jv.
    double *rec = colpartB + (size_t)it.pad * XT_C * so;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        __syncthreads();                                                     // every wave is done with its images (and with the previous round)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4)
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int g = 0; g < NG; ++g) tsw[((q4 * 2 + e) * NG + g) * 64 + lane] = Yc[4 * h + q4][e][g];
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int idx = ((wv * 2 + e) * NG + g) * 64 + lane;
                const double sum = (ts[idx] + ts[2 * XT_SUB + idx]) + (ts[4 * XT_SUB + idx] + ts[6 * XT_SUB + idx]);
                rec[(size_t)(32 * (4 * h + wv) + 8 * blk + 2 * rr + e) * so + 4 * g + jv] = sum;
            }
    }
}

// ---- neighbour part Xs x panel ------------------------------------------------------------------------------------------------------
// Workgroups [0, 2 XB_DSPLIT): the two driver rows (thousands of entries each), XB_DSPLIT slices per row, partial sums to drvpart (finished
// by k_xtb_rows).  The rest: 16 atom rows per workgroup, 16 lanes per row (one per vector).  Non-S rows are finished (scaled) here; S rows
// leave their sparse sum in T for k_xtb_rows.
// Slab-distributed solve (rowlist != nullptr): the atom rows are the nlist rows of this rank's list; a driver row's partial sums run over the
// columns this rank OWNS (the drivers' own columns 0 / 1 count for rank 0), the owners' partials are added in rank order by k_xtb_rows_slab.
__global__ __launch_bounds__(XT_NT) void k_xtb_neigh(int m, const xrp_t *__restrict__ rp, const int *__restrict__ ci, const double *__restrict__ val,
                                                     const double *__restrict__ P, const double *__restrict__ sc, const int *__restrict__ nsrank,
                                                     const XCtrl *ctrl, double *__restrict__ T, double *__restrict__ drvpart,
                                                     const int *__restrict__ rowlist = nullptr, int nlist = 0, const int *__restrict__ dlist = nullptr, int nd0 = 0, int nd1 = 0)
{
    __shared__ double red[16][16];
    if (ctrl->done) return;
    const int v = threadIdx.x & 15, g = threadIdx.x >> 4;
    if (blockIdx.x < 2 * XB_DSPLIT) {
        const int row = blockIdx.x / XB_DSPLIT, part = blockIdx.x % XB_DSPLIT;
        const xrp_t p0 = rp[row], p1 = rp[row + 1];
        const xrp_t len = p1 - p0, a = p0 + len * part / XB_DSPLIT, b = p0 + len * (part + 1) / XB_DSPLIT;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        xrp_t p = a + g;
        if (dlist) {
            // slab-distributed solve: dlist = positions of the entries of row 0 (nd0 of them) and row 1 (nd1) whose column this rank owns
            const int n = row == 0 ? nd0 : nd1, l0 = row == 0 ? 0 : nd0;
            const int ia = l0 + (int)((long long)n * part / XB_DSPLIT), ib = l0 + (int)((long long)n * (part + 1) / XB_DSPLIT);
            int i = ia + g;
            for (; i + 48 < ib; i += 64) {
                const int q0 = dlist[i], q1 = dlist[i + 16], q2 = dlist[i + 32], q3 = dlist[i + 48];
                const int c0 = ci[q0], c1 = ci[q1], c2 = ci[q2], c3 = ci[q3];
                s0 += val[q0] * (sc[c0] * P[(size_t)c0 * XB_SP + v]); s1 += val[q1] * (sc[c1] * P[(size_t)c1 * XB_SP + v]);
                s2 += val[q2] * (sc[c2] * P[(size_t)c2 * XB_SP + v]); s3 += val[q3] * (sc[c3] * P[(size_t)c3 * XB_SP + v]);
            }
            for (; i < ib; i += 16) { const int q0 = dlist[i], c0 = ci[q0]; s0 += val[q0] * (sc[c0] * P[(size_t)c0 * XB_SP + v]); }
            p = b;
        }
        for (; p + 48 < b; p += 64) {
            const int c0 = ci[p], c1 = ci[p + 16], c2 = ci[p + 32], c3 = ci[p + 48];
            const double a0 = val[p], a1 = val[p + 16], a2 = val[p + 32], a3 = val[p + 48];
            s0 += a0 * (sc[c0] * P[(size_t)c0 * XB_SP + v]); s1 += a1 * (sc[c1] * P[(size_t)c1 * XB_SP + v]);
            s2 += a2 * (sc[c2] * P[(size_t)c2 * XB_SP + v]); s3 += a3 * (sc[c3] * P[(size_t)c3 * XB_SP + v]);
        }
        for (; p < b; p += 16) { const int c0 = ci[p]; s0 += val[p] * (sc[c0] * P[(size_t)c0 * XB_SP + v]); }
        red[g][v] = (s0 + s1) + (s2 + s3);
        __syncthreads();
        if (g == 0) {
            double s = 0.0;
#pragma unroll
            for (int u = 0; u < 16; ++u) s += red[u][v];
            drvpart[(row * XB_DSPLIT + part) * XB_SP + v] = s;
        }
        return;
    }
    // atom rows (<= nn + 3 entries): the 16 lanes of a row first fetch ALL its entries, up to four per lane (64 entries per trip), and fold the column's
    // scaling into the values; then, batch of 16 by batch of 16, every entry is handed round the group (shuffles) and the 16 panel reads of the batch
    // are issued together.  A latency-bound kernel (its bytes need 0.05 ms per sweep at 9.4e5 sites, it took 0.27): what counts is dependent memory
    // rounds per row x registers per wave.  Round 4 fetched the entries of a batch inside the batch loop -- two dependent rounds per batch; here the row
    // pointer, the entries and one round per batch, at the same registers.  The additions keep the entry order of the row: same bits.
    // XCD-aware order of the row blocks: workgroups b and b + 8 share an XCD (and its 4 MiB L2), so XCD x takes the x-th CONTIGUOUS eighth of the
    // row blocks -- neighbouring rows (atoms in structure order: neighbours in space) then re-read panel rows from their own L2 instead of each of
    // the eight L2s pulling the whole panel from the Infinity Cache (1.4 GB of 128-byte gathers per sweep at 9.4e5 sites)
    const int nb = (int)gridDim.x - 2 * XB_DSPLIT, b = (int)blockIdx.x - 2 * XB_DSPLIT;     // (2 XB_DSPLIT is a multiple of 8)
    const int xq = nb >> 3, xr = nb & 7, xc = b & 7;
    const int lb = xc * xq + min(xc, xr) + (b >> 3);
    const int li = lb * 16 + g;
    const int row = rowlist ? (li < nlist ? rowlist[li] : m) : 2 + li;
    const bool ok = row < m;
    const xrp_t p0 = ok ? rp[row] : 0, p1 = ok ? rp[row + 1] : 0;
    const int nsr = ok ? nsrank[row] : 0;
    const double scr = ok ? sc[row] : 0.0;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (xrp_t base = p0; base < p1; base += 64) {
        int cm[4]; double wm[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) { const xrp_t pe = base + 16 * b + v; cm[b] = pe < p1 ? ci[pe] : -1; wm[b] = pe < p1 ? val[pe] : 0.0; }
#pragma unroll
        for (int b = 0; b < 4; ++b) wm[b] = cm[b] >= 0 ? wm[b] * sc[cm[b]] : 0.0;
        // entry u of the batch to all 16 lanes of the row's group: DPP row_newbcast (a VALU move; __shfl would be an LDS ds_bpermute per value -- three
        // per entry, 1.6e7 of them per sweep at 9.4e5 sites: that, not the gathers, was what the kernel took its time for)
#define XN_BC(x_, u_) __builtin_amdgcn_update_dpp(0, (x_), 0x150 + (u_), 0xf, 0xf, false)
#define XN_GATHER(u_) { const int cu_ = XN_BC(cmb, u_); \
            x[u_] = cu_ >= 0 ? *reinterpret_cast<const double *>(reinterpret_cast<const char *>(P) + ((unsigned)cu_ * (unsigned)(XB_SP * 8) + (unsigned)(v * 8))) : 0.0; }
#define XN_W(u_) __hiloint2double(XN_BC(whi, u_), XN_BC(wlo, u_))
#define XN_ACC(u_) { s0 += XN_W(u_) * x[u_]; s1 += XN_W(u_ + 1) * x[u_ + 1]; s2 += XN_W(u_ + 2) * x[u_ + 2]; s3 += XN_W(u_ + 3) * x[u_ + 3]; }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            if (base + 16 * b >= p1) break;
            double x[16];
            const int cmb = cm[b], wlo = __double2loint(wm[b]), whi = __double2hiint(wm[b]);
            // (32-bit byte offsets from the panel's base: the panel is m x 128 B, far below 4 GB)
            XN_GATHER(0) XN_GATHER(1) XN_GATHER(2) XN_GATHER(3) XN_GATHER(4) XN_GATHER(5) XN_GATHER(6) XN_GATHER(7)
            XN_GATHER(8) XN_GATHER(9) XN_GATHER(10) XN_GATHER(11) XN_GATHER(12) XN_GATHER(13) XN_GATHER(14) XN_GATHER(15)
            XN_ACC(0) XN_ACC(4) XN_ACC(8) XN_ACC(12)
        }
#undef XN_BC
#undef XN_GATHER
#undef XN_W
#undef XN_ACC
    }
    const double s = (s0 + s1) + (s2 + s3);
    if (ok) T[(size_t)row * XB_SP + v] = nsr < 0 ? scr * s : s;
}

// ---- row kernel: partial sums -> S rows of T, then the partial Gram matrices of this workgroup's rows ------------------------------------
// Thread (r4 = tid / 16, v = tid % 16) owns vector v of S rows r4 and r4 + 16 of a row block; wave wv thus holds rows 4 wv ... 4 wv + 3
// (and + 16) in exactly the operand map of v_mfma_f64_16x16x4_f64 with the ROW on the k index: X'Z over four rows is one MFMA.
// INIT: T = A Y0 has just been formed; R = T - B is stored and only R'R is accumulated.
__device__ __forceinline__ double xtb_list_sum(const double *__restrict__ p, size_t stride, int first, int n, int step)
{
    // terms first, first + step, ... < n of a strided list; 16 loads in flight (the tail predicated, not looped); term j goes to
    // accumulator j % 4 whatever the batching
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    for (int c = first; c < n; c += 16 * step) {
        double x[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int cu = c + u * step; x[u] = cu < n ? p[(size_t)cu * stride] : 0.0; }
#pragma unroll
        for (int u = 0; u < 16; u += 4) { a0 += x[u]; a1 += x[u + 1]; a2 += x[u + 2]; a3 += x[u + 3]; }
    }
    return (a0 + a1) + (a2 + a3);
}
// two lists of the same shape (the thread's rows r4 and r4 + 16, `off` doubles apart; p points at the thread's vector v of row r4).  Loaded as 16-byte
// pairs: the even lane of a pair takes vectors (v, v + 1) of row r4, the odd one (v - 1, v) of row r4 + 16 -- twice the bytes per load instruction of
// the 8-byte form at the same number of loads in flight -- and one DPP exchange hands each lane the sum it does not hold.  Every (row, vector) list is
// still added by ONE thread in the same grouping (term j to accumulator j % 4): same bits as the 8-byte form.
__device__ __forceinline__ void xtb_list_sum2(const double *__restrict__ p, size_t stride, size_t off, int first, int n, int v, double &sa, double &sb)
{
    const bool odd = (v & 1) != 0;
    const dbl2 *__restrict__ q = reinterpret_cast<const dbl2 *>(odd ? p - 1 + off : p);
    const size_t st2 = stride >> 1;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, b0 = 0.0, b1 = 0.0, b2 = 0.0, b3 = 0.0;
    for (int c = first; c < n; c += 16) {
        dbl2 x[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int cu = c + u; x[u] = cu < n ? __builtin_nontemporal_load(q + (size_t)cu * st2) : (dbl2)(0.0); }      // (read once per sweep: non-temporal)
#pragma unroll
        for (int u = 0; u < 16; u += 4) { a0 += x[u].x; a1 += x[u + 1].x; a2 += x[u + 2].x; a3 += x[u + 3].x; b0 += x[u].y; b1 += x[u + 1].y; b2 += x[u + 2].y; b3 += x[u + 3].y; }
    }
    const double sx = (a0 + a1) + (a2 + a3), sy = (b0 + b1) + (b2 + b3);
    const double got = xor_lane<1>(odd ? sx : sy);
    sa = odd ? got : sx; sb = odd ? sy : got;
}
// ---- sharded solve: this rank's tile sums of every S row -> the exchange buffer xbuf[ns][so]; behind them the control words: rank 0's stop
// decision and any rank's abort word.  [w_lo, w_hi): windows that can hold this rank's partial sums (every other cell of its arrays is zero)
__global__ __launch_bounds__(XT_NT) void k_xtb_fold_local(int ns, int nK, int nW, int so, const int2 *__restrict__ wrange, const int *__restrict__ nitem_w,
                                                          const double *__restrict__ rowpartB, const double *__restrict__ colpartB, double *__restrict__ xbuf,
                                                          const XCtrl *ctrl, int flag_rank0, int w_lo, int w_hi, const long long *__restrict__ sdst = nullptr)
{
    // sdst (slab-distributed solve): S rank q's so tile sums go to xbuf[sdst[q]] -- the piece for the rank that owns the row (the control words are
    // written by k_xtb_slab_tail)
    if (ctrl->done) return;
    const int v = threadIdx.x & 15, r4 = threadIdx.x >> 4;
    const int rec_shift = nitem_w[2 * (nW + 2)];
    for (int k = blockIdx.x; k < nK; k += gridDim.x) {
        int2 wr = wrange[k];
        wr.x = max(wr.x, w_lo); wr.y = min(wr.y, w_hi);
        const int wk = k / (XT_C / XT_R);
        const bool mine = wk >= w_lo && wk < w_hi;
        const int nc = mine ? nitem_w[wk] >> rec_shift : 0, cbase = nitem_w[nW + 2 + wk] >> rec_shift;
        double tA = 0.0, tB = 0.0;
        if (v < so) {
            const double *rpp = rowpartB + ((size_t)k * nW * XT_R + r4) * so + v;
            const double *cpp = colpartB + ((size_t)cbase * XT_C + (XT_R * (k % (XT_C / XT_R)) + r4)) * so + v;
            const size_t rs = (size_t)XT_R * so, cs = (size_t)XT_C * so;
            double cA, cB, rA_, rB_;
            xtb_list_sum2(cpp, cs, (size_t)16 * so, 0, nc, v, cA, cB);
            xtb_list_sum2(rpp, rs, (size_t)16 * so, wr.x, wr.y, v, rA_, rB_);
            tA = cA + rA_; tB = cB + rB_;
            const int sA = XT_R * k + r4, sB = sA + 16;
            if (sA < ns) xbuf[(sdst ? (size_t)sdst[sA] : (size_t)sA * so) + v] = tA;
            if (sB < ns) xbuf[(sdst ? (size_t)sdst[sB] : (size_t)sB * so) + v] = tB;
        }
    }
    if (!sdst && blockIdx.x == 0 && threadIdx.x == 0) {
        xbuf[(size_t)ns * so] = (flag_rank0 && ctrl->done_local) ? 1.0 : 0.0;
        xbuf[(size_t)ns * so + 1] = ctrl->abort_local ? 1.0 : 0.0;
    }
}
__global__ void k_xtb_abort_word(XCtrl *ctrl, double *xbuf, size_t slot) { ctrl->abort_local = 1; xbuf[slot] = 1.0; }
__global__ void k_xtb_set_sharded(XCtrl *ctrl) { ctrl->sharded = 1; }

// SH: sharded solve -- xbuf holds, after the all-gather, one slot of ns so + 2 doubles per rank: that rank's tile sums of the S rows, its stop
// decision and its abort word.  Every rank adds the slots IN RANK ORDER: the same bits everywhere by construction (no transport decides the
// grouping of the additions, nothing has to be re-published from rank 0), and turns rank 0's stop decision / any rank's abort word into `done`
// here, in the same iteration.
// NF (split polynomial preconditioner, dkmc_set_x_poly): T is final in every row -- folded by k_xtb_fold_rows and carried through L -- this pass only forms R (INIT) and
// the partial Gram matrices.
template <int INIT, int SH = 0, int NF = 0>
__global__ __launch_bounds__(XT_NT) void k_xtb_rows(int ns, int nK, int nW, int m, int s, int so, const int2 *__restrict__ wrange, const int *__restrict__ nitem_w,
                                                    const double *__restrict__ rowpartB, const double *__restrict__ colpartB,
                                                    const int *__restrict__ srow, const double *__restrict__ sS, const int *__restrict__ nsrank,
                                                    const double *__restrict__ sc, const double *__restrict__ drvpart, double *__restrict__ T,
                                                    const double *__restrict__ P, double *__restrict__ R, const double *__restrict__ b,
                                                    double *__restrict__ gpart, XCtrl *ctrl, const double *__restrict__ xbuf, int it, int nr,
                                                    const XbAux *__restrict__ aux, const double *__restrict__ ax, const double *__restrict__ ay, const double *__restrict__ az)
{
    const size_t xslot = (size_t)ns * so + 2;
    __shared__ double lg[4][XB_NG / 2][4][64];                                 // the four waves' Gram accumulators, three matrices at a time (24 KiB)
    __shared__ int sdone;
    if (threadIdx.x == 0) {
        sdone = ctrl->done;
        if (SH && !sdone) {
            bool ab = false;
            for (int r = 0; r < nr; ++r) ab |= xbuf[r * xslot + (size_t)ns * so + 1] != 0.0;
            if (xbuf[(size_t)ns * so] != 0.0 || ab) {                         // rank 0's stop decision; anybody's abort word
                sdone = 1;
                // stops this iteration's step kernel too; the set-up pass (it = -1) must latch a NON-ZERO stamp as well
                if (blockIdx.x == 0) { ctrl->done = it + 1 > 0 ? it + 1 : 1; if (ab) ctrl->aborted = 1; }
            }
        }
    }
    __syncthreads();
    if (sdone) return;
    const int v = threadIdx.x & 15, r4 = threadIdx.x >> 4;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int rec_shift = nitem_w[2 * (nW + 2)];
    dbl4 G[XB_NG];
#pragma unroll
    for (int g = 0; g < XB_NG; ++g) G[g] = (dbl4)(0.0);
    // ---- S rows: fold the partial sums of this workgroup's row blocks ----
    for (int k = blockIdx.x; k < nK; k += gridDim.x) {
        const int2 wr = wrange[k];
        const int wk = k / (XT_C / XT_R);
        const int nc = nitem_w[wk] >> rec_shift, cbase = nitem_w[nW + 2 + wk] >> rec_shift;
        double tA = 0.0, tB = 0.0;
        const int sA = XT_R * k + r4, sB = sA + 16;
        if (SH) {
            if (v < so) {
                for (int r = 0; r < nr; ++r) {
                    if (sA < ns) tA += xbuf[r * xslot + (size_t)sA * so + v];
                    if (sB < ns) tB += xbuf[r * xslot + (size_t)sB * so + v];
                }
            }
        } else if (!NF && v < so) {
            const double *rpp = rowpartB + ((size_t)k * nW * XT_R + r4) * so + v;
            const double *cpp = colpartB + ((size_t)cbase * XT_C + (XT_R * (k % (XT_C / XT_R)) + r4)) * so + v;
            const size_t rs = (size_t)XT_R * so, cs = (size_t)XT_C * so;
            double cA, cB, rA_, rB_;
            xtb_list_sum2(cpp, cs, (size_t)16 * so, 0, nc, v, cA, cB);
            xtb_list_sum2(rpp, rs, (size_t)16 * so, wr.x, wr.y, v, rA_, rB_);
            tA = cA + rA_; tB = cB + rB_;
        }
        double pA = 0.0, pB = 0.0, rA = 0.0, rB = 0.0;
        if (sA < ns) {
            const int row = srow[sA]; const size_t o = (size_t)row * XB_SP + v;
            if (NF) tA = T[o]; else { tA = sS[sA] * (T[o] + tA); T[o] = tA; }
            if (INIT) { rA = tA - xtb_rhs(b, row, v, s, aux, ax, ay, az, sc); R[o] = rA; } else { pA = P[o]; rA = R[o]; }
        } else tA = 0.0;
        if (sB < ns) {
            const int row = srow[sB]; const size_t o = (size_t)row * XB_SP + v;
            if (NF) tB = T[o]; else { tB = sS[sB] * (T[o] + tB); T[o] = tB; }
            if (INIT) { rB = tB - xtb_rhs(b, row, v, s, aux, ax, ay, az, sc); R[o] = rB; } else { pB = P[o]; rB = R[o]; }
        } else tB = 0.0;
        if (!INIT) {
            G[0] = XB_MFMA(pA, tA, G[0]); G[1] = XB_MFMA(pA, rA, G[1]); G[2] = XB_MFMA(tA, rA, G[2]); G[3] = XB_MFMA(tA, tA, G[3]); G[5] = XB_MFMA(pA, pA, G[5]);
            G[0] = XB_MFMA(pB, tB, G[0]); G[1] = XB_MFMA(pB, rB, G[1]); G[2] = XB_MFMA(tB, rB, G[2]); G[3] = XB_MFMA(tB, tB, G[3]); G[5] = XB_MFMA(pB, pB, G[5]);
        }
        G[4] = XB_MFMA(rA, rA, G[4]); G[4] = XB_MFMA(rB, rB, G[4]);
    }
    // ---- the other rows of this workgroup's share of the vector (finished by k_xtb_neigh; the driver rows from their partial sums).
    // Four k-steps (16 rows) of a wave in flight: the loads do not wait for the S-rank test (every row below i1 is in bounds) ----
    {
        const int chunk = ((m + (int)gridDim.x - 1) / (int)gridDim.x + 3) & ~3;
        const int i0 = blockIdx.x * chunk, i1 = min(m, i0 + chunk);
        for (int r0 = i0 + 4 * wv; r0 < i1; r0 += 64) {
            double tv[4], pv[4], rv[4]; int sr[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int row = r0 + 16 * u + (lane >> 4);
                const bool in = row < i1;
                const size_t o = (size_t)(in ? row : i0) * XB_SP + v;
                sr[u] = in ? nsrank[row] : 0;
                tv[u] = T[o];
                if (!INIT) { pv[u] = P[o]; rv[u] = R[o]; }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int row = r0 + 16 * u + (lane >> 4);
                if (r0 + 16 * u >= i1) break;                                 // uniform over the wave
                const bool use = row < i1 && sr[u] < 0;
                const size_t o = (size_t)row * XB_SP + v;
                double t_ = use ? tv[u] : 0.0, p_ = 0.0, r_ = 0.0;
                if (!NF && use && row < 2) {
                    double sd = 0.0;
#pragma unroll
                    for (int w = 0; w < XB_DSPLIT; ++w) sd += drvpart[(row * XB_DSPLIT + w) * XB_SP + v];
                    t_ = sc[row] * sd; T[o] = t_;
                }
                if (INIT) { if (use) { r_ = t_ - xtb_rhs(b, row, v, s, aux, ax, ay, az, sc); R[o] = r_; } }
                else if (use) { p_ = pv[u]; r_ = rv[u]; }
                if (!INIT) { G[0] = XB_MFMA(p_, t_, G[0]); G[1] = XB_MFMA(p_, r_, G[1]); G[2] = XB_MFMA(t_, r_, G[2]); G[3] = XB_MFMA(t_, t_, G[3]); G[5] = XB_MFMA(p_, p_, G[5]); }
                G[4] = XB_MFMA(r_, r_, G[4]);
            }
        }
    }
    // ---- one partial per workgroup: the four waves added in a fixed order; entry (register u, lane l) is (i = l / 16 + 4 u, j = l % 16) ----
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        __syncthreads();
#pragma unroll
        for (int g = 0; g < XB_NG / 2; ++g)
#pragma unroll
            for (int u = 0; u < 4; ++u) lg[wv][g][u][lane] = G[h * (XB_NG / 2) + g][u];
        __syncthreads();
        for (int e = threadIdx.x; e < (XB_NG / 2) * 256; e += XT_NT) {
            const int g = e >> 8, u = (e >> 6) & 3, l = e & 63;
            gpart[(size_t)blockIdx.x * (XB_NG * 256) + h * (XB_NG / 2) * 256 + e] = (lg[0][g][u][l] + lg[1][g][u][l]) + (lg[2][g][u][l] + lg[3][g][u][l]);
        }
    }
}

// ---- split polynomial preconditioner (dkmc_set_x_poly(d), one GPU) -------------------------------------------------------------------------
// The block loop runs on L A L, A = S X S the Jacobi-scaled operator, L = sum_j c_j N^j the degree-d truncation of the series of (I - N)^(-1/2),
// N = I - An, An = the neighbour part of A with A's full (unit) diagonal -- the couplings of the two driver nodes stay outside N.  CG on A
// preconditioned with An EXACTLY needs 8 iterations where A needs 666 (85 k sites, tools/precond_proto.py): the ill-conditioning of X lives in its
// sparse part, and 2 d sparse panel products per sweep buy 2-3x fewer passes over the tiles (tools/precond_block_proto.py).  The loop's algebra
// is untouched: it sees another SPD operator.  A start vector y0 enters as the right-hand side: A d = b - A y0, d = L dh, dh from zero.
// out = ca * add + cb * (N in): 16 lanes per row as in k_xtb_neigh; rows 0 / 1 (driver nodes) and their columns take no part in N
__global__ __launch_bounds__(XT_NT) void k_xtb_nmul(int m, const xrp_t *__restrict__ rp, const int *__restrict__ ci, const double *__restrict__ val,
                                                    const double *__restrict__ sc, const double *__restrict__ in, const double *__restrict__ add,
                                                    double ca, double cb, double *__restrict__ out, const XCtrl *ctrl)
{
    if (ctrl->done) return;
    const int v = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int nb = (int)gridDim.x, b = (int)blockIdx.x;
    const int xq = nb >> 3, xr = nb & 7, xc = b & 7;
    const int row = (xc * xq + min(xc, xr) + (b >> 3)) * 16 + g;               // XCD-contiguous row blocks (see k_xtb_neigh)
    const bool ok = row < m, atom = ok && row >= 2;
    const xrp_t p0 = atom ? rp[row] : 0, p1 = atom ? rp[row + 1] : 0;
    const double scr = ok ? sc[row] : 0.0;
    const double av = ok ? add[(size_t)row * XB_SP + v] : 0.0;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (xrp_t base = p0; base < p1; base += 64) {
        int cm[4]; double wm[4];
#pragma unroll
        for (int bq = 0; bq < 4; ++bq) { const xrp_t pe = base + 16 * bq + v; int c = pe < p1 ? ci[pe] : -1; if (c < 2 || c == row) c = -1; cm[bq] = c; wm[bq] = c >= 0 ? val[pe] : 0.0; }
#pragma unroll
        for (int bq = 0; bq < 4; ++bq) wm[bq] = cm[bq] >= 0 ? wm[bq] * sc[cm[bq]] : 0.0;
#define XN_BC(x_, u_) __builtin_amdgcn_update_dpp(0, (x_), 0x150 + (u_), 0xf, 0xf, false)
#define XN_GATHER(u_) { const int cu_ = XN_BC(cmb, u_); \
            x[u_] = cu_ >= 0 ? *reinterpret_cast<const double *>(reinterpret_cast<const char *>(in) + ((unsigned)cu_ * (unsigned)(XB_SP * 8) + (unsigned)(v * 8))) : 0.0; }
#define XN_W(u_) __hiloint2double(XN_BC(whi, u_), XN_BC(wlo, u_))
#define XN_ACC(u_) { s0 += XN_W(u_) * x[u_]; s1 += XN_W(u_ + 1) * x[u_ + 1]; s2 += XN_W(u_ + 2) * x[u_ + 2]; s3 += XN_W(u_ + 3) * x[u_ + 3]; }
#pragma unroll
        for (int bq = 0; bq < 4; ++bq) {
            if (base + 16 * bq >= p1) break;
            double x[16];
            const int cmb = cm[bq], wlo = __double2loint(wm[bq]), whi = __double2hiint(wm[bq]);
            XN_GATHER(0) XN_GATHER(1) XN_GATHER(2) XN_GATHER(3) XN_GATHER(4) XN_GATHER(5) XN_GATHER(6) XN_GATHER(7)
            XN_GATHER(8) XN_GATHER(9) XN_GATHER(10) XN_GATHER(11) XN_GATHER(12) XN_GATHER(13) XN_GATHER(14) XN_GATHER(15)
            XN_ACC(0) XN_ACC(4) XN_ACC(8) XN_ACC(12)
        }
#undef XN_BC
#undef XN_GATHER
#undef XN_W
#undef XN_ACC
    }
    if (ok) out[(size_t)row * XB_SP + v] = ca * av - cb * (scr * ((s0 + s1) + (s2 + s3)));
}
// QS (the compact, interleaved copy of the S rows the tile kernel reads) of an arbitrary panel
__global__ void k_xtb_qs_from(int m, const double *__restrict__ V, const double *__restrict__ sc, const int *__restrict__ nsrank, double *__restrict__ QS, const XCtrl *ctrl)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m * XB_SP || ctrl->done) return;
    const int row = i >> 4, v = i & 15;
    const int sr = nsrank[row];
    if (sr >= 0) QS[xtb_qs_pos(sr, v)] = sc[row] * V[i];
}
// the fold of k_xtb_rows alone: S rows of T <- scaling x (sparse sum + tile sums), driver rows from their partial sums
__global__ __launch_bounds__(XT_NT) void k_xtb_fold_rows(int ns, int nK, int nW, int so, const int2 *__restrict__ wrange, const int *__restrict__ nitem_w,
                                                         const double *__restrict__ rowpartB, const double *__restrict__ colpartB, const int *__restrict__ srow,
                                                         const double *__restrict__ sS, const double *__restrict__ sc, const double *__restrict__ drvpart,
                                                         double *__restrict__ T, const XCtrl *ctrl)
{
    if (ctrl->done) return;
    const int v = threadIdx.x & 15, r4 = threadIdx.x >> 4;
    const int rec_shift = nitem_w[2 * (nW + 2)];
    if (blockIdx.x == 0 && threadIdx.x < 32) {
        const int row = threadIdx.x >> 4;
        double sd = 0.0;
#pragma unroll
        for (int w = 0; w < XB_DSPLIT; ++w) sd += drvpart[(row * XB_DSPLIT + w) * XB_SP + v];
        T[(size_t)row * XB_SP + v] = sc[row] * sd;
    }
    for (int k = blockIdx.x; k < nK; k += gridDim.x) {
        const int2 wr = wrange[k];
        const int wk = k / (XT_C / XT_R);
        const int nc = nitem_w[wk] >> rec_shift, cbase = nitem_w[nW + 2 + wk] >> rec_shift;
        double tA = 0.0, tB = 0.0;
        const int sA = XT_R * k + r4, sB = sA + 16;
        if (v < so) {
            const double *rpp = rowpartB + ((size_t)k * nW * XT_R + r4) * so + v;
            const double *cpp = colpartB + ((size_t)cbase * XT_C + (XT_R * (k % (XT_C / XT_R)) + r4)) * so + v;
            const size_t rs = (size_t)XT_R * so, cs = (size_t)XT_C * so;
            double cA, cB, rA_, rB_;
            xtb_list_sum2(cpp, cs, (size_t)16 * so, 0, nc, v, cA, cB);
            xtb_list_sum2(rpp, rs, (size_t)16 * so, wr.x, wr.y, v, rA_, rB_);
            tA = cA + rA_; tB = cB + rB_;
        }
        if (sA < ns) { const size_t o = (size_t)srow[sA] * XB_SP + v; T[o] = sS[sA] * (T[o] + tA); }
        if (sB < ns) { const size_t o = (size_t)srow[sB] * XB_SP + v; T[o] = sS[sB] * (T[o] + tB); }
    }
}
// start of a preconditioned solve: W <- [T(:, 0) - b | 0 ... 0] (T = A Y0: the residual of the start vector, sign r = A y - b)
__global__ void k_xtb_pre_resid(int m, const double *__restrict__ T, const double *__restrict__ b, double *__restrict__ W)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m * XB_SP) return;
    W[i] = (i & 15) == 0 ? T[i] - b[i >> 4] : 0.0;
}
// W <- [y | 0 ... 0]
__global__ void k_xtb_pre_col0(int m, const double *__restrict__ y, double *__restrict__ W)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m * XB_SP) return;
    W[i] = (i & 15) == 0 ? y[i >> 4] : 0.0;
}
// end of a preconditioned solve: y <- y + Z(:, 0)  (Z = L dh, the correction)
__global__ void k_xtb_pre_add(int m, const double *__restrict__ Z, double *__restrict__ y)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) y[i] += Z[(size_t)i * XB_SP];
}
// ||T(:, 0) - b||^2 in one workgroup (fixed order): the TRUE residual of the unpreconditioned scaled system, for the stop test a caller relies on
__global__ __launch_bounds__(1024) void k_xtb_pre_rr(int m, const double *__restrict__ T, const double *__restrict__ b, double *__restrict__ out)
{
    __shared__ double red[1024];
    double a = 0.0;
    for (int i = threadIdx.x; i < m; i += 1024) { const double d = T[(size_t)i * XB_SP] - b[i]; a += d * d; }
    red[threadIdx.x] = a;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) out[0] = red[0];
}

// ---- Gram matrices: the row kernel's partials, reduced in a fixed order ----------------------------------------------------------------
// 16 entries per workgroup, 16 slices of the partial list per entry (slice q adds partials q, q + 16, ...), the slices combined in order.
__global__ __launch_bounds__(XT_NT) void k_xtb_gred(int npart, const double *__restrict__ gpart, double *__restrict__ gfin, const XCtrl *ctrl)
{
    __shared__ double red[16][17];
    if (ctrl->done) return;
    const int el = threadIdx.x & 15, q = threadIdx.x >> 4;
    const int e = 16 * (int)blockIdx.x + el;
    red[q][el] = xtb_list_sum(gpart + e, (size_t)XB_NG * 256, q, npart, 16);
    __syncthreads();
    if (q == 0) {
        double a = 0.0;
#pragma unroll
        for (int u = 0; u < 16; ++u) a += red[u][el];
        gfin[e] = a;
    }
}

// ---- the s x s algebra of one iteration: ONE wave (no workgroup barriers; the matrices live in LDS, entry (i, j) = e / 16, e % 16 of lane
// e % 64) -----------------------------------------------------------------------------------------------------------------------------------
// mats: c | M1 = -W | M2 = beta W | M3 = c M1, each 16 x 16 row-major, zero outside s x s.  it = -1: set-up (P = 0, R = A Y0 - B): only
// R'R is read; c = M2 = M3 = 0, W from R'R, first stop test on ||r|| (iterative_solvers_gpu.cu:418), later ones on ||r||^2 (:448).
#define XB_M(name) double (*name)[17]
#define XB_WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
__device__ __forceinline__ void xtb_mm(XB_M(C), XB_M(A), XB_M(B), bool ta, double alpha, XB_M(D), double delta)
{
    // C = alpha op(A) B + delta D; op = transpose if ta; D may be null; C may alias A, B or D.  All matrices are zero outside their
    // leading s x s block, so the loops run over the full 16 (compile-time bounds: the 80 LDS reads of a lane are issued together).
    // Lane l owns entries (l / 16 + 4 u, l % 16), u = 0 ... 3: one column index, four rows.
    const int j = threadIdx.x & 15, i0 = threadIdx.x >> 4;
    double bcol[16], acc[4];
#pragma unroll
    for (int k = 0; k < 16; ++k) bcol[k] = B[k][j];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int i = i0 + 4 * u;
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int k = 0; k < 16; k += 2) { a0 += (ta ? A[k][i] : A[i][k]) * bcol[k]; a1 += (ta ? A[k + 1][i] : A[i][k + 1]) * bcol[k + 1]; }
        acc[u] = alpha * (a0 + a1) + (D ? delta * D[i][j] : 0.0);
    }
    XB_WSYNC();
#pragma unroll
    for (int u = 0; u < 4; ++u) C[i0 + 4 * u][j] = acc[u];
    XB_WSYNC();
}
__device__ __forceinline__ void xtb_symmetrise(XB_M(A), int s)
{
    double acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int e = (int)threadIdx.x + 64 * u, i = e >> 4, j = e & 15; acc[u] = (i < s && j < s) ? 0.5 * (A[i][j] + A[j][i]) : 0.0; }
    XB_WSYNC();
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int e = (int)threadIdx.x + 64 * u; A[e >> 4][e & 15] = acc[u]; }
    XB_WSYNC();
}
// in-place Cholesky of the leading s x s block (lower triangle); false on a non-positive pivot (uniform: every lane reads the same pivots)
__device__ __forceinline__ bool xtb_chol(XB_M(A), int s)
{
    bool ok = true;
    if ((int)threadIdx.x >= s && threadIdx.x < 16) A[threadIdx.x][threadIdx.x] = 1.0;      // padding: unit diagonal (the rest of it is zero)
    XB_WSYNC();
    for (int k = 0; k < s; ++k) {
        const double d = A[k][k];
        if (!(d > 0.0)) ok = false;
        const double ld = sqrt(d > 0.0 ? d : 1.0);
        XB_WSYNC();
        if ((int)threadIdx.x == k) A[k][k] = ld;
        else if ((int)threadIdx.x > k && (int)threadIdx.x < s) A[threadIdx.x][k] = A[threadIdx.x][k] / ld;
        XB_WSYNC();
        const int j = threadIdx.x & 15, i0 = threadIdx.x >> 4;
        const double ljk = A[j][k];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + 4 * u;
            if (i > k && j > k && j <= i && i < s) A[i][j] -= A[i][k] * ljk;
        }
        XB_WSYNC();
    }
    return ok;
}
// (padding rows / columns of L beyond s carry a unit diagonal, see xtb_chol: the loops below run over the full 16 with compile-time bounds)
__device__ __forceinline__ void xtb_chol_solve(XB_M(X), XB_M(L), XB_M(B), int s)
{
    const int j = threadIdx.x & 15;
    double x[16];
    if (threadIdx.x < 16) {
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = B[i][j];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            double a = x[i];
#pragma unroll
            for (int p = 0; p < i; ++p) a -= L[i][p] * x[p];
            x[i] = a / L[i][i];
        }
#pragma unroll
        for (int i = 15; i >= 0; --i) {
            double a = x[i];
#pragma unroll
            for (int p = i + 1; p < 16; ++p) a -= L[p][i] * x[p];
            x[i] = a / L[i][i];
        }
    }
    XB_WSYNC();                                                               // X may alias B
    if (threadIdx.x < 16) {
#pragma unroll
        for (int i = 0; i < 16; ++i) X[i][j] = (i < s && j < s) ? x[i] : 0.0;
    }
    XB_WSYNC();
}
// nrg > 1 (slab-distributed solve): gfin holds one block of gstride doubles per rank -- that rank's partial Gram matrices and, behind them, its
// abort word -- all-gathered; every rank adds the blocks IN RANK ORDER: identical matrices, hence identical decisions, on every rank.
// Launched with 256 threads: all four waves load (and, nrg > 1, add) the Gram matrices into LDS, then wave 0 does the algebra alone (one wave
// summing 24 x nrg dependent loads took 73 us at nrg = 8).
__global__ __launch_bounds__(256) void k_xtb_small(int it, int s, const double *__restrict__ gfin, double *__restrict__ mats, XCtrl *ctrl, double tol2,
                                                   int nrg = 1, int gstride = 0)
{
    __shared__ double Gm[XB_NG][16][17], Lp[16][17], Cm[16][17], Vm[16][17], Bm[16][17], Grn[16][17], Gprn[16][17], Gg[16][17], Wm[16][17], Tm[16][17];
    if (ctrl->done) return;
    if (nrg > 1) {
        bool ab = false;
        for (int r = 0; r < nrg; ++r) ab |= gfin[(size_t)r * gstride + XB_NG * 256 + 1] != 0.0;
        if (ab) { if (threadIdx.x == 0) { ctrl->aborted = 1; ctrl->done = it + 1 > 0 ? it + 1 : 1; } return; }      // uniform: every lane read the same words
    }
    // Gram matrices: entry e = (register u, lane l) of the row kernel's accumulators is (l / 16 + 4 u, l % 16); thread t of the workgroup
    // takes (u = t / 64, l = t % 64) of every matrix, the ranks' blocks added in rank order (four loads in flight)
    {
        const int u = threadIdx.x >> 6, l = threadIdx.x & 63;
#pragma unroll
        for (int g = 0; g < XB_NG; ++g) {
            const double *src = gfin + g * 256 + u * 64 + l;
            double a = src[0];
            int r = 1;
            for (; r + 3 < nrg; r += 4) {
                const double x0 = src[(size_t)r * gstride], x1 = src[(size_t)(r + 1) * gstride], x2 = src[(size_t)(r + 2) * gstride], x3 = src[(size_t)(r + 3) * gstride];
                a += x0; a += x1; a += x2; a += x3;
            }
            for (; r < nrg; ++r) a += src[(size_t)r * gstride];
            Gm[g][(l >> 4) + 4 * u][l & 15] = a;
        }
    }
    __syncthreads();
    if (threadIdx.x >= 64) return;
    const bool init = it < 0;
    double rr_new;
    if (init) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = (int)threadIdx.x + 64 * u, i = e >> 4, j = e & 15;
            Cm[i][j] = 0.0; Bm[i][j] = 0.0;
            const double g = (i < s && j < s) ? 0.5 * (Gm[4][i][j] + Gm[4][j][i]) : 0.0;
            Grn[i][j] = g; Gg[i][j] = g;
        }
        XB_WSYNC();
        rr_new = Grn[0][0];
    } else {
        // P'T is symmetric by construction of A; symmetrise its rounding.  Tm keeps it, Lp becomes its Cholesky factor
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = (int)threadIdx.x + 64 * u, i = e >> 4, j = e & 15;
            const double g = (i < s && j < s) ? 0.5 * (Gm[0][i][j] + Gm[0][j][i]) : 0.0;
            Lp[i][j] = g; Tm[i][j] = g;
        }
        XB_WSYNC();
        if (!xtb_chol(Lp, s)) {                                               // no step can be taken: stop BEFORE this iteration's update (uniform)
            // (sharded: every rank holds the same Gram matrices -- the all-reduce hands all of them the same bits -- and takes this branch together)
            if (threadIdx.x == 0) { ctrl->pad[0] = 1; ctrl->done = it + 1; }
            return;
        }
        xtb_chol_solve(Cm, Lp, Gm[1], s);                                     // (P'T)^-1 P'R
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int e = (int)threadIdx.x + 64 * u; Cm[e >> 4][e & 15] = -Cm[e >> 4][e & 15]; }     // c = -(...)
        XB_WSYNC();
        xtb_mm(Vm, Gm[3], Cm, false, 1.0, Gm[2], 1.0);                     // V = T'T c + T'R      (= T'R+)
        xtb_chol_solve(Bm, Lp, Vm, s);                                        // beta
        // R+'R+ = R'R + c'T'R + (c'T'R)' + c'(T'T c) = R'R + c'V + (T'R)'c
        xtb_mm(Grn, Cm, Vm, true, 1.0, Gm[4], 1.0);
        xtb_mm(Grn, Gm[2], Cm, true, 1.0, Grn, 1.0);
        xtb_symmetrise(Grn, s);
        rr_new = Grn[0][0];
        // Gram matrix of D = -R+ + P beta: R+'R+ - beta'(P'R+) - (P'R+)'beta + beta'(P'P)beta, P'R+ = P'R + P'T c.  P'P is MEASURED (it is
        // the identity only as far as the previous orthonormalisation was exact), so the recurrence holds for whatever P the loop carries
        xtb_mm(Gprn, Tm, Cm, false, 1.0, Gm[1], 1.0);
        xtb_symmetrise(Gm[5], s);
        xtb_mm(Gg, Gm[5], Bm, false, 1.0, nullptr, 0.0);                      // (P'P) beta
        xtb_mm(Gg, Bm, Gg, true, 1.0, Grn, 1.0);
        xtb_mm(Gg, Bm, Gprn, true, -1.0, Gg, 1.0);
        xtb_mm(Gg, Gprn, Bm, true, -1.0, Gg, 1.0);
        xtb_symmetrise(Gg, s);
    }
    const bool stop = init ? !(sqrt(rr_new) > tol2) : !(rr_new > tol2);
    // W = L^-T of the Gram matrix's Cholesky factor: W' G W = I.  Close to convergence G can lose definiteness in its rounding (its
    // columns become dependent): the directions are then only scaled to unit length this iteration (W diagonal) -- P'P is measured, so
    // nothing downstream assumes more.  Only a non-positive DIAGONAL ends the block loop (the update of Y of this iteration stands).
    bool diag_ok = true;
    for (int k = 0; k < s; ++k) if (!(Gg[k][k] > 0.0)) diag_ok = false;
    double gdiag = (threadIdx.x < 16 && (int)threadIdx.x < s && diag_ok) ? Gg[threadIdx.x][threadIdx.x] : 1.0;
    XB_WSYNC();
    const bool cholg = xtb_chol(Gg, s);
    const bool okg = diag_ok;
    if (!cholg) {                                                             // uniform: Gg <- diag(sqrt(g_ii)) stands in for the factor
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int e = (int)threadIdx.x + 64 * u, i = e >> 4, j = e & 15; Gg[i][j] = (i == j) ? 1.0 : 0.0; }
        XB_WSYNC();
        if (threadIdx.x < 16) Gg[threadIdx.x][threadIdx.x] = sqrt(gdiag);
        XB_WSYNC();
    }
    // W = (L')^-1: column j of W solves L' w = e_j (upper triangular), by lane j
    if (threadIdx.x < 16) {
        const int jj = threadIdx.x;
        double w[16];
#pragma unroll
        for (int r = 15; r >= 0; --r) {
            double a = (r == jj) ? 1.0 : 0.0;
#pragma unroll
            for (int p = r + 1; p < 16; ++p) a -= Gg[p][r] * w[p];
            w[r] = a / Gg[r][r];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) Wm[r][jj] = (r < s && jj < s && okg) ? w[r] : 0.0;
    }
    XB_WSYNC();
    xtb_mm(Vm, Bm, Wm, false, 1.0, nullptr, 0.0);                          // M2 = beta W
    xtb_mm(Tm, Cm, Wm, false, -1.0, nullptr, 0.0);                         // M3 = c M1 = -c W
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int e = (int)threadIdx.x + 64 * u, i = e >> 4, j = e & 15;
        mats[0 * 256 + e] = Cm[i][j]; mats[1 * 256 + e] = -Wm[i][j]; mats[2 * 256 + e] = Vm[i][j]; mats[3 * 256 + e] = Tm[i][j];
    }
    if (threadIdx.x == 0) {
        ctrl->rr[(it + 1) & 1] = rr_new; ctrl->rr[it & 1] = rr_new;
        ctrl->iters = it + 1;
        if (!okg) ctrl->pad[0] = 2;
        else if (!cholg) ctrl->pad[1] += 1;                                   // iterations that only rescaled their directions (statistics)
        // the step kernel of this iteration still runs (see k_xt_step).  Sharded solve: only the local flag is set; rank 0's travels in the next
        // exchange and k_xtb_rows turns it into `done` on every rank in the same iteration (one wasted product per solve)
        if (stop || !okg) { if (ctrl->sharded) ctrl->done_local = 1; else ctrl->done = it + 2; }
    }
}

// ---- panel updates: Y += P c ; R += T c ; P = R M1 + T M3 + P M2 ; Q = S P -------------------------------------------------------------
// 16 rows per wave and step: the three panels as A operands (row on i, vector on k), the 16 x 16 matrices as B operands, Y and R as
// accumulator input.  P is formed from the ORIGINAL R, T, P (M3 = c M1), so no result has to change its register map.
// rowlist (slab-distributed solve): the rows are the m entries of this rank's list (the two driver rows, replicated, + the rows it owns).
__global__ __launch_bounds__(XT_NT) void k_xtb_step(int m, int it, const double *__restrict__ mats, double *__restrict__ y0, double *__restrict__ R,
                                                    double *__restrict__ P, const double *__restrict__ T, const double *__restrict__ sc,
                                                    const int *__restrict__ nsrank, double *__restrict__ QS, const XCtrl *ctrl,
                                                    const int *__restrict__ rowlist = nullptr, double *__restrict__ Ypanel = nullptr)
{
    __shared__ int sdone;
    if (threadIdx.x == 0) { const int d = ctrl->done; sdone = d != 0 && it + 1 >= d; }
    __syncthreads();
    if (sdone) return;
    const int lane = threadIdx.x & 63, a = lane & 15, b = lane >> 4;
    const int wv = threadIdx.x >> 6;
    double cB[4], m1B[4], m2B[4], m3B[4], c0[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const int o = (4 * kk + b) * 16 + a;
        cB[kk] = mats[o]; m1B[kk] = mats[256 + o]; m2B[kk] = mats[512 + o]; m3B[kk] = mats[768 + o];
        c0[kk] = mats[(4 * kk + b) * 16];                                     // column 0 of c: the physical column of Y += P c
    }
    const int ngroups = (m + 15) / 16;
    for (int g = (int)blockIdx.x * 4 + wv; g < ngroups; g += (int)gridDim.x * 4) {
        const int row0 = 16 * g;
        double pa[4], ta[4], ra[4];
        const bool oka = row0 + a < m;
        const int rowa = oka ? (rowlist ? rowlist[row0 + a] : row0 + a) : 0;
        const size_t oa = (size_t)rowa * XB_SP + b;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) { pa[kk] = oka ? P[oa + 4 * kk] : 0.0; ta[kk] = oka ? T[oa + 4 * kk] : 0.0; ra[kk] = oka ? R[oa + 4 * kk] : 0.0; }
        dbl4 rN, pN = (dbl4)(0.0), yN = (dbl4)(0.0);
        const double yold = (b == 0 && oka) ? y0[rowa] : 0.0;
        int rowu[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int li = row0 + b + 4 * u;
            rowu[u] = li < m ? (rowlist ? rowlist[li] : li) : -1;
            rN[u] = rowu[u] >= 0 ? R[(size_t)rowu[u] * XB_SP + a] : 0.0;
            if (Ypanel) yN[u] = rowu[u] >= 0 ? Ypanel[(size_t)rowu[u] * XB_SP + a] : 0.0;
        }
        // y0[row] += sum_i P[row][i] c[i][0]: lane (a, b) holds i = 4 kk + b; the four b-groups are added in a fixed order
        double yacc = (pa[0] * c0[0] + pa[1] * c0[1]) + (pa[2] * c0[2] + pa[3] * c0[3]);
        yacc += __shfl_xor(yacc, 16, WAVE); yacc += __shfl_xor(yacc, 32, WAVE);
        if (b == 0 && oka) y0[rowa] = yold + yacc;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            rN = XB_MFMA(ta[kk], cB[kk], rN);
            pN = XB_MFMA(ra[kk], m1B[kk], pN); pN = XB_MFMA(ta[kk], m3B[kk], pN); pN = XB_MFMA(pa[kk], m2B[kk], pN);
            if (Ypanel) yN = XB_MFMA(pa[kk], cB[kk], yN);                  // Y += P c, all columns (the auxiliary solutions are kept for the next solve)
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = rowu[u];
            if (row >= 0) {
                const size_t o = (size_t)row * XB_SP + a;
                R[o] = rN[u]; P[o] = pN[u];
                if (Ypanel) Ypanel[o] = yN[u];
                const int sr = nsrank[row];
                if (sr >= 0) QS[xtb_qs_pos(sr, a)] = sc[row] * pN[u];
            }
        }
    }
}
// before the first product: P = Y0 stands in for Y0 (the product below is A Y0), QS = S Y0 over S; y0 = column 0.  Column 0 starts from y;
// the auxiliary columns from zero, or -- yaux != nullptr: columns hs ... s - 1, the hash set -- from the UNSCALED solutions the previous solve left
// (their right-hand sides are the same in every solve): their residuals then carry what the previous solve had not yet resolved, which is what
// the block Krylov space of this one should contain (tools/warm_aux_proto.py).  Ypanel (may be null): the scaled block iterate Y, all columns.
__global__ void k_xtb_init(int m, const double *__restrict__ y, const double *__restrict__ sc, const int *__restrict__ nsrank,
                           double *__restrict__ y0, double *__restrict__ P, double *__restrict__ QS,
                           const double *__restrict__ yaux = nullptr, int hs = 16, int s = 16, double *__restrict__ Ypanel = nullptr)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m * XB_SP) return;
    const int row = i >> 4, v = i & 15;
    double yv = 0.0;
    if (v == 0) yv = y[row];
    else if (yaux && v >= hs && v < s) yv = yaux[i] / sc[row];
    P[i] = yv;
    if (Ypanel) Ypanel[i] = yv;
    if (v == 0) y0[row] = yv;
    const int sr = nsrank[row];
    if (sr >= 0) QS[xtb_qs_pos(sr, v)] = sc[row] * yv;
}
// the auxiliary solutions a solve leaves for the next one: unscaled (x = S y)
__global__ void k_xtb_yaux_out(int n, const int *__restrict__ rowlist, const double *__restrict__ Ypanel, const double *__restrict__ sc, double *__restrict__ yaux)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * XB_SP) return;
    const int row = rowlist ? rowlist[i >> 4] : (i >> 4), v = i & 15;
    yaux[(size_t)row * XB_SP + v] = Ypanel[(size_t)row * XB_SP + v] * sc[row];
}
__global__ void k_xtb_zero(long long n, double *__restrict__ p)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0.0;
}

// Second stream of the block loop on large systems: the neighbour part (an L2 / Infinity-Cache gather, 0.3 ms per sweep at 9.4e5 sites) runs
// BESIDE the tile x panel kernel (one wave per SIMD on the matrix pipe) instead of behind it.  Events from a ring: the host runs a whole
// batch ahead.  On small systems the two cross-stream edges cost more than the overlap gains (xt.hip measured ~15 us per iteration).
#define XB_SIDE_RING 64
struct XbSide { hipStream_t st = nullptr; hipEvent_t a[XB_SIDE_RING], b[XB_SIDE_RING]; int device = -1; unsigned seq = 0; bool ready = false; };
static XbSide g_xb_side;
static int xtb_side_init()
{
    XbSide &S = g_xb_side; const int dev = eng().device;
    if (S.ready && S.device == dev) return 0;
    if (S.ready) { (void)hipStreamDestroy(S.st); for (int i = 0; i < XB_SIDE_RING; ++i) { (void)hipEventDestroy(S.a[i]); (void)hipEventDestroy(S.b[i]); } S.ready = false; }
    HIPCHK(hipStreamCreateWithFlags(&S.st, hipStreamNonBlocking));
    for (int i = 0; i < XB_SIDE_RING; ++i) { HIPCHK(hipEventCreateWithFlags(&S.a[i], hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&S.b[i], hipEventDisableTiming)); }
    S.device = dev; S.seq = 0; S.ready = true;
    return 0;
}

// test aid: make this rank fail once on the host side of block-CG iteration `iteration` of a sharded solve (dkmc_debug_inject_fault(3, it))
int g_xtb_fault_iter = -1;

// ---- host loop ------------------------------------------------------------------------------------------------------------------------
// Returns 0 with the scaled solution of column 0 in A.y; DKMC_XTB_BREAKDOWN (> 0, no error recorded) when an s x s system lost
// definiteness: A.y then holds the last good iterate and the caller continues with the single-vector loop from it.
// Auxiliary columns (dkmc_set_x_aux, dkmc_set_x_aux_warm).  Returns hs: columns 1 ... hs - 1 take the SMOOTH set (lowest Laplacian modes of the
// bounding box, always from a zero start: their systems converge before the physical one does, a warm start would leave them without a
// residual), columns hs ... s - 1 the HASH set -- started from the previous solve's solutions when the caller keeps them (warm).
//   mode 0: all hash.  1 / 3: all smooth.  2 (default): at tolerances of 1e-8 and looser the smooth set -- or, with the warm auxiliary start,
//   half smooth + half hash (measured on the oracle's X, tools/warm_aux_proto.py: 2.5 nm 33 -> 16-19 sweeps, against 40 with column 0 alone
//   warm-started); below 1e-8 all hash (smooth systems lose rank close to a converged tolerance).
static int xtb_aux_split(int mode, double tol2, bool warm, int s)
{
    if (mode == 0) return 1;
    if (mode == 1 || mode == 3) return 16;
    if (!(tol2 >= 1e-16)) return 1;
    return warm ? std::max(1, s / 2) : 16;
}
static int xtb_cg_body(const XtbArgs &A, int *iters_out, double *rr_out, bool *peer_used);
static int xtb_cg_slab(const XtbArgs &A, int nr, int me0, const XShare *emu_shares, int time_rank, int *iters_out, double *rr_out);
int xtb_cg(const XtbArgs &A, int *iters_out, double *rr_out)
{
    bool peer_used = false;
    // more than one rank: the state of the solve is distributed by row slabs (xtb_slab.inc; dkmc_set_x_slab(0) keeps the all-gather variant
    // below, which shards the tile stream only)
    if (A.sharded && eng().x_slab && comm_nranks() > 1 && comm_nranks() <= XS_MAXR && A.ay && A.az)
        return xtb_cg_slab(A, comm_nranks(), comm_rank(), nullptr, -1, iters_out, rr_out);
    double rr0 = 0.0;
    int rc = xtb_cg_body(A, iters_out, &rr0, &peer_used);
    // split polynomial preconditioner: the loop stops on the residual of L A L; when the TRUE residual of column 0 does not meet the stop test yet, the
    // solve is re-entered from the iterate it reached (the check and the code DKMC_XTB_AGAIN: end of xtb_cg_body).  A round that no longer reduces the
    // true residual by a factor of four has reached what the arithmetic gives (at tolerances of 1e-12 the recurrence residual the reference's test --
    // and the plain loop -- stop on goes below what the true one can reach): the solve ends there, as the plain loop's does.
    for (int round = 1; rc == DKMC_XTB_AGAIN && round < 6; ++round) {
        int it2 = 0; double rr1 = 0.0;
        rc = xtb_cg_body(A, &it2, &rr1, &peer_used);
        if (iters_out) *iters_out += it2;
        if (rc == DKMC_XTB_AGAIN && !(rr1 < 0.25 * rr0)) rc = 0;
        rr0 = rr1;
    }
    if (rc == DKMC_XTB_AGAIN) rc = 0;
    if (rr_out) *rr_out = rr0;
    // a sharded solve that failed with the peer-write exchange in use: the ranks' sequence counters may have drifted (comm.hip)
    if (rc != 0 && rc != DKMC_XTB_BREAKDOWN && peer_used) comm_peer_drop();
    return rc;
}
static int xtb_cg_body(const XtbArgs &A, int *iters_out, double *rr_out, bool *peer_used)
{
    Engine &e = eng(); hipStream_t st = e.stream;
    const int m = A.m, s = A.s, so = 4 * ((s + 3) / 4);                       // vector groups of four: the matrix instruction's width
    const size_t pan = (size_t)m * XB_SP;
    const long long ncell = (long long)A.nK * A.nW;
    const int ng = std::max(128, std::min(A.nK, 4096));                                     // row-kernel workgroups = partial Gram matrices (one row block each up to 1.3e5 S rows)
    double *panels = (double *)scratch(S_XTB_PANELS, (pan * 3 + m + 16) * 8);
    double *QS = (double *)scratch(S_XTB_QS, (size_t)A.ns_pad * XB_SP * 8);
    double *rowpartB = (double *)scratch(S_XTB_ROWPART, (size_t)(ncell + 1) * XT_R * so * 8);
    double *colpartB = (double *)scratch(S_XTB_COLPART, (size_t)(A.nrecords + 1) * XT_C * so * 8);
    double *gpart = (double *)scratch(S_XTB_GRAM, (size_t)ng * XB_NG * 256 * 8);
    double *small = (double *)scratch(S_XTB_SMALL, (size_t)(4 * 256 + 2 * XB_DSPLIT * XB_SP + XB_NG * 256) * 8);
    if (!panels || !QS || !rowpartB || !colpartB || !gpart || !small) return e.err_code;
    double *R = panels, *P = panels + pan, *T = panels + 2 * pan, *y0 = panels + 3 * pan;
    double *mats = small, *drvpart = small + 4 * 256, *gfin = drvpart + 2 * XB_DSPLIT * XB_SP;
    // split polynomial preconditioner (dkmc_set_x_poly; one GPU): see k_xtb_nmul.  Vp = L P, Zp = A Vp before the second L, W1 / W2 the Horner steps;
    // behind Vp: m zeros (the right-hand side of column 0 once the start vector has gone into it) and one double for the true residual
    const int pd = (!A.sharded && m > 2 && A.ns > 0) ? std::min(e.x_poly, XB_MAXPOLY) : 0;
    double *Vp = nullptr, *W1 = nullptr, *W2 = nullptr, *Zp = nullptr, *bz = nullptr;
    if (pd > 0) {
        Vp = (double *)scratch(S_XTB_PRE_V, (pan + m + 16) * 8); W1 = (double *)scratch(S_XTB_PRE_W1, pan * 8); W2 = (double *)scratch(S_XTB_PRE_W2, pan * 8);
        Zp = (double *)scratch(S_XTB_PRE_Z, pan * 8);
        if (!Vp || !W1 || !W2 || !Zp) return e.err_code;
        bz = Vp + pan;
        HIPCHK(hipMemsetAsync(bz, 0, (size_t)(m + 16) * 8, st));
    }
    // coefficients of L = p(N), p ~ (1 - x)^(-1/2): the Chebyshev interpolant of degree d on [-1, 1 - delta], delta = min(0.5, 1.6 / d^2), in the monomial
    // basis (Horner).  Against the Taylor series of the same degree -- which is exact at 0 and weakest where it matters, towards x -> 1 (the largest
    // eigenvalue of N is 0.99994 at 9.4 k sites) -- the block loop needs a third fewer sweeps (85 k sites, d = 4: 34 -> 24, 95 without preconditioner).
    double pc[XB_MAXPOLY + 1] = {1.0};
    if (pd > 0) {
        const int d = pd, n = d + 1;
        const double a = -1.0, b = 1.0 - std::min(0.5, 1.6 / (double)(d * d));
        double fx[XB_MAXPOLY + 1], c[XB_MAXPOLY + 1], pt[XB_MAXPOLY + 1] = {0}, Tm2[XB_MAXPOLY + 1] = {0}, Tm1[XB_MAXPOLY + 1] = {0};
        for (int k = 0; k < n; ++k) { const double t = cos(M_PI * (k + 0.5) / n), x = 0.5 * (b - a) * t + 0.5 * (b + a); fx[k] = 1.0 / sqrt(1.0 - x); }
        for (int j = 0; j < n; ++j) { double acc = 0.0; for (int k = 0; k < n; ++k) acc += fx[k] * cos(M_PI * j * (k + 0.5) / n); c[j] = acc * 2.0 / n; }
        c[0] *= 0.5;
        Tm2[0] = 1.0; Tm1[1] = 1.0;                                           // T_0, T_1 in powers of t
        pt[0] += c[0]; pt[1] += c[1];
        for (int j = 2; j <= d; ++j) {
            double Tj[XB_MAXPOLY + 1];
            for (int i = 0; i <= XB_MAXPOLY; ++i) Tj[i] = (i >= 1 ? 2.0 * Tm1[i - 1] : 0.0) - Tm2[i];
            for (int i = 0; i <= XB_MAXPOLY; ++i) { pt[i] += c[j] * Tj[i]; Tm2[i] = Tm1[i]; Tm1[i] = Tj[i]; }
        }
        const double al = 2.0 / (b - a), be = -(a + b) / (b - a);             // t = al x + be
        double res[XB_MAXPOLY + 2] = {0}; res[0] = pt[d]; int deg = 0;
        for (int i = d - 1; i >= 0; --i) {
            double nr[XB_MAXPOLY + 2] = {0};
            for (int q = 0; q <= deg; ++q) { nr[q] += res[q] * be; nr[q + 1] += res[q] * al; }
            ++deg; nr[0] += pt[i];
            for (int q = 0; q <= XB_MAXPOLY + 1; ++q) res[q] = nr[q];
        }
        for (int q = 0; q <= d; ++q) pc[q] = res[q];
    }
    const double tol2_loop = pd > 0 ? A.tol2 / 2.25 : A.tol2;                 // ||r|| <= 1.42 ||L r||: the loop stops a little early, the true residual is checked at the end
    HIPCHK(hipMemsetAsync(QS, 0, (size_t)A.ns_pad * XB_SP * 8, st));
    HIPCHK(hipMemsetAsync(rowpartB, 0, (size_t)(ncell + 1) * XT_R * so * 8, st));
    HIPCHK(hipMemsetAsync(colpartB, 0, (size_t)(A.nrecords + 1) * XT_C * so * 8, st));
    HIPCHK(hipMemsetAsync(A.ctrl, 0, sizeof(XCtrl), st));
    // sharded solve (comm.hip): this rank streams its share of the tiles; the tile sums of the S rows are completed by ONE all-gather of
    // ns x so doubles (+ two control words) per rank and sweep, added in rank order by every rank (k_xtb_rows): a sixteenth of the
    // exchanges of the single-vector loop, and bit-identical results on every rank by construction
    const bool sharded = A.sharded;
    const int nr = sharded ? comm_nranks() : 1, me = sharded ? comm_rank() : 0;
    double *xbuf = nullptr;
    const size_t xcount = (size_t)A.ns * so + 2;
    // exchange: the one-shot peer-write exchange when it is attached (comm.hip: push + signal + wait, no collective call), else one all-gather
    const bool peer = sharded && comm_peer_ready(xcount);
    *peer_used = peer;
    int xpar = 0;
    if (sharded) {
        if (!peer) { xbuf = (double *)scratch(S_CG_XCHG, xcount * nr * 8); if (!xbuf) return e.err_code; }
        hipLaunchKernelGGL(k_xtb_set_sharded, dim3(1), dim3(1), 0, st, A.ctrl);
    }
    // smooth auxiliary columns (see xtb_rhs): dkmc_set_x_aux 0 never, 1 always, 2 (default) at tolerances of 1e-8 and looser, 3 always with x modes only
    XbAux *aux = nullptr;
    // which auxiliary columns take which set, and which start from the previous solve's solutions (A.yaux, dkmc_set_x_aux_warm): see xtb_aux_split
    const int hs = xtb_aux_split(e.x_aux, A.tol2, A.yaux != nullptr, s);
    const bool keep_aux = A.yaux != nullptr && pd == 0;
    double *Ypanel = keep_aux ? (double *)scratch(S_XTB_YPANEL, pan * 8) : nullptr;
    if (keep_aux && !Ypanel) return e.err_code;
    if (A.ax && A.ay && A.az && m > 2 && hs > 1) {
        aux = (XbAux *)scratch(S_XTB_XI, sizeof(XbAux));
        if (!aux) return e.err_code;
        XbAux h0{};
        h0.hs = hs;
        for (int d = 0; d < 3; ++d) { h0.mm[2 * d] = ~0ull; h0.mm[2 * d + 1] = 0ull; }
        HIPCHK(hipMemcpyAsync(aux, &h0, sizeof(XbAux), hipMemcpyHostToDevice, st));       // (pageable source: copied before the call returns)
        hipLaunchKernelGGL(k_xtb_box, dim3(std::min((m - 2 + 255) / 256, 256)), dim3(256), 0, st, m - 2, A.ax, A.ay, A.az, aux);
        hipLaunchKernelGGL(k_xtb_modes, dim3(1), dim3(512), 0, st, aux, e.x_aux == 3 ? 1 : 0);
    }
    e.stats.xb_aux = aux ? (hs >= s ? 1 : 2) : 0;
    int local_fail = 0;
    hipLaunchKernelGGL(k_xtb_init, dim3((unsigned)((pan + 255) / 256)), dim3(256), 0, st, m, (const double *)A.y, A.sc, A.nsrank, y0, P, QS,
                       (const double *)((keep_aux && A.yaux_valid) ? A.yaux : nullptr), hs, s, Ypanel);
    const int ntb = (A.item_n + 3) / 4;
    const int nnb = 2 * XB_DSPLIT + (std::max(m - 2, 1) + 15) / 16;
    const int gs = xt_grid((m + 15) / 16, 4, 2048);
    const bool prof = e.profiling != 0;
    static hipEvent_t evs[4 * 8]; static bool evs_ready = false;
    if (prof && !evs_ready) { for (auto &ev : evs) HIPCHK(hipEventCreate(&ev)); evs_ready = true; }
    double prof_long_ms = 0.0, prof_short_ms = 0.0; int prof_long_n = 0, prof_short_n = 0;
    // neighbour part beside the tile kernel (see XbSide): only in a sharded solve, where a rank's tile pass is 1 / N of the sweep and the
    // (replicated) neighbour part would otherwise be a serial 0.3 ms behind it.  On ONE GPU the overlap was measured and does not pay: the two
    // kernels share the memory system, the tile pass slows by what the neighbour part takes (9.4e5 sites: 3.95 + 0 against 3.65 + 0.32 ms; 85 k sites,
    // where both are latency-bound: 15.0 against 13.5 ms per superstep -- the two event hand-overs per sweep cost more than the overlap gains).
    const bool side = sharded && m > 20000 && ntb > 0 && xtb_side_init() == 0;
    auto product = [&](hipEvent_t e0, hipEvent_t e1) {
        int sl = 0;
        if (side) {
            XbSide &S = g_xb_side; sl = (int)(S.seq++ % XB_SIDE_RING);
            (void)hipEventRecord(S.a[sl], st);                                 // P of the previous step (and the stop word) are final
            (void)hipStreamWaitEvent(S.st, S.a[sl], 0);
            hipLaunchKernelGGL(k_xtb_neigh, dim3(nnb), dim3(XT_NT), 0, S.st, m, A.rp, A.ci, A.val, (const double *)P, A.sc, A.nsrank, (const XCtrl *)A.ctrl, T, drvpart);
            (void)hipEventRecord(S.b[sl], S.st);
        }
        if (ntb > 0) {
#define XB_APPLY_ARGS_ A.item_n, A.items, A.tiles, A.sub_base, A.tval, (const double *)QS, A.nW, rowpartB, colpartB, (const XCtrl *)A.ctrl
#define XB_APPLY(NTL_, NG_) do { if (e.x_apply_form == 1) hipExtLaunchKernelGGL((k_xtb_apply<NTL_, NG_, 8>), dim3(ntb), dim3(XT_NT), 0, st, e0, e1, 0, XB_APPLY_ARGS_); \
                                 else hipExtLaunchKernelGGL((k_xtb_apply<NTL_, NG_>), dim3(ntb), dim3(XT_NT), 0, st, e0, e1, 0, XB_APPLY_ARGS_); } while (0)
            if (A.nt_loads) { if (so == 4) XB_APPLY(1, 1); else if (so == 8) XB_APPLY(1, 2); else if (so == 12) XB_APPLY(1, 3); else XB_APPLY(1, 4); }
            else { if (so == 4) XB_APPLY(0, 1); else if (so == 8) XB_APPLY(0, 2); else if (so == 12) XB_APPLY(0, 3); else XB_APPLY(0, 4); }
#undef XB_APPLY
#undef XB_APPLY_ARGS_
        }
        if (side) (void)hipStreamWaitEvent(st, g_xb_side.b[sl], 0);           // the sparse sums are in T before the row kernel reads them
        else hipLaunchKernelGGL(k_xtb_neigh, dim3(nnb), dim3(XT_NT), 0, st, m, A.rp, A.ci, A.val, (const double *)P, A.sc, A.nsrank, (const XCtrl *)A.ctrl, T, drvpart);
    };
    // dst = L src (Horner: d sparse panel products); dst must be none of src, W1, W2
    const int nmb = (m + 15) / 16;
    auto applyL = [&](const double *src, double *dst) {
        if (pd <= 0) return;
        const double *in = src;
        for (int i = 0; i < pd; ++i) {
            double *out = (i == pd - 1) ? dst : ((i & 1) ? W2 : W1);
            const int j = pd - 1 - i;                                         // out = c_j src + N (previous), the first step carries c_d
            hipLaunchKernelGGL(k_xtb_nmul, dim3(nmb), dim3(XT_NT), 0, st, m, A.rp, A.ci, A.val, A.sc, in, src, pc[j], i == 0 ? pc[pd] : 1.0, out, (const XCtrl *)A.ctrl);
            in = out;
        }
    };
    auto fold_rows = [&](double *Tt) {
        hipLaunchKernelGGL(k_xtb_fold_rows, dim3(std::max(ng, 1)), dim3(XT_NT), 0, st, A.ns, A.nK, A.nW, so, A.wrange, A.nitem_w, (const double *)rowpartB, (const double *)colpartB,
                           A.srow, A.sS, A.sc, (const double *)drvpart, Tt, (const XCtrl *)A.ctrl);
    };
    // T = L A L P
    auto product_pre = [&](hipEvent_t e0, hipEvent_t e1) {
        applyL((const double *)P, Vp);
        hipLaunchKernelGGL(k_xtb_qs_from, dim3((unsigned)((pan + 255) / 256)), dim3(256), 0, st, m, (const double *)Vp, A.sc, A.nsrank, QS, (const XCtrl *)A.ctrl);
        double *Pk = P, *Tk = T; P = Vp; T = Zp;                              // (the product reads P and QS, writes T)
        product(e0, e1);
        P = Pk; T = Tk;
        fold_rows(Zp);
        applyL((const double *)Zp, T);
    };
    const double *bsel = A.b;
#define XB_ROWS_ARGS(IT_) A.ns, A.nK, A.nW, m, s, so, A.wrange, A.nitem_w, (const double *)rowpartB, (const double *)colpartB, A.srow, A.sS, A.nsrank, A.sc, \
                     (const double *)drvpart, T, (const double *)P, R, bsel, gpart, A.ctrl, (const double *)xbuf, IT_, nr, (const XbAux *)aux, A.ax, A.ay, A.az
    // S rows of T + partial Gram matrices; a sharded solve exchanges the tile sums first.  A host-side failure of this rank between two
    // collectives must not leave the peers in the all-reduce: it still joins, with the abort word set, and every rank leaves together.
    auto rows = [&](bool init, int itn, hipEvent_t e2, hipEvent_t e3) -> int {
        if (sharded) {
            if (peer) { xbuf = comm_peer_slots(xpar); }
            hipLaunchKernelGGL(k_xtb_fold_local, dim3(std::max(A.nK, 1)), dim3(XT_NT), 0, st, A.ns, A.nK, A.nW, so, A.wrange, A.nitem_w, (const double *)rowpartB,
                               (const double *)colpartB, xbuf + me * xcount, (const XCtrl *)A.ctrl, 1, A.w_lo, A.w_hi);
            if (hipGetLastError() != hipSuccess || local_fail) {
                if (!local_fail) local_fail = dkmc_fail(92, "block-CG: launch failed between two collectives", __FILE__, __LINE__);
                hipLaunchKernelGGL(k_xtb_abort_word, dim3(1), dim3(1), 0, st, A.ctrl, xbuf + me * xcount, (size_t)A.ns * so + 1);
            }
            if (peer) {
                if (int rcx = comm_peer_exchange(xpar, xcount, &A.ctrl->done, &A.ctrl->aborted, &A.ctrl->xchg_timeout, itn + 2)) return rcx;
                xpar ^= 1;
            } else if (int rcx = comm_allgather_f64(xbuf, xcount)) return rcx;
            if (init) hipLaunchKernelGGL((k_xtb_rows<1, 1>), dim3(ng), dim3(XT_NT), 0, st, XB_ROWS_ARGS(itn));
            else hipExtLaunchKernelGGL((k_xtb_rows<0, 1>), dim3(ng), dim3(XT_NT), 0, st, e2, e3, 0, XB_ROWS_ARGS(itn));
        } else if (pd > 0) {
            if (init) hipLaunchKernelGGL((k_xtb_rows<1, 0, 1>), dim3(ng), dim3(XT_NT), 0, st, XB_ROWS_ARGS(itn));
            else hipExtLaunchKernelGGL((k_xtb_rows<0, 0, 1>), dim3(ng), dim3(XT_NT), 0, st, e2, e3, 0, XB_ROWS_ARGS(itn));
        } else {
            if (init) hipLaunchKernelGGL((k_xtb_rows<1, 0>), dim3(ng), dim3(XT_NT), 0, st, XB_ROWS_ARGS(itn));
            else hipExtLaunchKernelGGL((k_xtb_rows<0, 0>), dim3(ng), dim3(XT_NT), 0, st, e2, e3, 0, XB_ROWS_ARGS(itn));
        }
        return 0;
    };
    // ---- R = A Y0 - B ; first directions ----
    product(nullptr, nullptr);
    if (pd > 0) {
        // the start vector goes into the right-hand side: column 0 solves L A L dh = L (b - A y0) from zero, the auxiliary columns keep their own
        fold_rows(T);                                                         // T = A Y0
        hipLaunchKernelGGL(k_xtb_pre_resid, dim3((unsigned)((pan + 255) / 256)), dim3(256), 0, st, m, (const double *)T, A.b, Zp);
        applyL((const double *)Zp, T);                                        // T(:, 0) = L (A y0 - b), the other columns 0
        bsel = bz;                                                            // R = T - [0 | auxiliary right-hand sides]
        HIPCHK(hipMemsetAsync(y0, 0, (size_t)m * 8, st));
    }
    if (int rcx = rows(true, -1, nullptr, nullptr)) return rcx;
    hipLaunchKernelGGL(k_xtb_gred, dim3(XB_NG * 16), dim3(XT_NT), 0, st, ng, (const double *)gpart, gfin, (const XCtrl *)A.ctrl);
    hipLaunchKernelGGL(k_xtb_small, dim3(1), dim3(256), 0, st, -1, s, (const double *)gfin, mats, A.ctrl, tol2_loop);
    hipLaunchKernelGGL(k_xtb_zero, dim3((unsigned)((pan + 255) / 256)), dim3(256), 0, st, (long long)pan, P);          // Y0 has served: P_{-1} = 0
    hipLaunchKernelGGL(k_xtb_step, dim3(gs), dim3(XT_NT), 0, st, m, -1, (const double *)mats, y0, R, P, (const double *)T, A.sc, A.nsrank, QS, (const XCtrl *)A.ctrl,
                       (const int *)nullptr, Ypanel);
    KCHK();
    int it = 0, launched = 0, batch = 4;
    // launch plan: the first batch covers three quarters of the previous solve's sweeps -- with the warm start the count moves by +-30 per cent from
    // step to step (9.4e5 sites: 268 ... 505), and a batch sized to the previous count left up to a quarter of its launches as no-ops --, then batches of 8
    if (e.x_iter_hint > 12) batch = std::max(4, e.x_iter_hint * 3 / 4);
    XCtrl h{};
    for (;;) {
        HIPCHK(hipMemcpyAsync(&h, A.ctrl, sizeof(XCtrl), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (prof && launched && ntb > 0) {
            for (int bq = 0; bq < 8 && bq * XT_PROF_STRIDE < launched; ++bq) {
                if (it - launched + bq * XT_PROF_STRIDE >= h.iters) break;
                float ms = 0.f;
                HIPCHK(hipEventElapsedTime(&ms, evs[4 * bq], evs[4 * bq + 1])); prof_long_ms += ms; ++prof_long_n;
                HIPCHK(hipEventElapsedTime(&ms, evs[4 * bq + 2], evs[4 * bq + 3])); prof_short_ms += ms; ++prof_short_n;
            }
        }
        if (h.done) break;
        if (it >= 200000) { dkmc_fail(4, "block-CG: no convergence after 200000 iterations", __FILE__, __LINE__); break; }
        for (int bq = 0; bq < batch; ++bq, ++it) {
            const bool pb = prof && ntb > 0 && bq < 8 * XT_PROF_STRIDE && (bq % XT_PROF_STRIDE == 0);      // (no tile launch: its events would never be recorded)
            const int sl = bq / XT_PROF_STRIDE;
            if (g_xtb_fault_iter >= 0 && sharded && it >= g_xtb_fault_iter) { g_xtb_fault_iter = -1; local_fail = dkmc_fail(91, "injected fault (block-CG iteration)", __FILE__, __LINE__); }
            if (pd > 0) product_pre(pb ? evs[4 * sl] : nullptr, pb ? evs[4 * sl + 1] : nullptr);
            else product(pb ? evs[4 * sl] : nullptr, pb ? evs[4 * sl + 1] : nullptr);
            if (int rcx = rows(false, it, pb ? evs[4 * sl + 2] : nullptr, pb ? evs[4 * sl + 3] : nullptr)) return rcx;
            hipLaunchKernelGGL(k_xtb_gred, dim3(XB_NG * 16), dim3(XT_NT), 0, st, ng, (const double *)gpart, gfin, (const XCtrl *)A.ctrl);
            hipLaunchKernelGGL(k_xtb_small, dim3(1), dim3(256), 0, st, it, s, (const double *)gfin, mats, A.ctrl, tol2_loop);
            hipLaunchKernelGGL(k_xtb_step, dim3(gs), dim3(XT_NT), 0, st, m, it, (const double *)mats, y0, R, P, (const double *)T, A.sc, A.nsrank, QS, (const XCtrl *)A.ctrl,
                               (const int *)nullptr, Ypanel);
        }
        launched = batch;
        KCHK();
        if (e.x_iter_hint > 12) batch = 8; else if (batch < 64) batch *= 2;
    }
#undef XB_ROWS_ARGS
    if (local_fail) return local_fail;                                         // the peers were told (abort word)
    if (h.xchg_timeout) return dkmc_fail(48, "block-CG: the peer-write exchange timed out waiting for a peer's slot", __FILE__, __LINE__);
    if (h.aborted) return dkmc_fail(46, "a peer rank aborted the sharded current solve", __FILE__, __LINE__);
    if (e.err_code) return e.err_code;
    bool again = false;
    if (pd > 0) {
        // y = y0 + L dh, then the TRUE residual of column 0 in the unpreconditioned system: one more pass over the tiles
        HIPCHK(hipMemsetAsync(A.ctrl, 0, sizeof(XCtrl), st));                 // (the kernels below are gated by `done`)
        hipLaunchKernelGGL(k_xtb_pre_col0, dim3((unsigned)((pan + 255) / 256)), dim3(256), 0, st, m, (const double *)y0, Zp);
        applyL((const double *)Zp, Vp);
        hipLaunchKernelGGL(k_xtb_pre_add, dim3((m + 255) / 256), dim3(256), 0, st, m, (const double *)Vp, A.y);
        if (!h.pad[0]) {
            hipLaunchKernelGGL(k_xtb_pre_col0, dim3((unsigned)((pan + 255) / 256)), dim3(256), 0, st, m, (const double *)A.y, P);
            hipLaunchKernelGGL(k_xtb_qs_from, dim3((unsigned)((pan + 255) / 256)), dim3(256), 0, st, m, (const double *)P, A.sc, A.nsrank, QS, (const XCtrl *)A.ctrl);
            product(nullptr, nullptr);
            fold_rows(T);
            hipLaunchKernelGGL(k_xtb_pre_rr, dim3(1), dim3(1024), 0, st, m, (const double *)T, A.b, bz + m);
            double rr_true = 0.0;
            HIPCHK(hipMemcpyAsync(&rr_true, bz + m, 8, hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            h.rr[h.iters & 1] = rr_true;
            again = rr_true > A.tol2;
        }
    } else
    HIPCHK(hipMemcpyAsync(A.y, y0, (size_t)m * 8, hipMemcpyDeviceToDevice, st));
    if (keep_aux) hipLaunchKernelGGL(k_xtb_yaux_out, dim3((unsigned)((pan + 255) / 256)), dim3(256), 0, st, m, (const int *)nullptr, (const double *)Ypanel, A.sc, A.yaux);
    e.x_iter_hint = h.iters;
    if (iters_out) *iters_out = h.iters;
    if (rr_out) *rr_out = h.rr[h.iters & 1];
    if (prof) {
        e.stats.spmv_long_ms = prof_long_ms; e.stats.spmv_short_ms = prof_short_ms;
        e.stats.spmv_long_launches = prof_long_n; e.stats.spmv_short_launches = prof_short_n;
    }
    if (h.pad[0]) return DKMC_XTB_BREAKDOWN;
    return again ? DKMC_XTB_AGAIN : 0;
}

#include "xtb_slab.inc"
// emulation entry (xt.hip: dkmc_xtb_emulate_slabs): nr virtual ranks in this process on the resident X of a single-GPU solve
int xtb_cg_slab_emulate(const XtbArgs &A, int nr, const XShare *shares, int time_rank, int sweep_cap, int *iters_out, double *rr_out, double *times_us, long long *xdoubles)
{
    g_slab_sweep_cap = sweep_cap;
    const int rc = xtb_cg_slab(A, nr, 0, shares, time_rank, iters_out, rr_out);
    g_slab_sweep_cap = 0;
    if (times_us) for (int c = 0; c < 8; ++c) times_us[c] = g_slab_times.us[c];
    if (xdoubles) for (int c = 0; c < 3; ++c) xdoubles[c] = g_slab_xbytes[c];
    return rc;
}

// ---- test aid (tests/test_gpu_block_cg.py; no counterpart in the reference) ---------------------------------------------------------------
// On the X left resident by the last single-GPU solve: the tile x panel product of 16 test vectors (k_xtb_apply + the fold of k_xtb_rows)
// against 16 passes of the single-vector tile kernel (xt_tile_sums_mv).  Reports the largest deviation and the largest sum.
__global__ void k_xtb_test_panel(int ns, double *__restrict__ QS)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ns * XB_SP) return;
    const int r = i >> 4, v = i & 15;
    QS[xtb_qs_pos(r, v)] = 0.25 + (double)((((unsigned)r * 2654435761u) ^ ((unsigned)v * 40503u)) >> 20) / 4096.0 + 0.125 * v;
}
__global__ void k_xtb_test_column(int ns, int v, const double *__restrict__ QS, double *__restrict__ vS)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < ns) vS[r] = QS[xtb_qs_pos(r, v)];
}
__global__ void k_xtb_test_compare(int ns, int v, const int *__restrict__ srow, const double *__restrict__ T, const double *__restrict__ ref, double *__restrict__ out)
{
    // out[0] = max |T - ref|, out[1] = max |ref| (atomicMax on the bits of non-negative doubles)
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= ns) return;
    const double a = T[(size_t)srow[r] * XB_SP + v], b = ref[r];
    atomicMax(reinterpret_cast<unsigned long long *>(out), (unsigned long long)__double_as_longlong(fabs(a - b)));
    atomicMax(reinterpret_cast<unsigned long long *>(out + 1), (unsigned long long)__double_as_longlong(fabs(b)));
}
int xt_tile_sums_mv(const double *vS, double *out);
extern "C" int dkmc_xtb_check_product(int width, double *max_abs_diff, double *max_abs)
{
    Engine &e = eng(); hipStream_t st = e.stream; const XTState &X = g_xt;
    if (!X.valid || comm_attached() || X.tile_n != X.ntiles || X.ns <= 0) return dkmc_fail(13, "xtb_check_product: needs the X of a single-GPU solve", __FILE__, __LINE__);
    const int m = X.Nsub, ns = X.ns, s = std::max(2, std::min(width, 16)), so = 4 * ((s + 3) / 4);
    const size_t pan = (size_t)m * XB_SP;
    const long long ncell = (long long)X.nK * X.nW;
    const int nrec = X.nitems >> X.rec_shift, ng = 64;
    double *panels = (double *)scratch(S_XTB_PANELS, (pan * 3 + m + 16) * 8);
    double *QS = (double *)scratch(S_XTB_QS, (size_t)X.ns_pad * XB_SP * 8);
    double *rowpartB = (double *)scratch(S_XTB_ROWPART, (size_t)(ncell + 1) * XT_R * so * 8);
    double *colpartB = (double *)scratch(S_XTB_COLPART, (size_t)(nrec + 1) * XT_C * so * 8);
    double *gpart = (double *)scratch(S_XTB_GRAM, (size_t)4096 * XB_NG * 256 * 8);
    double *small = (double *)scratch(S_XTB_SMALL, (size_t)(4 * 256 + 2 * XB_DSPLIT * XB_SP + XB_NG * 256) * 8);
    double *tmp = (double *)scratch(S_XT_T_MISC, (size_t)4 * X.ns_pad * 8);
    XCtrl *ctrl = (XCtrl *)scratch(S_MISC2, 256);
    double *sS = (double *)e.buf[S_CG_PS], *sc = (double *)e.buf[S_CG_S], *rhs = (double *)e.buf[S_X_RHS];
    if (!panels || !QS || !rowpartB || !colpartB || !gpart || !small || !tmp || !ctrl || !sS || !sc || !rhs) return e.err_code ? e.err_code : dkmc_fail(13, "xtb_check_product: no solver state", __FILE__, __LINE__);
    sS += X.ns_pad;
    double *T = panels + 2 * pan, *R = panels, *P = panels + pan, *drvpart = small + 4 * 256;
    double *vS = tmp, *ref = tmp + X.ns_pad, *res = tmp + 2 * (size_t)X.ns_pad;
    HIPCHK(hipMemsetAsync(tmp, 0, (size_t)4 * X.ns_pad * 8, st));
    HIPCHK(hipMemsetAsync(QS, 0, (size_t)X.ns_pad * XB_SP * 8, st));
    HIPCHK(hipMemsetAsync(panels, 0, (pan * 3 + m + 16) * 8, st));
    HIPCHK(hipMemsetAsync(small, 0, (size_t)(4 * 256 + 2 * XB_DSPLIT * XB_SP) * 8, st));
    HIPCHK(hipMemsetAsync(rowpartB, 0, (size_t)(ncell + 1) * XT_R * so * 8, st));
    HIPCHK(hipMemsetAsync(colpartB, 0, (size_t)(nrec + 1) * XT_C * so * 8, st));
    HIPCHK(hipMemsetAsync(ctrl, 0, sizeof(XCtrl), st));
    hipLaunchKernelGGL(k_xtb_test_panel, dim3((ns * XB_SP + 255) / 256), dim3(256), 0, st, ns, QS);
#define XB_APPLY_ARGS_ X.item_n, (const XItem *)g_xb.items + X.item_lo, (const XTile *)g_xb.tiles, (int)X.sub_base, (const double *)g_xb.tval, (const double *)QS, X.nW, rowpartB, colpartB, (const XCtrl *)ctrl
#define XB_APPLY(NG_) do { if (e.x_apply_form == 1) hipLaunchKernelGGL((k_xtb_apply<1, NG_, 8>), dim3((X.item_n + 3) / 4), dim3(XT_NT), 0, st, XB_APPLY_ARGS_); \
                           else hipLaunchKernelGGL((k_xtb_apply<1, NG_>), dim3((X.item_n + 3) / 4), dim3(XT_NT), 0, st, XB_APPLY_ARGS_); } while (0)
    if (so == 4) XB_APPLY(1); else if (so == 8) XB_APPLY(2); else if (so == 12) XB_APPLY(3); else XB_APPLY(4);
#undef XB_APPLY
    hipLaunchKernelGGL((k_xtb_rows<1, 0>), dim3(ng), dim3(XT_NT), 0, st, ns, X.nK, X.nW, m, s, so, (const int2 *)g_xb.wrange, (const int *)g_xb.nitem_w, (const double *)rowpartB,
                       (const double *)colpartB, (const int *)g_xb.srow, (const double *)sS, (const int *)g_xb.nsrank, (const double *)sc, (const double *)drvpart, T,
                       (const double *)P, R, (const double *)rhs, gpart, ctrl, (const double *)nullptr, -1, 1, (const XbAux *)nullptr, (const double *)nullptr, (const double *)nullptr, (const double *)nullptr);
    KCHK();
    const int gb = (ns + 255) / 256;
    for (int v = 0; v < so; ++v) {
        hipLaunchKernelGGL(k_xtb_test_column, dim3(gb), dim3(256), 0, st, ns, v, (const double *)QS, vS);
        int rc = xt_tile_sums_mv(vS, ref); if (rc) return rc;
        // T holds sS * (tile sums): compare with sS * ref
        hipLaunchKernelGGL(k_xt_vec_mul, dim3(gb), dim3(256), 0, st, ns, ref, (const double *)sS);
        hipLaunchKernelGGL(k_xtb_test_compare, dim3(gb), dim3(256), 0, st, ns, v, (const int *)g_xb.srow, (const double *)T, (const double *)ref, res);
    }
    double h[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(h, res, 16, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (max_abs_diff) *max_abs_diff = h[0];
    if (max_abs) *max_abs = h[1];
    return e.err_code;
}

// ---- measurement aid (bench.py / tools; no counterpart in the reference) ------------------------------------------------------------------
// Average duration of the tile x panel kernel over the X left resident by the last single-GPU solve, `reps` launches back to back.
// variant 0: the kernel as a solve runs it; 1: without its matrix instructions; 2: without re-reading the tile stream (see k_xtb_apply).
extern "C" int dkmc_xtb_time_apply(int width, int variant, int reps, double *us)
{
    Engine &e = eng(); hipStream_t st = e.stream; const XTState &X = g_xt;
    if (!X.valid || comm_attached() || X.tile_n != X.ntiles || X.ns <= 0 || reps < 1) return dkmc_fail(13, "xtb_time_apply: needs the X of a single-GPU solve", __FILE__, __LINE__);
#ifndef DKMC_MEASURE_VARIANTS
    if (variant != 0) return dkmc_fail(13, "xtb_time_apply: this build carries no measurement variants (DKMC_MEASURE_VARIANTS=1 python __graft_entry__.py)", __FILE__, __LINE__);
#endif
    const int s = std::max(2, std::min(width, 16)), so = 4 * ((s + 3) / 4);
    const long long ncell = (long long)X.nK * X.nW;
    const int nrec = X.nitems >> X.rec_shift;
    double *QS = (double *)scratch(S_XTB_QS, (size_t)X.ns_pad * XB_SP * 8);
    double *rowpartB = (double *)scratch(S_XTB_ROWPART, (size_t)(ncell + 1) * XT_R * so * 8);
    double *colpartB = (double *)scratch(S_XTB_COLPART, (size_t)(nrec + 1) * XT_C * so * 8);
    XCtrl *ctrl = (XCtrl *)scratch(S_MISC2, 256);
    if (!QS || !rowpartB || !colpartB || !ctrl) return e.err_code;
    HIPCHK(hipMemsetAsync(ctrl, 0, sizeof(XCtrl), st));
    hipLaunchKernelGGL(k_xtb_test_panel, dim3((X.ns * XB_SP + 255) / 256), dim3(256), 0, st, X.ns, QS);
    hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    for (int r = -1; r < reps; ++r) {
        if (r == 0) HIPCHK(hipEventRecord(e0, st));
#define XB_APPLY(NG_, V_) do { if (e.x_apply_form == 1 && (V_) == 0) hipLaunchKernelGGL((k_xtb_apply<1, NG_, 8>), dim3((X.item_n + 3) / 4), dim3(XT_NT), 0, st, XB_APPLY_ARGS_); \
                               else hipLaunchKernelGGL((k_xtb_apply<1, NG_, V_>), dim3((X.item_n + 3) / 4), dim3(XT_NT), 0, st, XB_APPLY_ARGS_); } while (0)
#ifdef DKMC_MEASURE_VARIANTS
        if (so == 4) XB_APPLY(1, 0);
        else if (so == 8) { if (variant == 1) XB_APPLY(2, 1); else if (variant == 2) XB_APPLY(2, 2); else XB_APPLY(2, 0); }
        else if (so == 12) XB_APPLY(3, 0);
        else { if (variant == 1) XB_APPLY(4, 1); else if (variant == 2) XB_APPLY(4, 2); else if (variant == 3) XB_APPLY(4, 3); else if (variant == 4) XB_APPLY(4, 4); else if (variant == 7) XB_APPLY(4, 7); else if (variant == 10) XB_APPLY(4, 10); else if (variant == 12) XB_APPLY(4, 12); else XB_APPLY(4, 0); }
#else
        // (the library as shipped carries the product kernel and the round-4 form only; DKMC_MEASURE_VARIANTS=1 python __graft_entry__.py builds the rest)
        if (so == 4) XB_APPLY(1, 0); else if (so == 8) XB_APPLY(2, 0); else if (so == 12) XB_APPLY(3, 0); else XB_APPLY(4, 0);
#endif
#undef XB_APPLY
    }
    HIPCHK(hipEventRecord(e1, st));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0.f; HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    KCHK();
    if (us) *us = (double)ms * 1e3 / reps;
    return e.err_code;
}
