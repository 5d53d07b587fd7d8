// kcg.hip -- Jacobi-scaled CG for the potential matrix K (background potential, CB edge).
//
// Replaces Assemble_A / Assemble_A_CB (potential_solver_gpu.cu:397-593) + solve_sparse_CG_Jacobi (iterative_solvers_gpu.cu:309-480)
// for K.  Same algorithm, sign convention and stop tests as cg.hip (r = A y - b, p = -r, alpha = r.r / p.Ap, first test on ||r||,
// later ones on ||r||^2, both against tol^2).  What is different from the general CSR solver of cg.hip:
//   * K's off-diagonal entries take only two values, -high_G and -low_G (potential_solver_gpu.cu:202-249), decided by the classes of
//     the two sites.  So the matrix is never written out: per stored entry one int (column | class bit 31), per row the diagonal.
//     4 B per non-zero in the iteration instead of 12 (value + column), and no per-step pass that forms S K S: the Jacobi scaling
//     is applied to the vectors, t = S K (S p).
//   * 8 lanes per row (26 entries on average), every load of a row issued before the first use: three dependent memory latencies
//     per row (row pointers -> columns -> q) instead of one per 16 entries.
// Bytes per iteration: 4 nnz + 4 (m + 1) + 15 vector touches of 8 m  (CSR formulation, SURVEY 8d: 12 nnz + 4 (m + 1) + 96 m).
#include "common.h"

#define KC_NT 256
struct KCtrl { double rr[2]; double pad; int done; int iters; };

// ---- assembly: class bits + diagonal + rhs in one pass (16 lanes per row) ------------------------------------------------
// CB: 0 potential rule, 1 CB-edge rule, 2 CB-edge rule on atoms only (dkmc_set_cb_edge_domain(1): links to interstitial sites, DEFECT or
// OXYGEN_DEFECT, do not exist -- the domain the revision behind the reference's own CSR dump and current log solved the CB edge on)
__device__ __forceinline__ bool k_interstitial(int e) { return e == DEFECT || e == OXYGEN_DEFECT; }
template <int CB>
__device__ __forceinline__ bool k_high(int ei, int ej, int qi, int qj, const MetalSet &ms)
{
    const bool m1 = is_metal(ei, ms), m2 = is_metal(ej, ms);
    if (CB) return m1 || m2;                                            // potential_solver_gpu.cu:239-249
    const bool cv1 = (ei == VACANCY) && (qi == 0), cv2 = (ej == VACANCY) && (qj == 0);
    return (m1 && m2) || (cv1 && cv2);                                  // :202-217
}

template <int CB>
__global__ __launch_bounds__(KC_NT) void k_kc_assemble(int m, int N_left, const int *__restrict__ element, const int *__restrict__ charge,
                                                       MetalSet ms, double high_G, double low_G,
                                                       const int *__restrict__ rp, const int *__restrict__ ci,
                                                       const int *__restrict__ lrp, const int *__restrict__ lci,
                                                       const int *__restrict__ rrp, const int *__restrict__ rci,
                                                       double VL, double VR, int *__restrict__ cf, double *__restrict__ diag, double *__restrict__ rhs)
{
    const int LPR = 16;
    const int g = threadIdx.x / LPR, l = threadIdx.x % LPR;
    const int r = blockIdx.x * (KC_NT / LPR) + g;
    if (r >= m) return;
    const int i = N_left + r;
    const int ei = element[i], qi = charge[i];
    double off = 0.0, kl = 0.0, kr = 0.0;
    for (int p = rp[r] + l; p < rp[r + 1]; p += LPR) {
        const int c = ci[p];
        if (c == r) { cf[p] = c; continue; }
        const int j = N_left + c;
        const int ej = element[j];
        if (CB == 2 && (k_interstitial(ei) || k_interstitial(ej))) { cf[p] = r; continue; }      // no link: stored as the row's own column, which k_kc_apply skips
        const bool hi = k_high<CB>(ei, ej, qi, charge[j], ms);
        cf[p] = hi ? (c | (int)0x80000000) : c;
        off += hi ? high_G : low_G;
    }
    const bool cut = CB == 2 && k_interstitial(ei);
    for (int p = lrp[r] + l; p < lrp[r + 1] && !cut; p += LPR) { const int j = lci[p]; const int ej = element[j]; if (CB == 2 && k_interstitial(ej)) continue; kl += k_high<CB>(ei, ej, qi, charge[j], ms) ? high_G : low_G; }
    for (int p = rrp[r] + l; p < rrp[r + 1] && !cut; p += LPR) { const int j = N_left + m + rci[p]; const int ej = element[j]; if (CB == 2 && k_interstitial(ej)) continue; kr += k_high<CB>(ei, ej, qi, charge[j], ms) ? high_G : low_G; }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) { off += __shfl_xor(off, o, LPR); kl += __shfl_xor(kl, o, LPR); kr += __shfl_xor(kr, o, LPR); }
    if (l == 0) {
        double d = off;          // reduce_rows_into_diag: -(sum of off-diagonals)
        d += kl;                 // add_vector_to_diagonal (left)
        d += kr;                 // add_vector_to_diagonal (right)
        if (CB == 2 && d == 0.0) d = 1.0;      // an unlinked interstitial site: identity row, value 0
        diag[r] = d;
        rhs[r] = kl * VL + kr * VR;
    }
}

// s = 1/sqrt(diag); b *= s; y /= s; q = s y
__global__ void k_kc_scale(int m, const double *__restrict__ diag, double *__restrict__ s, double *__restrict__ b, double *__restrict__ y, double *__restrict__ q)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const double sv = 1.0 / sqrt(diag[i]);
    s[i] = sv;
    b[i] = b[i] * sv;
    const double ys = y[i] * 1 / sv;
    y[i] = ys;
    q[i] = sv * ys;
}

// t = S K q (q = S p): 8 lanes per row.  MODE 0: + partial p.t (iteration).  MODE 1: r = t - b, p = -r, partial r.r (start).
template <int MODE>
__global__ __launch_bounds__(KC_NT) void k_kc_apply(int m, const int *__restrict__ rp, const int *__restrict__ cf, const double *__restrict__ diag,
                                                    const double *__restrict__ s, const double *__restrict__ q, double high_G, double low_G,
                                                    const double *__restrict__ pv, double *__restrict__ t, double *__restrict__ part, const KCtrl *ctrl,
                                                    const double *__restrict__ b, double *__restrict__ r, double *__restrict__ p)
{
    __shared__ double red[KC_NT / 64];
    __shared__ int sdone;
    if (MODE == 0) {
        if (threadIdx.x == 0) sdone = ctrl->done;
        __syncthreads();
        if (sdone) return;
    }
    const int g = threadIdx.x >> 3, l = threadIdx.x & 7;
    double acc = 0.0;
    for (int row = blockIdx.x * (KC_NT / 8) + g; row < m; row += gridDim.x * (KC_NT / 8)) {
        const int p0 = rp[row], p1 = rp[row + 1];
        const double dq = diag[row] * q[row], sv = s[row];
        double sum = 0.0;
        for (int pb = p0 + l; pb < p1; pb += 32) {
            int c[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int pp = pb + 8 * u; c[u] = pp < p1 ? cf[pp] : row; }      // out of range / diagonal: skipped below
            double x[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) x[u] = q[c[u] & 0x7fffffff];
#pragma unroll
            for (int u = 0; u < 4; ++u) sum += (c[u] & 0x7fffffff) == row ? 0.0 : (c[u] < 0 ? high_G : low_G) * x[u];
        }
        sum += __shfl_xor(sum, 4, 8); sum += __shfl_xor(sum, 2, 8); sum += __shfl_xor(sum, 1, 8);
        if (l == 0) {
            const double tv = sv * (dq - sum);
            if (MODE == 0) { t[row] = tv; acc += pv[row] * tv; }
            else { const double rv = -b[row] + tv; r[row] = rv; p[row] = -rv; acc += rv * rv; }      // (q = s p: k_kc_q, after every row has read the old q)
        }
    }
    const double tot = block_sum_all<KC_NT>(acc, red);
    if (threadIdx.x == 0) part[blockIdx.x] = tot;
}

__device__ __forceinline__ double kc_reduce(const double *part, int n, double *red, const KCtrl *ctrl, bool *done)
{
    __shared__ int sdone;
    if (threadIdx.x == 0) sdone = ctrl->done;
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += KC_NT) s += part[i];
    s = block_sum_all<KC_NT>(s, red);
    *done = sdone != 0;
    return s;
}
__global__ __launch_bounds__(KC_NT) void k_kc_check0(const double *part, int npart, KCtrl *ctrl, double tol2)
{
    __shared__ double red[KC_NT / 64];
    double s = 0.0;
    for (int i = threadIdx.x; i < npart; i += KC_NT) s += part[i];
    const double rr = block_sum_all<KC_NT>(s, red);
    if (threadIdx.x == 0) { ctrl->rr[0] = rr; ctrl->rr[1] = rr; ctrl->iters = 0; ctrl->done = !(sqrt(rr) > tol2); }
}
// alpha = rr / p.t ; y += alpha p ; r += alpha t ; partial r.r
__global__ __launch_bounds__(KC_NT) void k_kc_update(int m, int it, const double *__restrict__ part_pt, int npart, const double *__restrict__ p,
                                                     const double *__restrict__ t, double *__restrict__ y, double *__restrict__ r,
                                                     double *__restrict__ part_rr, const KCtrl *ctrl)
{
    __shared__ double red[KC_NT / 64];
    bool done;
    const double pAp = kc_reduce(part_pt, npart, red, ctrl, &done);
    if (done) return;
    const double alpha = ctrl->rr[it & 1] / pAp;
    double acc = 0.0;
    for (int i = blockIdx.x * KC_NT + threadIdx.x; i < m; i += gridDim.x * KC_NT) {
        y[i] += alpha * p[i];
        const double rn = r[i] + alpha * t[i];
        r[i] = rn;
        acc += rn * rn;
    }
    const double tot = block_sum_all<KC_NT>(acc, red);
    if (threadIdx.x == 0) part_rr[blockIdx.x] = tot;
}
// beta = rr' / rr ; p = beta p - r ; q = s p ; stop test on rr'
__global__ __launch_bounds__(KC_NT) void k_kc_direction(int m, int it, const double *__restrict__ part_rr, int npart, const double *__restrict__ r,
                                                        double *__restrict__ p, const double *__restrict__ s, double *__restrict__ q, KCtrl *ctrl, double tol2)
{
    __shared__ double red[KC_NT / 64];
    bool done;
    const double rr_new = kc_reduce(part_rr, npart, red, ctrl, &done);
    if (done) return;
    const double beta = rr_new / ctrl->rr[it & 1];
    for (int i = blockIdx.x * KC_NT + threadIdx.x; i < m; i += gridDim.x * KC_NT) {
        const double pn = p[i] * beta - r[i];
        p[i] = pn;
        q[i] = s[i] * pn;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        ctrl->rr[(it + 1) & 1] = rr_new;
        ctrl->iters = it + 1;
        if (!(rr_new > tol2)) ctrl->done = 1;
    }
}
__global__ void k_kc_q(int m, const double *__restrict__ s, const double *__restrict__ p, double *__restrict__ q)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) q[i] = s[i] * p[i];
}
__global__ void k_kc_unscale(int m, double *__restrict__ y, const double *__restrict__ s)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) y[i] = y[i] * s[i];
}

static inline int kc_grid(long long work, int per_block, int cap)
{
    long long b = (work + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (int)b;
}

// Assemble K for the current elements / charges and solve K y = rhs in place in y (warm start = y on entry).
int kcg_assemble_and_solve(int cb, int m, int N_left, const int *element, const int *charge, MetalSet ms, double high_G, double low_G,
                           const int *rp, const int *ci, int nnz, const int *lrp, const int *lci, const int *rrp, const int *rci,
                           double VL, double VR, double *y, int *iters_out, double *rr_out)
{
    Engine &e = eng(); hipStream_t st = e.stream;
    if (m <= 0) { if (iters_out) *iters_out = 0; if (rr_out) *rr_out = 0; return 0; }
    int *cf = (int *)scratch(S_K_DATA, (size_t)nnz * 4);
    double *rhs = (double *)scratch(S_K_RHS, (size_t)m * 8 * 2);
    double *s = (double *)scratch(S_CG_S, (size_t)m * 8), *r = (double *)scratch(S_CG_R, (size_t)m * 8);
    double *p = (double *)scratch(S_CG_P, (size_t)m * 8), *t = (double *)scratch(S_CG_T, (size_t)m * 8);
    double *q = (double *)scratch(S_XT_Q, (size_t)m * 8);
    double *part = (double *)scratch(S_CG_PART, (size_t)3 * 8192 * 8);
    KCtrl *ctrl = (KCtrl *)scratch(S_CG_CTRL, sizeof(KCtrl));
    if (!cf || !rhs || !s || !r || !p || !t || !q || !part || !ctrl) return e.err_code;
    double *diag = rhs + m, *part_pt = part, *part_rr = part + 4096;
    const double tol2 = e.cg_tol * e.cg_tol;
    const int ab = (m + 15) / 16;
    if (cb == 2) hipLaunchKernelGGL((k_kc_assemble<2>), dim3(ab), dim3(KC_NT), 0, st, m, N_left, element, charge, ms, high_G, low_G, rp, ci, lrp, lci, rrp, rci, VL, VR, cf, diag, rhs);
    else if (cb) hipLaunchKernelGGL((k_kc_assemble<1>), dim3(ab), dim3(KC_NT), 0, st, m, N_left, element, charge, ms, high_G, low_G, rp, ci, lrp, lci, rrp, rci, VL, VR, cf, diag, rhs);
    else hipLaunchKernelGGL((k_kc_assemble<0>), dim3(ab), dim3(KC_NT), 0, st, m, N_left, element, charge, ms, high_G, low_G, rp, ci, lrp, lci, rrp, rci, VL, VR, cf, diag, rhs);
    const int vb = (m + 255) / 256;
    hipLaunchKernelGGL(k_kc_scale, dim3(vb), dim3(256), 0, st, m, (const double *)diag, s, rhs, y, q);
    const int ga = kc_grid(m, KC_NT / 8, 2048);          // apply: 32 rows per block and pass
    const int gv = kc_grid(m, KC_NT * 4, 512);
    HIPCHK(hipMemsetAsync(ctrl, 0, sizeof(KCtrl), st));
    hipLaunchKernelGGL((k_kc_apply<1>), dim3(ga), dim3(KC_NT), 0, st, m, rp, (const int *)cf, (const double *)diag, (const double *)s, (const double *)q, high_G, low_G,
                       (const double *)nullptr, t, part_rr, (const KCtrl *)ctrl, (const double *)rhs, r, p);
    hipLaunchKernelGGL(k_kc_q, dim3(vb), dim3(256), 0, st, m, (const double *)s, (const double *)p, q);
    hipLaunchKernelGGL(k_kc_check0, dim3(1), dim3(KC_NT), 0, st, (const double *)part_rr, ga, ctrl, tol2);
    KCHK();
    static hipEvent_t evk[2]; static bool evk_ready = false;
    const bool prof = e.profiling != 0;
    if (prof) {
        if (!evk_ready) { HIPCHK(hipEventCreate(&evk[0])); HIPCHK(hipEventCreate(&evk[1])); evk_ready = true; }
        HIPCHK(hipEventRecord(evk[0], st));
    }
    int it = 0, batch = 8;
    KCtrl h{};
    for (;;) {
        HIPCHK(hipMemcpyAsync(&h, ctrl, sizeof(KCtrl), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (h.done) break;
        if (it >= 200000) { dkmc_fail(4, "CG: no convergence after 200000 iterations", __FILE__, __LINE__); break; }
        for (int b = 0; b < batch; ++b, ++it) {
            hipLaunchKernelGGL((k_kc_apply<0>), dim3(ga), dim3(KC_NT), 0, st, m, rp, (const int *)cf, (const double *)diag, (const double *)s, (const double *)q, high_G, low_G,
                               (const double *)p, t, part_pt, (const KCtrl *)ctrl, (const double *)nullptr, (double *)nullptr, (double *)nullptr);
            hipLaunchKernelGGL(k_kc_update, dim3(gv), dim3(KC_NT), 0, st, m, it, (const double *)part_pt, ga, (const double *)p, (const double *)t, y, r, part_rr, (const KCtrl *)ctrl);
            hipLaunchKernelGGL(k_kc_direction, dim3(gv), dim3(KC_NT), 0, st, m, it, (const double *)part_rr, gv, (const double *)r, p, (const double *)s, q, ctrl, tol2);
        }
        KCHK();
        if (batch < 64) batch *= 2;
    }
    if (prof) {
        HIPCHK(hipEventRecord(evk[1], st));
        HIPCHK(hipEventSynchronize(evk[1]));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, evk[0], evk[1]));
        e.stats.kcg_ms = ms; e.stats.kcg_iters_timed = h.iters;
    } else { e.stats.kcg_ms = 0.0; e.stats.kcg_iters_timed = 0; }
    hipLaunchKernelGGL(k_kc_unscale, dim3(vb), dim3(256), 0, st, m, y, (const double *)s);
    KCHK();
    if (iters_out) *iters_out = h.iters;
    if (rr_out) *rr_out = h.rr[h.iters & 1];
    return e.err_code;
}
