// ATTIC (not compiled): the two-launch CG loop on the tiled X (dkmc_set_x_loop(1)) as it stood at commit 1c8b2b2 (round 3).
// Measured 20 % slower than the three-launch loop and removed from csrc/xt.hip in round 4: A/B record in
// profiles/r03_ab_cg_loop_two_vs_three_launches.json (85 071 sites: 31.1 against 39.1 steps/s; 234 975 sites: 717 against 612 ms per step).
// The kernels below relied on the UV = 1 template paths of xt_tile_role / xt_neigh_roles (q = beta * U - V formed on the fly), also removed;
// `git show 1c8b2b2:devicekmc_amd/csrc/xt.hip` holds the complete, tested version (tests/test_gpu_parity.py::test_two_launch_loop_agrees_with_three_launch_loop there).

// ---- two launches per CG iteration (single GPU; dkmc_set_x_loop(1), NOT the default: measured slower, see the end of this comment) ----
// Iteration k of solve_sparse_CG_Jacobi (iterative_solvers_gpu.cu:405-455): t = A p_k; alpha = r.r / p.t; y += alpha p; r' = r + alpha t;
// beta = r'.r' / r.r; p' = beta p - r'.  The reference needs the two dot products as host-synchronised reductions; the three-launch
// loop above (product, row sums + dots, vector step) needs the second one -- r'.r', wanted for beta before r' exists everywhere -- as a
// recurrence.  Here the loop is cut where the data dependencies allow a cut without any reduction on the critical path:
//   k_xt_apply2(k)    every wave first sums the r.r partials of the two previous fold/step launches (fixed order, every wave the same
//                     bits): beta_k and the stop test, uniform over the launch without a flag.  The direction is not stored yet:
//                     the product is taken of q = S p_k = beta_k U - V formed on the fly (U = S p_{k-1}, V = S r_k, written by the
//                     previous fold/step launch).  p_k.t = q' X q is bilinear in what the product already holds: every tile wave adds
//                     q_row * (partial row sum) over its tiles (each stored entry of the upper triangle once, doubled at the end), every
//                     neighbour row q_row * (row sum): one partial per workgroup, no fold needed for alpha.
//   k_xt_fold_step(k) alpha_k from those partials; per S row block the fold of the tile partial sums (as k_xt_rows), then for every row
//                     p_k = beta_k p_{k-1} - r_k (stored now), y += alpha p_k, r_{k+1} = r_k + alpha t, U = S p_k, V = S r_{k+1} and the DIRECT
//                     partial sums of r_{k+1}^2 for the next beta.
// Both dot products are direct sums like the reference's (no recurrence), and one kernel boundary per iteration is gone.  Results equal
// the three-launch loop to rounding.  MEASURED (MI355X, same box, bench.py --x-loop 0 / 1): 85 071 sites 24.0 + 6.3 + 5.4 us per iteration
// (three launches) against 32.2 + 13.4 us (two): 39.1 against 31.1 steps/s; 234 975 sites 612 against 717 ms per step.  The boundary that
// disappears costs less than what the two remaining kernels gain in dependent latency (the scalar reductions in front of every wave, the
// second vector stream of the on-the-fly direction, seven stores per row in the fold kernel).  Kept as a tested alternative, off by default.
#define XT_RRS 1024              // distance between the three r.r partial arrays of the two-launch loop
__device__ __forceinline__ double xt_wave_total(const double *__restrict__ part, int n)
{
    double s = 0.0;
    for (int i = threadIdx.x & 63; i < n; i += 64) s += part[i];
    return wave_sum_all(s);
}
// beta and the stop test of iteration it (first test on ||r||, later ones on ||r||^2, both against tol^2: iterative_solvers_gpu.cu:418,448)
// ctrl->done makes the stop sticky: the launches the host has enqueued beyond the converging iteration find the partial arrays of the
// rotation stale.  It is written by workgroup 0 of the k_xt_apply2 launch that detects convergence; a workgroup of that same launch that
// already sees it has reached the same verdict from the partial sums (no decision depends on the timing of the write).
__device__ __forceinline__ void xt_iter_head(const double *__restrict__ part_rr, int it, int n_cur, int n_prev, double tol2, double &beta, double &rr, bool &stop,
                                             const XCtrl *ctrl)
{
    const int was_done = ctrl->done;
    // three partial arrays in rotation: the fold/step launch of iteration it reads those of it and it - 1 while its own workgroups
    // already write those of it + 1
    rr = xt_wave_total(part_rr + XT_RRS * (it % 3), n_cur);
    if (it == 0) { beta = 0.0; stop = !(sqrt(rr) > tol2); }
    else { const double rr_old = xt_wave_total(part_rr + XT_RRS * ((it + 2) % 3), n_prev); beta = rr / rr_old; stop = !(rr > tol2); }
    if (was_done) stop = true;
}
// workgroup roles as in k_xt_apply (vb0: role offset of the first workgroup; a multi-GB sweep launches the tile roles alone, vb0 = 2, nsb = 0,
// and the neighbour part as k_xt_neigh2).  ppart: [0, ntb) tile workgroups, [ntb, ntb + nsb) neighbour workgroups, then the two driver rows.
template <int NTL>
__global__ __launch_bounds__(XT_NT) void k_xt_apply2(int nitems, const XItem *__restrict__ items, const XTile *__restrict__ tiles, int sub_base,
                                                     const double *__restrict__ tval, const double *__restrict__ US, const double *__restrict__ VS,
                                                     int nW, int ns_pad, double *__restrict__ rowpart, double *__restrict__ colpart, XCtrl *ctrl,
                                                     int ntb, int nsb, int Nsub, const xrp_t *__restrict__ rp, const int *__restrict__ ci,
                                                     const double *__restrict__ val, const double *__restrict__ U, const double *__restrict__ V,
                                                     const double *__restrict__ sc, const int *__restrict__ nsrank, double *__restrict__ t, int vb0,
                                                     const double *__restrict__ part_rr, int it, int n_cur, int n_prev, double tol2, double *__restrict__ ppart)
{
    __shared__ double red[XT_NT / 64];
    __shared__ __attribute__((aligned(16))) double lcol[XT_NT / 64][2 * XT_C];
    const int vb = (int)blockIdx.x + vb0;
    int tile_idx = -1, nb_idx = -1;
    if (vb >= 2) {
        const int i = vb - 2, nmix = NTL ? 0 : (min(ntb, nsb) >> 3) << 3;
        if (i < 2 * nmix) { const int grp = i >> 3, idx = ((grp >> 1) << 3) + (i & 7); if (grp & 1) nb_idx = idx; else tile_idx = idx; }
        else { const int j = i - 2 * nmix; if (j < ntb - nmix) tile_idx = nmix + j; else nb_idx = nmix + j - (ntb - nmix); }
    }
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    XItem itm{};
    int item = 0;
    if (tile_idx >= 0) { item = tile_idx * (XT_NT / 64) + wv; itm = items[min(item, nitems - 1)]; }      // in flight behind the partial sums
    double beta, rr; bool stop;
    xt_iter_head(part_rr, it, n_cur, n_prev, tol2, beta, rr, stop, ctrl);
    if (blockIdx.x == 0 && threadIdx.x == 0 && !ctrl->done) {       // rr / iters: for the host's poll only
        ctrl->rr[0] = rr; ctrl->rr[1] = rr; ctrl->iters = it;
        if (stop) ctrl->done = 1;
    }
    if (stop) return;
    if (tile_idx >= 0) {
        double pt = 0.0;
        if (item < nitems)
            xt_tile_role<0, NTL, 1>(itm, tiles, sub_base, tval, US, nW, ns_pad, rowpart, colpart, true, lcol[wv], lcol[wv] + XT_C, VS, beta, &pt);
        const double tot = block_sum_all<XT_NT>(pt, red);
        if (threadIdx.x == 0) ppart[tile_idx] = 2.0 * tot;         // upper triangle stored: every pair counted once
        return;
    }
    const int role = vb < 2 ? nsb + vb : nb_idx;
    const double pt = xt_neigh_roles<XT_RPG_FUSED, 1>(role, nsb, Nsub, rp, ci, val, U, sc, nsrank, nullptr, t, red, V, beta);
    if (role < nsb) { const double tot = block_sum_all<XT_NT>(pt, red); if (threadIdx.x == 0) ppart[ntb + role] = tot; }
    else if (threadIdx.x == 0) ppart[ntb + role] = pt;
}
// the neighbour part alone, behind a tiles-only k_xt_apply2 (multi-GB sweeps: see k_xt_neigh)
__global__ __launch_bounds__(XT_NT) void k_xt_neigh2(int nsb, int Nsub, const xrp_t *__restrict__ rp, const int *__restrict__ ci, const double *__restrict__ val,
                                                     const double *__restrict__ U, const double *__restrict__ V, const double *__restrict__ sc,
                                                     const int *__restrict__ nsrank, double *__restrict__ t,
                                                     const double *__restrict__ part_rr, int it, int n_cur, int n_prev, double tol2, double *__restrict__ ppart_nb,
                                                     const XCtrl *ctrl)
{
    __shared__ double red[XT_NT / 64];
    const int vb = (int)blockIdx.x;
    double beta, rr; bool stop;
    xt_iter_head(part_rr, it, n_cur, n_prev, tol2, beta, rr, stop, ctrl);
    if (stop) return;
    const int role = vb < 2 ? nsb + vb : vb - 2;
    const double pt = xt_neigh_roles<1, 1>(role, nsb, Nsub, rp, ci, val, U, sc, nsrank, nullptr, t, red, V, beta);
    if (role < nsb) { const double tot = block_sum_all<XT_NT>(pt, red); if (threadIdx.x == 0) ppart_nb[role] = tot; }
    else if (threadIdx.x == 0) ppart_nb[role] = pt;
}

// second launch of the two-launch loop: fold + every vector update (see k_xt_apply2).  Grid and row-block ownership as k_xt_rows<0>.
__global__ __launch_bounds__(XT_NT) void k_xt_fold_step(int ns, int nK, int nW, int ns_pad, const int2 *__restrict__ wrange, const int *__restrict__ nitem_w,
                                                        const double *__restrict__ rowpart, const double *__restrict__ colpart,
                                                        const int *__restrict__ srow, const double *__restrict__ sS, int m, const int *__restrict__ nsrank,
                                                        const double *__restrict__ sc, const double *__restrict__ t, double *__restrict__ P, double *__restrict__ R,
                                                        double *__restrict__ y, double *__restrict__ U, double *__restrict__ V, double *__restrict__ US,
                                                        double *__restrict__ VS, const double *__restrict__ ppart, int np1, double *__restrict__ part_rr,
                                                        int it, int n_cur, int n_prev, double tol2, const XCtrl *ctrl)
{
    __shared__ double red[XT_NT / 64];
    __shared__ double sl_sum[8][XT_R];
    __shared__ double scal[4];
    // what does not depend on the scalars or the partial sums is fetched first (as in k_xt_rows)
    const int chunk = (m + (int)gridDim.x - 1) / (int)gridDim.x;
    const int i0 = blockIdx.x * chunk, i1 = min(m, i0 + chunk);
    const int ifirst = i0 + (int)threadIdx.x;
    double fP = 0.0, fR = 0.0, fy = 0.0, ft = 0.0, fs = 0.0; int fsr = 0;
    if (ifirst < i1) { fsr = nsrank[ifirst]; fP = P[ifirst]; fR = R[ifirst]; fy = y[ifirst]; ft = t[ifirst]; fs = sc[ifirst]; }
    int row0 = -1; double s0 = 0.0, t0 = 0.0, P0 = 0.0, R0 = 0.0, y0 = 0.0, sc0 = 0.0;
    {
        const int s = XT_R * (int)blockIdx.x + (int)threadIdx.x;
        if ((int)blockIdx.x < nK && threadIdx.x < XT_R && s < ns) { row0 = srow[s]; s0 = sS[s]; t0 = t[row0]; P0 = P[row0]; R0 = R[row0]; y0 = y[row0]; sc0 = sc[row0]; }
    }
    int2 wr0 = make_int2(0, 0); int nc0 = 0;
    int cb0 = 0;
    if ((int)blockIdx.x < nK) { wr0 = wrange[blockIdx.x]; nc0 = xt_row_block_runs(blockIdx.x, nitem_w, 0, nW, nW); cb0 = xt_row_block_cbase(blockIdx.x, nW, nitem_w); }
    if (threadIdx.x < 64) {
        double beta, rr; bool stop;
        xt_iter_head(part_rr, it, n_cur, n_prev, tol2, beta, rr, stop, ctrl);
        const double pAp = xt_wave_total(ppart, np1);
        if (threadIdx.x == 0) { scal[0] = stop ? 1.0 : 0.0; scal[1] = beta; scal[2] = rr / pAp; }
    }
    __syncthreads();
    if (scal[0] != 0.0) return;
    const double beta = scal[1], alpha = scal[2];
    double acc = 0.0;
#define XT_UPDATE(i_, P_, R_, y_, t_, s_, sidx_)                                         \
    {                                                                                    \
        const double pn_ = beta * (P_) - (R_);                                           \
        P[i_] = pn_; y[i_] = (y_) + alpha * pn_;                                         \
        const double rn_ = (R_) + alpha * (t_);                                          \
        R[i_] = rn_; acc += rn_ * rn_;                                                   \
        const double u_ = (s_) * pn_, v_ = (s_) * rn_;                                   \
        U[i_] = u_; V[i_] = v_;                                                          \
        if ((sidx_) >= 0) { US[sidx_] = u_; VS[sidx_] = v_; }                            \
    }
    for (int k = blockIdx.x; k < nK; k += gridDim.x) {
        const bool own = k == (int)blockIdx.x;
        const double sum = xt_row_block_sum(k, nW, own ? cb0 : xt_row_block_cbase(k, nW, nitem_w), own ? wr0 : wrange[k], own ? nc0 : xt_row_block_runs(k, nitem_w, 0, nW, nW), rowpart, colpart, sl_sum, 0, nW);
        const int s = XT_R * k + (int)threadIdx.x;
        if (threadIdx.x < XT_R && s < ns) {
            if (own) { const double tv = s0 * (t0 + sum); XT_UPDATE(row0, P0, R0, y0, tv, sc0, s) }
            else { const int row = srow[s]; const double tv = sS[s] * (t[row] + sum); XT_UPDATE(row, P[row], R[row], y[row], tv, sc[row], s) }
        }
    }
    // the non-S rows of this workgroup's share of the vector (finished, scaled, by the product launch)
    if (ifirst < i1 && fsr < 0) XT_UPDATE(ifirst, fP, fR, fy, ft, fs, -1)
    for (int i = ifirst + XT_NT; i < i1; i += XT_NT) if (nsrank[i] < 0) XT_UPDATE(i, P[i], R[i], y[i], t[i], sc[i], -1)
#undef XT_UPDATE
    const double tot = block_sum_all<XT_NT>(acc, red);
    if (threadIdx.x == 0) part_rr[XT_RRS * ((it + 1) % 3) + blockIdx.x] = tot;
}
// state of the two-launch loop before iteration 0: p_{-1} = 0 (beta_0 = 0 makes p_0 = -r_0), U = 0, V = S r_0
__global__ void k_xt_uv_init(int m, const double *__restrict__ r, const double *__restrict__ sc, const int *__restrict__ nsrank, double *__restrict__ P,
                             double *__restrict__ U, double *__restrict__ V, double *__restrict__ US, double *__restrict__ VS)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const double v = sc[i] * r[i];
    P[i] = 0.0; U[i] = 0.0; V[i] = v;
    const int sr = nsrank[i];
    if (sr >= 0) { US[sr] = 0.0; VS[sr] = v; }
}

