// tools/attic: the K solve as ONE persistent cooperative launch (round 4), removed from csrc/kcg.hip after measurement.
// Built on the blocked form of K (kcg.hip): one workgroup per block of rows stays resident for the whole solve, padded rows in LDS, vector
// slices in registers, two grid barriers per iteration on a counter in device memory (release / acquire at agent scope by one wave per
// workgroup, relaxed polls, bounded by the wall clock; host falls back to the two-launch loop when a barrier gives up).  Results equal to the
// two-launch loop (2.5 nm device, tol 1e-12: max |phi - phi'| = 1.8e-10, same iteration counts).  Measured on MI355X, 9 111 rows, 141
// workgroups: 23 us per iteration (90 us with acquire-polls and per-wave fences) against 13 us for the two-launch loop of kcg.hip: a grid
// barrier across the 8 XCDs (L2 write-back + invalidate + an atomic round trip through the fabric) costs ~10 us, more than a kernel boundary.
// Not compiled; kept for the record.  Fragments: the kernel, then the host path that launched it.
// ---- the whole solve in ONE launch (blocked form, dkmc_set_k_blocked(2)) -------------------------------------------------------------
// At 1e5 rows an iteration of the two-launch loop is two chains launch -> loads -> reduce of 6-9 us each for data that sits on the chip.
// Here one workgroup per block of rows stays resident for the whole solve: its padded rows live in LDS, its slices of y, r, p, s, d, b in
// registers; per iteration it re-reads only its window of q (written by its neighbours) and the nb x 4 partial sums.  Two grid barriers per
// iteration (after the partial sums, after q) on one counter in device memory, release / acquire at agent scope.  The arithmetic is the
// two-launch loop's (same sums, same order, same stop test).  Launched with hipLaunchCooperativeKernel (all workgroups co-resident or the
// launch is refused); a barrier additionally gives up after KB_BAR_LIMIT ticks of the 100 MHz wall clock -- another process holding CUs
// can delay, never hang it -- and the host then repeats the solve with the two-launch loop from the saved start vector.
#define KB_BAR_LIMIT 20000000LL
__device__ __forceinline__ bool kb_grid_barrier(unsigned *cnt, unsigned target, int *s_to)
{
    // one wave per workgroup does the cache maintenance (the L2 of an XCD and the L1 of a CU are shared by its waves): write back before
    // arriving, invalidate after leaving; the polls themselves are relaxed loads
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const long long t0 = wall_clock64();
        int to = 0;
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (wall_clock64() - t0 > KB_BAR_LIMIT) { to = 1; break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        *s_to = to;
    }
    __syncthreads();
    return *s_to != 0;
}
__global__ __launch_bounds__(KB_NT) void k_kb_solve(int m, int R, const int4 *__restrict__ blk, const int *__restrict__ cf, const double *__restrict__ diag,
                                                    const double *__restrict__ s, const double *__restrict__ b, double *qa, double *qb,
                                                    double *__restrict__ y, double *part, KCtrl *ctrl, unsigned *bar,
                                                    double high_G, double low_G, double tol2, int it_max, int win_cap)
{
    extern __shared__ double win[];                                            // win_cap doubles, then this block's padded rows
    int *cfl = reinterpret_cast<int *>(win + win_cap);
    __shared__ double red[4][KB_NT / 64];
    __shared__ int s_to;
    const int nb = gridDim.x;
    const int4 bi = blk[blockIdx.x];
    const int wlo = bi.x, wn = bi.y;
    const int r0 = blockIdx.x * R, nrows = min(R, m - r0);
    const int g = threadIdx.x >> 2, l = threadIdx.x & 3;
    {
        const int n4 = (bi.w * 64 + (nrows - bi.w) * 32) / 4;
        const int4 *src = reinterpret_cast<const int4 *>(cf + bi.z);
        int4 *dst = reinterpret_cast<int4 *>(cfl);
        for (int i = threadIdx.x; i < n4; i += KB_NT) dst[i] = src[i];
    }
    // rows k = g and g + 256 of the block (every lane of a row's group of four keeps a copy of the row's scalars)
    bool in[2]; int row[2], base[2], width[2];
    double dg[2], sv[2], yv[2], rv[2], pv[2], tv[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int k = g + j * (KB_NT / 4);
        in[j] = k < nrows; row[j] = r0 + (in[j] ? k : 0);
        base[j] = kb_row_base(bi, in[j] ? k : 0, &width[j]) - bi.z;
        dg[j] = diag[row[j]]; sv[j] = s[row[j]]; yv[j] = y[row[j]];
        rv[j] = 0.0; pv[j] = 0.0; tv[j] = 0.0;
    }
    double *prr = part + 3 * KC_NPA;
    unsigned nbar = 0;
    double *qr = qa, *qw = qb;                                                 // the product reads qr; the q formed after it goes to qw
    int it = -1, iters = 0, converged = 0, timed_out = 0;
    double rr_fin = 0.0;
    __syncthreads();
    for (;;) {
        for (int idx = threadIdx.x; idx < wn; idx += KB_NT) win[idx] = qr[wlo + idx];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            double sum = 0.0;
            if (in[j]) {
                const int4 *cr = reinterpret_cast<const int4 *>(cfl + base[j]) + l;
#pragma unroll 1
                for (int h = 0; h < width[j] / 32; ++h) {
                    const int4 c0 = cr[8 * h], c1 = cr[8 * h + 4];
                    const int c[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
                    double x[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) x[u] = win[(c[u] & 0x7fffffff) - wlo];
#pragma unroll
                    for (int u = 0; u < 8; ++u) x[u] = (c[u] & 0x7fffffff) == row[j] ? 0.0 : (c[u] < 0 ? high_G : low_G) * x[u];
                    sum += ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]));
                }
            }
            sum += __shfl_xor(sum, 2, 4); sum += __shfl_xor(sum, 1, 4);
            if (in[j]) tv[j] = sv[j] * (dg[j] * win[row[j] - wlo] - sum);
        }
        double acc[3] = {0.0, 0.0, 0.0};
        if (it < 0) {
#pragma unroll
            for (int j = 0; j < 2; ++j) if (in[j]) { rv[j] = -b[row[j]] + tv[j]; pv[j] = -rv[j]; if (l == 0) { acc[0] += rv[j] * rv[j]; qw[row[j]] = sv[j] * pv[j]; } }
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) if (in[j] && l == 0) { acc[0] += pv[j] * tv[j]; acc[1] += rv[j] * tv[j]; acc[2] += tv[j] * tv[j]; }
        }
        block_sum_n<KB_NT, 3>(acc, red);
        if (threadIdx.x == 0) {
            if (it < 0) prr[blockIdx.x] = acc[0];
            else { part[blockIdx.x] = acc[0]; part[KC_NPA + blockIdx.x] = acc[1]; part[2 * KC_NPA + blockIdx.x] = acc[2]; }
        }
        nbar += nb;
        if (kb_grid_barrier(bar, nbar, &s_to)) { timed_out = 1; break; }
        double v[4] = {0.0, 0.0, 0.0, 0.0};
        if (it < 0) {
            for (int i = threadIdx.x; i < nb; i += KB_NT) v[3] += prr[i];
        } else {
            const double *pin = prr + (it & 1) * KC_NP;
            for (int i = threadIdx.x; i < nb; i += KB_NT) { v[0] += part[i]; v[1] += part[KC_NPA + i]; v[2] += part[2 * KC_NPA + i]; v[3] += pin[i]; }
        }
        block_sum_n<KB_NT, 4>(v, red);
        const double rr = v[3];
        if (it < 0) {
            if (!(sqrt(rr) > tol2)) { converged = 1; iters = 0; rr_fin = rr; break; }       // the first test is on ||r|| (iterative_solvers_gpu.cu:411)
            double *tq = qr; qr = qw; qw = tq;
            it = 0;
            continue;
        }
        if (it > 0 && !(rr > tol2)) { converged = 1; iters = it; rr_fin = rr; break; }
        if (it >= it_max) { iters = it; rr_fin = rr; break; }
        const double alpha = rr / v[0];
        const double rr_new = rr + alpha * (2.0 * v[1] + alpha * v[2]);
        const double beta = rr_new / rr;
        double accr[1] = {0.0};
#pragma unroll
        for (int j = 0; j < 2; ++j) if (in[j]) {
            yv[j] += alpha * pv[j];
            const double rn = rv[j] + alpha * tv[j];
            rv[j] = rn;
            const double pn = pv[j] * beta - rn;
            pv[j] = pn;
            if (l == 0) { accr[0] += rn * rn; qw[row[j]] = sv[j] * pn; }
        }
        block_sum_n<KB_NT, 1>(accr, red);
        if (threadIdx.x == 0) prr[((it + 1) & 1) * KC_NP + blockIdx.x] = accr[0];
        nbar += nb;
        if (kb_grid_barrier(bar, nbar, &s_to)) { timed_out = 1; break; }
        { double *tq = qr; qr = qw; qw = tq; }
        ++it;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) if (in[j] && l == 0) y[row[j]] = yv[j];
    if (threadIdx.x == 0) {
        if (timed_out) ctrl->timeout = 1;
        if (blockIdx.x == 0) { ctrl->rr[0] = rr_fin; ctrl->iters = iters; ctrl->done = converged; }
    }
}


/* ---- host path (inside kcg_assemble_and_solve, before the two-launch loop) ----
    // ---- the whole solve in one persistent launch (k_kb_solve) ----
    const size_t lds_p = kb ? (size_t)kb->maxwin * 8 + (size_t)kb->maxints * 4 : 0;
    if (kb && e.k_blocked >= 2 && kb->R <= 2 * (KB_NT / 4) && lds_p <= 158 * 1024) {
        static int coop = -1, ncu = 0;
        if (coop < 0) {
            int v = 0; coop = hipDeviceGetAttribute(&v, hipDeviceAttributeCooperativeLaunch, e.device) == hipSuccess && v ? 1 : 0;
            if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, e.device) != hipSuccess) ncu = 0;
            if (hipFuncSetAttribute((const void *)k_kb_solve, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024) != hipSuccess) coop = 0;
            (void)hipGetLastError();
        }
        int per_cu = 0;
        if (coop && hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_kb_solve, KB_NT, lds_p) != hipSuccess) { per_cu = 0; (void)hipGetLastError(); }
        if (coop && (long long)per_cu * ncu >= kb->nb) {
            double *ysave = (double *)scratch(S_K_SAVE, (size_t)m * 8 + 64);
            if (!ysave) return e.err_code;
            unsigned *bar = reinterpret_cast<unsigned *>(ysave + m);
            HIPCHK(hipMemcpyAsync(ysave, y, (size_t)m * 8, hipMemcpyDeviceToDevice, st));
            HIPCHK(hipMemsetAsync(bar, 0, 64, st));
            HIPCHK(hipMemsetAsync(ctrl, 0, sizeof(KCtrl), st));
            HIPCHK(hipMemsetAsync(part, 0, (size_t)KC_PART_DOUBLES * 8, st));
            int m_ = m, R_ = kb->R, itmax = 200000, wcap = kb->maxwin;
            const int4 *blk_ = kb->blk; const int *cf_ = cf; const double *diag_ = diag, *s_ = s, *b_ = rhs;
            double *qa_ = q, *qb_ = t, *y_ = y, *part_ = part; KCtrl *ctrl_ = ctrl; double hg = high_G, lg = low_G, tl = tol2;
            void *args[] = {&m_, &R_, &blk_, &cf_, &diag_, &s_, &b_, &qa_, &qb_, &y_, &part_, &ctrl_, &bar, &hg, &lg, &tl, &itmax, &wcap};
            if (prof) HIPCHK(hipEventRecord(evk[0], st));
            hipError_t lrc = hipLaunchCooperativeKernel((const void *)k_kb_solve, dim3(kb->nb), dim3(KB_NT), args, (unsigned)lds_p, st);
            if (lrc == hipSuccess) {
                if (prof) HIPCHK(hipEventRecord(evk[1], st));
                HIPCHK(hipMemcpyAsync(&h, ctrl, sizeof(KCtrl), hipMemcpyDeviceToHost, st));
                HIPCHK(hipStreamSynchronize(st));
                if (h.done && !h.timeout) {
                    solved = true;
                    if (prof) { float ms = 0.f; HIPCHK(hipEventElapsedTime(&ms, evk[0], evk[1])); e.stats.kcg_ms = ms; e.stats.kcg_iters_timed = h.iters; }
                }
            } else (void)hipGetLastError();
            if (!solved) {            // refused, gave up at a barrier or ran out of iterations: the two-launch loop from the saved start vector
                HIPCHK(hipMemcpyAsync(y, ysave, (size_t)m * 8, hipMemcpyDeviceToDevice, st));
                hipLaunchKernelGGL(k_kc_q, dim3(vb), dim3(256), 0, st, m, (const double *)s, (const double *)y, q);
                e.stats.kcg_persistent_fallbacks += 1;
            }
        }
    }
*/


// ======================================================================================================================================
// Second variant (round 4, also removed after measurement): ONE launch per iteration on the blocked form.  rocprofv3 kernel durations on
// MI355X, tol 1e-12: 9 111 rows: k_kb_iter 10.1 us against k_kb_apply 6.5 + k_kc_step 4.2; 82 479 rows: 16.5 us against 8.7 + 6.9 -- the
// merged launch takes what the two take together (HIP events per iteration: 13.5 vs 15.3 us and 17.1 vs 15.5 us): the iteration is a chain of
// dependent memory latencies and workgroup-wide reductions, not launch overhead, and forming q' for the whole window in every workgroup adds
// work.  Fragments: the kernel, then the two host fragments.
// ONE launch per iteration on the blocked form (k_kb_iter = the step of iteration c followed by the product of iteration c + 1).  A block
// needs q' = s p' on its whole WINDOW before it can multiply, and p' = beta p - (r + alpha t) is elementwise: so every workgroup forms q' for
// its window itself, redundantly (17 x its own rows at the crossbar: 4 coalesced reads of 61 KB instead of 1, from L2), straight into LDS --
// q never exists in memory -- and writes y, r', p' for its own rows only.  p, r, t and the partial sums alternate between two buffers (a
// neighbour may still read the old ones).  The stop test is the reference's, on the summed r.r of the r this launch STARTS from (formed and
// summed by its predecessor): launch c tests "after update c" and then changes nothing.  Per iteration: one launch of ~9 us instead of two
// of 6-9 us.
__global__ __launch_bounds__(KB_NT) void k_kb_iter(int m, int R, int c, const int4 *__restrict__ blk, const int *__restrict__ cf, const double *__restrict__ diag,
                                                   const double *__restrict__ s, const double *__restrict__ Pold, const double *__restrict__ Rold,
                                                   const double *__restrict__ Told, double *__restrict__ Pnew, double *__restrict__ Rnew, double *__restrict__ Tnew,
                                                   double *__restrict__ y, const double *__restrict__ part_in, double *__restrict__ part_out,
                                                   const double *__restrict__ prr_in, double *__restrict__ prr_out, KCtrl *ctrl,
                                                   double high_G, double low_G, double tol2, int npa, int wcap)
{
    extern __shared__ double win[];                                            // wcap doubles of q', then p' and r' of the block's own rows
    double *ps = win + wcap, *rs = ps + R;
    __shared__ double red[4][KB_NT / 64];
    __shared__ int sdone;
    const int4 bi = blk[blockIdx.x];
    const int wlo = bi.x, wn = bi.y;
    const int r0 = blockIdx.x * R, nrows = min(R, m - r0);
    const int g = threadIdx.x >> 2, l = threadIdx.x & 3;
    if (threadIdx.x == 0) sdone = ctrl->done;
    // requested before the partial sums are reduced: the first four window elements of this thread
    double pw[4], rw[4], tw[4], sw[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int idx = threadIdx.x + u * KB_NT;
        const bool ok = idx < wn;
        const int w = wlo + (ok ? idx : 0);
        pw[u] = Pold[w]; rw[u] = Rold[w]; tw[u] = Told[w]; sw[u] = s[w];
    }
    double v[4] = {0.0, 0.0, 0.0, 0.0};
    for (int j = threadIdx.x; j < npa; j += KB_NT) { v[0] += part_in[j]; v[1] += part_in[KC_NPA + j]; v[2] += part_in[2 * KC_NPA + j]; }
    for (int j = threadIdx.x; j < KC_NP; j += KB_NT) v[3] += prr_in[j];
    block_sum_n<KB_NT, 4>(v, red);
    if (sdone) return;
    const double rr = v[3];
    if (c > 0 && !(rr > tol2)) {
        if (blockIdx.x == 0 && threadIdx.x == 0) { ctrl->rr[0] = rr; ctrl->iters = c; ctrl->done = 1; }
        return;
    }
    const double alpha = rr / v[0];
    const double rr_new = rr + alpha * (2.0 * v[1] + alpha * v[2]);
    const double beta = rr_new / rr;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int base = 0; base < wn; base += 4 * KB_NT) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = base + threadIdx.x + u * KB_NT;
            if (idx < wn) {
                const int w = wlo + idx;
                const double rn = rw[u] + alpha * tw[u];
                const double pn = pw[u] * beta - rn;
                win[idx] = sw[u] * pn;
                const int kk = w - r0;
                if (kk >= 0 && kk < nrows) {
                    y[w] += alpha * pw[u];
                    Rnew[w] = rn; Pnew[w] = pn;
                    ps[kk] = pn; rs[kk] = rn;
                    acc[3] += rn * rn;
                }
            }
        }
        if (base + 4 * KB_NT < wn) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + 4 * KB_NT + threadIdx.x + u * KB_NT;
                const int w = wlo + (idx < wn ? idx : 0);
                pw[u] = Pold[w]; rw[u] = Rold[w]; tw[u] = Told[w]; sw[u] = s[w];
            }
        }
    }
    int4 cc[4];
    bool lg;
    int k = g;
    kb_load_row(cf, bi, k, l, k < nrows, cc, &lg);                             // (in flight across the barrier)
    __syncthreads();
    for (; k < nrows; k += KB_NT / 4) {
        const int row = r0 + k;
        const double qr = win[row - wlo], dg = diag[row], sv = s[row];
        int cidx[16];
#pragma unroll
        for (int j = 0; j < 4; ++j) { cidx[4 * j] = cc[j].x; cidx[4 * j + 1] = cc[j].y; cidx[4 * j + 2] = cc[j].z; cidx[4 * j + 3] = cc[j].w; }
        const bool lgc = lg;
        double x[16];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = win[(cidx[u] & 0x7fffffff) - wlo];
        if (lgc) {
#pragma unroll
            for (int u = 8; u < 16; ++u) x[u] = win[(cidx[u] & 0x7fffffff) - wlo];
        }
        kb_load_row(cf, bi, k + KB_NT / 4, l, k + KB_NT / 4 < nrows, cc, &lg);
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = (cidx[u] & 0x7fffffff) == row ? 0.0 : (cidx[u] < 0 ? high_G : low_G) * x[u];
        double sum = ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]));
        if (lgc) {
#pragma unroll
            for (int u = 8; u < 16; ++u) x[u] = (cidx[u] & 0x7fffffff) == row ? 0.0 : (cidx[u] < 0 ? high_G : low_G) * x[u];
            sum += ((x[8] + x[9]) + (x[10] + x[11])) + ((x[12] + x[13]) + (x[14] + x[15]));
        }
        sum += __shfl_xor(sum, 2, 4); sum += __shfl_xor(sum, 1, 4);
        if (l == 0) {
            const double tv = sv * (dg * qr - sum);
            Tnew[row] = tv;
            acc[0] += ps[k] * tv; acc[1] += rs[k] * tv; acc[2] += tv * tv;
        }
    }
    block_sum_n<KB_NT, 4>(acc, red);
    if (threadIdx.x == 0) {
        part_out[blockIdx.x] = acc[0]; part_out[KC_NPA + blockIdx.x] = acc[1]; part_out[2 * KC_NPA + blockIdx.x] = acc[2];
        prr_out[blockIdx.x] = acc[3];
        if (blockIdx.x == 0) { ctrl->rr[0] = rr_new; ctrl->iters = c + 1; }
    }
}


/* ---- host fragments ----
    // blocked form: ONE launch per iteration (k_kb_iter) after a first product; else product + step
    const bool one_launch = kb && e.k_blocked >= 2;
    double *set1 = nullptr, *part1 = part + 3 * KC_NPA + 2 * KC_NP, *prr = part + 3 * KC_NPA;
    size_t lds_i = 0;
    if (one_launch) {
        set1 = (double *)scratch(S_K_PING, (size_t)3 * m * 8);
        if (!set1) return e.err_code;
        lds_i = ((size_t)kb->maxwin + 2 * (size_t)kb->R) * 8;
        static bool attr_i = false;
        if (!attr_i) { HIPCHK(hipFuncSetAttribute((const void *)k_kb_iter, hipFuncAttributeMaxDynamicSharedMemorySize, (KB_MAXWIN + 2 * 1024) * 8)); attr_i = true; }
        KC_APPLY(0, (const int *)cf, (const double *)diag, (const double *)s, (const double *)q, high_G, low_G,
                 (const double *)p, t, part, (const KCtrl *)ctrl, (const double *)nullptr, r, (double *)nullptr);
    }

...
            if (one_launch) {
                const int o = it & 1;
                double *P0 = o ? set1 : p, *R0 = o ? set1 + m : r, *T0 = o ? set1 + 2 * (size_t)m : t;
                double *P1 = o ? p : set1, *R1 = o ? r : set1 + m, *T1 = o ? t : set1 + 2 * (size_t)m;
                hipLaunchKernelGGL(k_kb_iter, dim3(ga), dim3(KB_NT), lds_i, st, m, kb->R, it, (const int4 *)kb->blk, (const int *)cf, (const double *)diag, (const double *)s,
                                   (const double *)P0, (const double *)R0, (const double *)T0, P1, R1, T1, y, (const double *)(o ? part1 : part), o ? part : part1,
                                   (const double *)(prr + o * KC_NP), prr + (o ^ 1) * KC_NP, ctrl, high_G, low_G, tol2, npa, kb->maxwin);
                continue;
            }
*/
