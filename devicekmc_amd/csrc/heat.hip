// heat.hip -- global temperature update.
#include "common.h"

#define HT_NT 256
__global__ __launch_bounds__(HT_NT) void k_power_partials(int N, const double *__restrict__ p, double *__restrict__ part)
{
    __shared__ double red[HT_NT / 64];
    double s = 0.0;
    for (int i = blockIdx.x * HT_NT + threadIdx.x; i < N; i += gridDim.x * HT_NT) s += p[i];
    const double t = block_sum_all<HT_NT>(s, red);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}

// mode 0: heat_solver.cpp:322-334 (what the reference runs, on the host); mode 1: heat_solver_gpu.cu:42-48
__global__ __launch_bounds__(HT_NT) void k_temp_update(const double *__restrict__ part, int npart, double *T_bg, double *P_out, int mode,
                                                       double a0, double a1, double a2, double a3, double a4)
{
    __shared__ double red[HT_NT / 64];
    double s = 0.0;
    for (int i = threadIdx.x; i < npart; i += HT_NT) s += part[i];
    const double P = block_sum_all<HT_NT>(s, red);
    if (threadIdx.x != 0) return;
    if (P_out) *P_out = P;
    if (mode == 0) {
        const double event_time = a0, diss = a1, C = a2;
        const double a = diss / C;
        const double c = (diss / C) * (*T_bg) + (1 / C) * P;
        *T_bg = (c / a) + (*T_bg - c / a) * exp(-a * event_time);
    } else {
        const double a_coeff = a0, b_coeff = a1, number_steps = a2, C_thermal = a3, small_step = a4;
        const double c_coeff = b_coeff + P / C_thermal * small_step;
        const double T_int = *T_bg;
        const int step = (int)number_steps;
        *T_bg = c_coeff * (1.0 - pow(a_coeff, (double)step)) / (1.0 - a_coeff) + pow(a_coeff, (double)step) * T_int;
    }
}

static int temp_update(const double *site_power, double *T_bg, int N, int mode, double a0, double a1, double a2, double a3, double a4, double *h_P)
{
    Engine &e = eng(); hipStream_t st = e.stream;
    int nb = (N + HT_NT - 1) / HT_NT; if (nb > 1024) nb = 1024; if (nb < 1) nb = 1;
    double *part = (double *)scratch(S_HEAT, (size_t)(nb + 2) * 8);
    if (!part) return e.err_code;
    hipLaunchKernelGGL(k_power_partials, dim3(nb), dim3(HT_NT), 0, st, N, site_power, part);
    hipLaunchKernelGGL(k_temp_update, dim3(1), dim3(HT_NT), 0, st, part, nb, T_bg, part + nb, mode, a0, a1, a2, a3, a4);
    KCHK();
    if (h_P) { HIPCHK(hipMemcpyAsync(h_P, part + nb, 8, hipMemcpyDeviceToHost, st)); HIPCHK(hipStreamSynchronize(st)); }
    return 0;
}

extern "C" int dkmc_update_temperatureglobal_gpu(const double *site_power, double *T_bg, int N, double a_coeff, double b_coeff,
                                                 double number_steps, double C_thermal, double small_step)
{
    return temp_update(site_power, T_bg, N, 1, a_coeff, b_coeff, number_steps, C_thermal, small_step, nullptr);
}

extern "C" int dkmc_update_temperature_global_analytic(const double *site_power, double *T_bg, int N, double event_time,
                                                       double dissipation_constant, double t_ox, double A, double c_p, double *h_P_tot)
{
    const double C = A * t_ox * c_p * (1e6);
    return temp_update(site_power, T_bg, N, 0, event_time, dissipation_constant, C, 0, 0, h_P_tot);
}

// =============================================================================================================================
// Local temperature model (heat_solver.cpp:40-246, 286-308, 354-513).  The reference runs it on the host: it inverts the dense
// N_interface x N_interface matrices (I - dt*tau*L) and L once (constructLaplacian) and multiplies by the dense inverses in every
// update -- O(N_interface^2) memory, which stops at a few 1e4 sites.  Here the same linear systems are solved in their sparse
// form with the Jacobi-CG of cg.hip: L is the graph Laplacian of the padded neighbour index restricted to the interface sites
// (1 per neighbour pair, diagonal -gamma*[a neighbour is metallic] - degree), so (I - s*L) and -L are symmetric positive
// definite with ~26 entries per row.
//   transient  :  (I - s L) y = T_vec + P c s ,  T = y (T_1 - T_0) + T_0           (updateLocalTemperature, :354-437)
//   steady     :  (-L) y = P c                ,  T = y (T_1 - T_0) + T_0           (updateLocalTemperatureSteadyState, :441-513:
//                                                                                   T = -(L^-1 P c)(T_1 - T_0) + T_0)
// with c = p_transfer of the site (:369-370) and T_vec = (T - T_0)/(T_1 - T_0).  T_bg = mean of site_temperature over
// [num_atoms_contact, N - num_atoms_contact) (:423-432).  The power vector is gpubuf.site_power, i.e. the one update_power just
// wrote (the reference's GPU build multiplies a host copy that is never refreshed in this branch).
int cg_solve_jacobi(double *a, const int *rp, const int *ci, int nnz, int m, double *x, double *y,
                    int uniform_rows, const int *srank, int ns, int *iters_out, double *rr_out);

#define DKMC_T1 50.0            // Device.h:117

struct HeatPat {
    int N = 0, nn = 0, lo = 0, m = 0, nnz = 0;      // interface sites [lo, lo+m)
    int *rp = nullptr, *ci = nullptr;
    double *diagL = nullptr;                          // diagonal of L per interface row
    const int *neigh = nullptr;
};
static HeatPat g_heat;
static double g_heat_tol = 1e-10;

extern "C" void dkmc_set_heat_cg_tolerance(double tol) { g_heat_tol = tol; }

__global__ void k_heat_count(int m, int lo, int nn, const int *__restrict__ neigh, const int *__restrict__ element, MetalSet ms, double gamma,
                             int *__restrict__ cnt, double *__restrict__ diagL)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= m) return;
    const int *row = neigh + (size_t)(lo + r) * nn;
    int deg = 0; bool bnd = false;
    for (int s = 0; s < nn; ++s) {
        const int j = row[s];
        if (j < 0) continue;
        if (j >= lo && j < lo + m) ++deg;                 // L[ii][jj] = 1 (:120-123)
        bnd |= is_metal(element[j], ms);                  // :125-132, any neighbour, in the interface or not
    }
    cnt[r] = deg + 1;
    diagL[r] = (bnd ? -gamma : 0.0) - (double)deg;        // :142-153
}

__global__ void k_heat_fill(int m, int lo, int nn, const int *__restrict__ neigh, const int *__restrict__ rp, int *__restrict__ ci)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= m) return;
    const int i = lo + r;
    const int *row = neigh + (size_t)i * nn;
    int p = rp[r]; bool diag_done = false;
    for (int s = 0; s < nn; ++s) {
        const int j = row[s];
        if (j < lo || j >= lo + m) continue;
        if (!diag_done && j > i) { ci[p++] = r; diag_done = true; }
        ci[p++] = j - lo;
    }
    if (!diag_done) ci[p++] = r;
}

__global__ void k_set_last_h(int *rp, int m, const int *total) { if (threadIdx.x == 0 && blockIdx.x == 0) rp[m] = *total; }

// Device::constructLaplacian (heat_solver.cpp:40-246) in sparse form.  N_left_tot / N_right_tot: get_num_in_contacts (:5-37),
// computed by the caller from the host copy of site_element; gamma: :86.
extern "C" int dkmc_construct_laplacian(const dkmc_gpubuf *buf, int N_left_tot, int N_right_tot, double gamma)
{
    Engine &e = eng(); hipStream_t st = e.stream;
    HeatPat &h = g_heat;
    const int N = buf->N_, nn = buf->nn_, m = N - N_left_tot - N_right_tot;
    if (N_left_tot < 0 || N_right_tot < 0 || m <= 0) return dkmc_fail(30, "construct_laplacian: no interface sites", __FILE__, __LINE__);
    if (h.rp) (void)hipFree(h.rp);
    if (h.ci) (void)hipFree(h.ci);
    if (h.diagL) (void)hipFree(h.diagL);
    h = HeatPat{};
    HIPCHK(hipMalloc((void **)&h.rp, (size_t)(m + 1) * sizeof(int)));
    HIPCHK(hipMalloc((void **)&h.diagL, (size_t)m * sizeof(double)));
    int *cnt = (int *)scratch(S_MISC0, (size_t)m * sizeof(int));
    int *tot = (int *)scratch(S_MISC1, 4 * sizeof(int));
    if (!cnt || !tot) return e.err_code;
    MetalSet ms = load_metals(buf->metal_types, buf->num_metal_types_);
    const int blocks = (m + 255) / 256;
    hipLaunchKernelGGL(k_heat_count, dim3(blocks), dim3(256), 0, st, m, N_left_tot, nn, (const int *)buf->neigh_idx, (const int *)buf->site_element, ms,
                       gamma, cnt, h.diagL);
    int rc = dkmc_exclusive_scan_i32(cnt, h.rp, m, tot); if (rc) return rc;
    hipLaunchKernelGGL(k_set_last_h, dim3(1), dim3(1), 0, st, h.rp, m, (const int *)tot);
    int nnz = 0;
    HIPCHK(hipMemcpyAsync(&nnz, tot, sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipMalloc((void **)&h.ci, (size_t)nnz * sizeof(int)));
    hipLaunchKernelGGL(k_heat_fill, dim3(blocks), dim3(256), 0, st, m, N_left_tot, nn, (const int *)buf->neigh_idx, (const int *)h.rp, h.ci);
    KCHK();
    HIPCHK(hipStreamSynchronize(st));
    h.N = N; h.nn = nn; h.lo = N_left_tot; h.m = m; h.nnz = nnz; h.neigh = buf->neigh_idx;
    return 0;
}

// matrix values, right-hand side and start vector of one solve.  steady = 0: A = I - s L, b = T_vec + P c s, y0 = T_vec;
// steady = 1: A = -L, b = P c, y0 = T_vec as well (any start vector gives the same solution)
__global__ __launch_bounds__(256) void k_heat_system(int m, int lo, const int *__restrict__ rp, const int *__restrict__ ci, const double *__restrict__ diagL,
                                                     const int *__restrict__ element, const double *__restrict__ T, const double *__restrict__ P,
                                                     double s, int steady, double T0, double pv, double pn,
                                                     double *__restrict__ a, double *__restrict__ b, double *__restrict__ y)
{
    const int LPR = 16;
    const int g = threadIdx.x / LPR, l = threadIdx.x % LPR;
    const int r = blockIdx.x * (256 / LPR) + g;
    if (r >= m) return;
    const double dl = diagL[r];
    for (int p = rp[r] + l; p < rp[r + 1]; p += LPR) {
        const bool dg = ci[p] == r;
        a[p] = steady ? (dg ? -dl : -1.0) : (dg ? 1.0 - s * dl : -s);
    }
    if (l == 0) {
        const int i = lo + r;
        const double tv = (T[i] - T0) / (DKMC_T1 - T0);
        const double c = (element[i] == VACANCY) ? pv : pn;
        b[r] = steady ? P[i] * c : tv + P[i] * c * s;
        y[r] = tv;
    }
}

__global__ void k_heat_store(int m, int lo, const double *__restrict__ y, double T0, double *__restrict__ T)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < m) T[lo + r] = y[r] * (DKMC_T1 - T0) + T0;
}

__global__ __launch_bounds__(HT_NT) void k_mean_final(const double *__restrict__ part, int npart, double denom, double *T_bg)
{
    __shared__ double red[HT_NT / 64];
    double s = 0.0;
    for (int i = threadIdx.x; i < npart; i += HT_NT) s += part[i];
    const double tot = block_sum_all<HT_NT>(s, red);
    if (threadIdx.x == 0) *T_bg = tot / denom;
}

// Device::updateTemperature, local branch (heat_solver.cpp:286-308): steady state when step_time > 1e3 * delta_t, otherwise
// int(step_time / delta_t) + 1 transient updates of length delta_t.  Writes gpubuf.site_temperature and gpubuf.T_bg.
extern "C" int dkmc_update_temperature_local(dkmc_gpubuf *buf, double step_time, double delta_t, double tau, double background_temp,
                                             double k_th_interface, double k_th_vacancies, double nn_dist, int num_atoms_contact,
                                             int *n_solves_out, int *steady_out, int *cg_iters_out, double *T_bg_out)
{
    Engine &e = eng(); hipStream_t st = e.stream;
    HeatPat &h = g_heat;
    if (!h.rp || h.N != buf->N_ || h.nn != buf->nn_ || h.neigh != buf->neigh_idx)
        return dkmc_fail(31, "update_temperature_local: construct_laplacian has not been called for these buffers", __FILE__, __LINE__);
    const int N = h.N, m = h.m;
    if (N - 2 * num_atoms_contact <= 0) return dkmc_fail(32, "update_temperature_local: empty averaging window", __FILE__, __LINE__);
    double *a = (double *)scratch(S_HEAT_A, (size_t)h.nnz * 8), *b = (double *)scratch(S_HEAT_B, (size_t)m * 8), *y = (double *)scratch(S_HEAT_Y, (size_t)m * 8);
    if (!a || !b || !y) return e.err_code;
    const double T0 = background_temp;
    const double pv = 1.0 / ((nn_dist * (1e-10) * k_th_interface) * (DKMC_T1 - background_temp));      // :369 (names as in the reference)
    const double pn = 1.0 / ((nn_dist * (1e-10) * k_th_vacancies) * (DKMC_T1 - background_temp));      // :370
    const int steady = step_time > 1e3 * delta_t;
    const int nsolve = steady ? 1 : (int)(step_time / delta_t) + 1;
    const double s = delta_t * tau;
    const double saved_tol = e.cg_tol;
    e.cg_tol = g_heat_tol;
    int iters_total = 0, rc = 0;
    for (int k = 0; k < nsolve && !rc; ++k) {
        hipLaunchKernelGGL(k_heat_system, dim3((m + 15) / 16), dim3(256), 0, st, m, h.lo, (const int *)h.rp, (const int *)h.ci, (const double *)h.diagL,
                           (const int *)buf->site_element, (const double *)buf->site_temperature, (const double *)buf->site_power, s, steady, T0, pv, pn, a, b, y);
        int it = 0;
        rc = cg_solve_jacobi(a, h.rp, h.ci, h.nnz, m, b, y, 1, nullptr, 0, &it, nullptr);
        iters_total += it;
        if (!rc) hipLaunchKernelGGL(k_heat_store, dim3((m + 255) / 256), dim3(256), 0, st, m, h.lo, (const double *)y, T0, buf->site_temperature);
    }
    e.cg_tol = saved_tol;
    if (rc) return rc;
    // T_bg = mean over [num_atoms_contact, N - num_atoms_contact)
    const int cntN = N - 2 * num_atoms_contact;
    int nb = (cntN + HT_NT - 1) / HT_NT; if (nb > 1024) nb = 1024;
    double *part = (double *)scratch(S_HEAT, (size_t)(nb + 2) * 8);
    if (!part) return e.err_code;
    hipLaunchKernelGGL(k_power_partials, dim3(nb), dim3(HT_NT), 0, st, cntN, (const double *)buf->site_temperature + num_atoms_contact, part);
    hipLaunchKernelGGL(k_mean_final, dim3(1), dim3(HT_NT), 0, st, (const double *)part, nb, (double)cntN, buf->T_bg);
    KCHK();
    if (T_bg_out) { HIPCHK(hipMemcpyAsync(T_bg_out, buf->T_bg, 8, hipMemcpyDeviceToHost, st)); HIPCHK(hipStreamSynchronize(st)); }
    if (n_solves_out) *n_solves_out = nsolve;
    if (steady_out) *steady_out = steady;
    if (cg_iters_out) *cg_iters_out = iters_total;
    return 0;
}
