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
