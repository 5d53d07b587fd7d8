/*
 * gpu_solvers.h -- drop-in for the live part of the reference's gpu_solvers.h (:36-208): the same function names and
 * argument lists, implemented as thin inline forwards to the C ABI (devicekmc_hip.h).  The two CUDA library handles of
 * the reference signatures are carried as opaque dkmc_handle_t values and ignored.
 *
 * Differences a caller can observe:
 *   - execute_kmc_step_gpu draws its random numbers in batches: it copies the generator, hands a batch of numbers
 *     to the device loop, then advances the caller's generator by exactly the 2 numbers per executed event the
 *     reference would have drawn (kmc_events.cu:221,348), so the stream position afterwards is identical;
 *   - the dense variants (background_potential_gpu, update_power_gpu) are not provided: they are unreachable in the reference
 *     (potential_solver.cpp:238, current_solver.cpp:21 hard-code the sparse path); update_power_gpu_split IS provided (below).
 */
#pragma once
#include <vector>
#include "gpu_buffers.h"

/* Linkage of the functions below.  Default: inline forwards (header-only shim; the host is recompiled against this header).
 * devicekmc_amd/host/shim_exports.cpp compiles the same bodies with DKMC_SHIM_API = extern "C" + default visibility into
 * libdevicekmc_shim.so, which then EXPORTS the reference's unmangled names (gpu_solvers.h:36-208 declares them extern "C") for
 * host objects that were compiled against a declaration-only copy of this header (DKMC_SHIM_DECLARATIONS_ONLY). */
#ifndef DKMC_SHIM_API
#define DKMC_SHIM_API inline
#endif

DKMC_SHIM_API void get_gpu_info(char *gpu_string, int dev) { GPUBuffers::report(dkmc_get_gpu_info(gpu_string, 1000, dev)); }   // kmc_events.cu:15
DKMC_SHIM_API void set_gpu(int dev) { GPUBuffers::report(dkmc_set_gpu(dev)); }                                                 // kmc_events.cu:30

DKMC_SHIM_API void copytoConstMemory(std::vector<double> E_gen, std::vector<double> E_rec, std::vector<double> E_Vdiff, std::vector<double> E_Odiff)
{
    GPUBuffers::report(dkmc_copy_to_const_memory(E_gen.data(), E_rec.data(), E_Vdiff.data(), E_Odiff.data(), (int)E_gen.size()));
}

DKMC_SHIM_API void initialize_sparsity(GPUBuffers &gpubuf, int pbc, const double nn_dist, int num_atoms_contact)
{
    GPUBuffers::report(dkmc_initialize_sparsity(&gpubuf, pbc, nn_dist, num_atoms_contact));
}

DKMC_SHIM_API void update_CB_edge_gpu_sparse(dkmc_handle_t, dkmc_handle_t, GPUBuffers &gpubuf, const int N, const int N_left_tot,
                                      const int N_right_tot, const double d_Vd, const int pbc, const double d_high_G,
                                      const double d_low_G, const double nn_dist, const int num_metals)
{
    GPUBuffers::report(dkmc_update_CB_edge_gpu_sparse(&gpubuf, N, N_left_tot, N_right_tot, d_Vd, pbc, d_high_G, d_low_G, nn_dist, num_metals));
}

DKMC_SHIM_API void update_charge_gpu(ELEMENT *gpu_site_element, int *gpu_site_charge, int *gpu_neigh_idx, int N, int nn,
                              const ELEMENT *metals, const int num_metals)
{
    GPUBuffers::report(dkmc_update_charge_gpu(reinterpret_cast<int *>(gpu_site_element), gpu_site_charge, gpu_neigh_idx, N, nn,
                                              reinterpret_cast<const int *>(metals), num_metals));
}

DKMC_SHIM_API void background_potential_gpu_sparse(dkmc_handle_t, dkmc_handle_t, GPUBuffers &gpubuf, const int N, const int N_left_tot,
                                            const int N_right_tot, const double d_Vd, const int pbc, const double d_high_G,
                                            const double d_low_G, const double nn_dist, const int num_metals, int kmc_step_count)
{
    GPUBuffers::report(dkmc_background_potential_gpu_sparse(&gpubuf, N, N_left_tot, N_right_tot, d_Vd, pbc, d_high_G, d_low_G, nn_dist,
                                                            num_metals, kmc_step_count));
}

DKMC_SHIM_API void poisson_gridless_gpu(const int num_atoms_contact, const int pbc, const int N, const double *lattice, const double *sigma,
                                 const double *k, const double *posx, const double *posy, const double *posz, const int *site_charge,
                                 double *site_potential_charge)
{
    GPUBuffers::report(dkmc_poisson_gridless_gpu(num_atoms_contact, pbc, N, lattice, sigma, k, posx, posy, posz, site_charge, site_potential_charge));
}

DKMC_SHIM_API void solve_sparse_CG_Jacobi(dkmc_handle_t, dkmc_handle_t, double *A_data, int *A_row_ptr, int *A_col_indices, const int A_nnz,
                                   int m, double *d_x, double *d_y)
{
    GPUBuffers::report(dkmc_solve_sparse_CG_Jacobi(A_data, A_row_ptr, A_col_indices, A_nnz, m, d_x, d_y, nullptr, nullptr));
}

// declared at gpu_solvers.h:66 of the reference; there the body ends in exit(1) and the caller (update_power_gpu_split) is never reached
DKMC_SHIM_API void solve_sparse_CG_splitmatrix(dkmc_handle_t, dkmc_handle_t, double *M, int msub, double *A_data, int *A_row_ptr, int *A_col_indices,
                                               const int A_nnz, int m, int *insertion_indices, double *d_x, double *d_y)
{
    GPUBuffers::report(dkmc_solve_sparse_CG_splitmatrix(M, msub, A_data, A_row_ptr, A_col_indices, A_nnz, m, insertion_indices, 2, d_x, d_y, 1e-5,
                                                        nullptr, nullptr));
}

DKMC_SHIM_API double execute_kmc_step_gpu(const int N, const int nn, const int *neigh_idx, const int *site_layer, const double *lattice,
                                   const int pbc, const double *T_bg, const double *freq, const double *sigma, const double *k,
                                   const double *posx, const double *posy, const double *posz, const double *site_potential_boundary,
                                   const double *site_potential_charge, const double *site_temperature, ELEMENT *site_element,
                                   int *site_charge, RandomNumberGenerator &rng, const int * /*neigh_idx_host*/)
{
    const int batch = 64;                       // events worth of numbers per device launch
    double event_time = 0.0;
    int resume = 0;
    for (;;) {
        RandomNumberGenerator probe = rng;       // std::mt19937 is copyable: peek ahead without consuming
        std::vector<double> u(2 * batch);
        for (auto &v : u) v = probe.getRandomNumber();
        int n_events = 0, exhausted = 0;
        GPUBuffers::report(dkmc_execute_kmc_step_gpu(N, nn, neigh_idx, site_layer, lattice, pbc, T_bg, freq, sigma, k, posx, posy, posz,
                                                     site_potential_boundary, site_potential_charge, site_temperature,
                                                     reinterpret_cast<int *>(site_element), site_charge, u.data(), (int)u.size(), resume,
                                                     &n_events, &exhausted, nullptr, &event_time));
        for (int i = 0; i < 2 * n_events; ++i) (void)rng.getRandomNumber();
        if (!exhausted) break;
        resume = 1;
    }
    return event_time;
}

DKMC_SHIM_API void update_power_gpu_sparse(dkmc_handle_t, dkmc_handle_t, GPUBuffers &gpubuf, const int num_source_inj, const int num_ground_ext,
                                    const int num_layers_contact, const double Vd, const int pbc, const double high_G, const double low_G,
                                    const double loop_G, const double G0, const double tol, const double nn_dist, const double m_e,
                                    const double V0, int num_metals, double *imacro, const bool solve_heating_local,
                                    const bool solve_heating_global, const double alpha_disp)
{
    GPUBuffers::report(dkmc_update_power_gpu_sparse(&gpubuf, num_source_inj, num_ground_ext, num_layers_contact, Vd, pbc, high_G, low_G, loop_G,
                                                    G0, tol, nn_dist, m_e, V0, num_metals, imacro, solve_heating_local, solve_heating_global, alpha_disp));
}

// update_power_gpu_split (gpu_solvers.h:167-172 of the reference; current_solver_gpu.cu:475-776): "mixed sparse neighbor matrix + dense
// tunneling submatrix".  In the reference the call is commented out (current_solver.cpp:32-35) and its solver ends in exit(1)
// (iterative_solvers_gpu.cu:656-821).  Here the split form IS the default layout of X -- neighbour part as a small CSR, tunnelling block
// as symmetric tiles (csrc/xt.hip) -- so the entry point is provided with the reference's argument list: it forces that layout for the
// call and restores the caller's dkmc_set_x_format afterwards.  Same results as update_power_gpu_sparse.
DKMC_SHIM_API void update_power_gpu_split(dkmc_handle_t, dkmc_handle_t, GPUBuffers &gpubuf, const int num_source_inj, const int num_ground_ext,
                                   const int num_layers_contact, const double Vd, const int pbc, const double high_G, const double low_G,
                                   const double loop_G, const double G0, const double tol, const double nn_dist, const double m_e,
                                   const double V0, int num_metals, double *imacro, const bool solve_heating_local,
                                   const bool solve_heating_global, const double alpha_disp)
{
    const int fmt = dkmc_get_x_format();
    dkmc_set_x_format(1);
    GPUBuffers::report(dkmc_update_power_gpu_sparse(&gpubuf, num_source_inj, num_ground_ext, num_layers_contact, Vd, pbc, high_G, low_G, loop_G,
                                                    G0, tol, nn_dist, m_e, V0, num_metals, imacro, solve_heating_local, solve_heating_global, alpha_disp));
    dkmc_set_x_format(fmt);
}

DKMC_SHIM_API void update_temperatureglobal_gpu(const double *site_power, double *T_bg, const int N, const double a_coeff, const double b_coeff,
                                         const double number_steps, const double C_thermal, const double small_step)
{
    GPUBuffers::report(dkmc_update_temperatureglobal_gpu(site_power, T_bg, N, a_coeff, b_coeff, number_steps, C_thermal, small_step));
}

// ---- additions without a counterpart in the reference's gpu_solvers.h ------------------------------------------------------
// Local temperature model: the reference keeps it on the host with dense inverses (heat_solver.cpp:40-246, 286-308, 354-513).
// Device::constructLaplacian would call the first (N_left_tot / N_right_tot from get_num_in_contacts, gamma from :86), the
// local branch of Device::updateTemperature the second; both work on the GPUBuffers arrays.
DKMC_SHIM_API void construct_laplacian_gpu(GPUBuffers &gpubuf, const int N_left_tot, const int N_right_tot, const double gamma)
{
    GPUBuffers::report(dkmc_construct_laplacian(&gpubuf, N_left_tot, N_right_tot, gamma));
}

DKMC_SHIM_API double update_temperature_local_gpu(GPUBuffers &gpubuf, const double step_time, const double delta_t, const double tau,
                                           const double background_temp, const double k_th_interface, const double k_th_vacancies,
                                           const double nn_dist, const int num_atoms_contact)
{
    double T_bg = background_temp;
    GPUBuffers::report(dkmc_update_temperature_local(&gpubuf, step_time, delta_t, tau, background_temp, k_th_interface, k_th_vacancies,
                                                     nn_dist, num_atoms_contact, nullptr, nullptr, nullptr, &T_bg));
    return T_bg;
}
