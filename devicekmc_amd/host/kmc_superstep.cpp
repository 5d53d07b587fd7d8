// kmc_superstep.cpp -- C++ host driver over the drop-in shim (include/gpu_buffers.h, include/gpu_solvers.h).
//
// Mirrors what kmc_main.cpp:109-279 does for one bias point, with the reference's call sequence and argument lists:
// GPUBuffers ctor -> sync_HostToGPU -> initialize_sparsity -> update_CB_edge_gpu_sparse (setLaplacePotential) ->
// per step { update_charge_gpu, background_potential_gpu_sparse, poisson_gridless_gpu, execute_kmc_step_gpu,
// update_power_gpu_sparse, update_temperature (host formula on the device) }.
// Input / output are raw binary bundles written / read by tests/test_cpp_host.py (no reference parser here:
// input_parser / xyz reading are out of scope).
//
//   kmc_superstep <bundle.in> <bundle.out> <steps> [warmup]     (warmup given: `warmup` untimed steps first, then a TIMING line on stdout)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "gpu_solvers.h"

struct HostDevice {             // the public vectors of the reference's Device (Device.h:66-106) that the GPU path touches
    int N = 0, N_atom = 0, max_num_neighbors = 0;
    double T_bg = 300.0, imacro = 0.0;
    std::vector<ELEMENT> site_element;
    std::vector<int> site_charge, neigh_idx, site_layer;
    std::vector<double> site_x, site_y, site_z, site_power, site_CB_edge, site_potential_boundary, site_potential_charge,
        site_temperature, atom_CB_edge;
};

template <class T> static void rd(FILE *f, T *p, size_t n) { if (fread(p, sizeof(T), n, f) != n) { fprintf(stderr, "short read\n"); exit(2); } }
template <class T> static void wr(FILE *f, const T *p, size_t n) { if (fwrite(p, sizeof(T), n, f) != n) { fprintf(stderr, "short write\n"); exit(2); } }

int main(int argc, char **argv)
{
    if (argc < 4) { fprintf(stderr, "usage: %s bundle.in bundle.out steps\n", argv[0]); return 1; }
    const int steps = atoi(argv[3]);
    const int warmup = argc > 4 ? atoi(argv[4]) : 0;
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    int hdr[8]; rd(f, hdr, 8);
    const int N = hdr[0], nn = hdr[1], N_atom = hdr[2], n_layers = hdr[3], n_metals = hdr[4], n_first = hdr[5], n_lay_contact = hdr[6], pbc = hdr[7];
    double par[16]; rd(f, par, 16);
    const double Vd = par[0], freq = par[1], sigma = par[2], k = par[3], nn_dist = par[4], high_G = par[5], low_G = par[6], m_e = par[7], V0 = par[8];
    const double T_bg0 = par[9], diss = par[10], t_ox = par[11], A = par[12], c_p = par[13];
    const unsigned seed_kmc = (unsigned)par[14];
    std::vector<double> lattice(3); rd(f, lattice.data(), 3);
    std::vector<Layer> layers(n_layers);
    for (auto &l : layers) { double e[4]; rd(f, e, 4); l.E_gen_0 = e[0]; l.E_rec_1 = e[1]; l.E_diff_2 = e[2]; l.E_diff_3 = e[3]; }
    std::vector<int> metals_i(n_metals); rd(f, metals_i.data(), n_metals);
    std::vector<ELEMENT> metals; for (int m : metals_i) metals.push_back((ELEMENT)m);
    HostDevice dev; dev.N = N; dev.N_atom = N_atom; dev.max_num_neighbors = nn; dev.T_bg = T_bg0;
    std::vector<int> el(N); rd(f, el.data(), N);
    dev.site_element.resize(N); for (int i = 0; i < N; ++i) dev.site_element[i] = (ELEMENT)el[i];
    dev.site_layer.resize(N); rd(f, dev.site_layer.data(), N);
    dev.site_x.resize(N); dev.site_y.resize(N); dev.site_z.resize(N);
    rd(f, dev.site_x.data(), N); rd(f, dev.site_y.data(), N); rd(f, dev.site_z.data(), N);
    dev.neigh_idx.resize((size_t)N * nn); rd(f, dev.neigh_idx.data(), (size_t)N * nn);
    fclose(f);
    dev.site_charge.assign(N, 0); dev.site_power.assign(N, 0.0); dev.site_CB_edge.assign(N, 0.0);
    dev.site_potential_boundary.assign(N, 0.0); dev.site_potential_charge.assign(N, 0.0); dev.site_temperature.assign(N, T_bg0);

    char name[1000]; get_gpu_info(name, 0); set_gpu(0);
    fprintf(stderr, "Will use this GPU: %s\n", name);
    RandomNumberGenerator rng; rng.setSeed(seed_kmc);                                    // KMCProcess.cpp:21
    GPUBuffers gpubuf(layers, dev.site_layer, freq, N, N_atom, dev.site_x, dev.site_y, dev.site_z, nn, sigma, k, lattice,
                      dev.neigh_idx, metals, (int)metals.size());                        // kmc_main.cpp:116-119
    gpubuf.sync_HostToGPU(dev);
    initialize_sparsity(gpubuf, pbc, nn_dist, n_first);                                  // kmc_main.cpp:121
    dkmc_handle_t h = nullptr;
    update_CB_edge_gpu_sparse(h, h, gpubuf, N, n_first, n_first, Vd, pbc, high_G, low_G, nn_dist, (int)metals.size());   // potential_solver.cpp:15
    gpubuf.sync_GPUToHost(dev); gpubuf.sync_HostToGPU(dev);

    std::vector<double> out_dt, out_I, out_T;
    // current_solver.cpp:8-17
    const double X_loop_G = high_G * 10000000, X_high_G = high_G * 100000, X_low_G = low_G, G0 = 2 * 3.8612e-5 * 1e-5, tol = 1.60217663e-19 * 0.01;
    std::chrono::steady_clock::time_point t_start;
    for (int step = -warmup; step < steps; ++step) {
        if (step == 0 && argc > 4) { dkmc_synchronize(); t_start = std::chrono::steady_clock::now(); }
        update_charge_gpu(gpubuf.site_element, gpubuf.site_charge, gpubuf.neigh_idx, gpubuf.N_, gpubuf.nn_,
                          gpubuf.metal_types, gpubuf.num_metal_types_);                   // potential_solver.cpp:152, verbatim
        background_potential_gpu_sparse(h, h, gpubuf, N, n_first, n_first, Vd, pbc, high_G, low_G, nn_dist, (int)metals.size(), step + warmup);
        poisson_gridless_gpu(n_first, pbc, gpubuf.N_, gpubuf.lattice, gpubuf.sigma, gpubuf.k, gpubuf.site_x, gpubuf.site_y, gpubuf.site_z,
                             gpubuf.site_charge, gpubuf.site_potential_charge);
        const double dt = execute_kmc_step_gpu(N, nn, gpubuf.neigh_idx, gpubuf.site_layer, gpubuf.lattice, pbc, gpubuf.T_bg, gpubuf.freq,
                                               gpubuf.sigma, gpubuf.k, gpubuf.site_x, gpubuf.site_y, gpubuf.site_z,
                                               gpubuf.site_potential_boundary, gpubuf.site_potential_charge, gpubuf.site_temperature,
                                               gpubuf.site_element, gpubuf.site_charge, rng, dev.neigh_idx.data());   // KMCProcess.cpp:269-274, verbatim
        update_power_gpu_sparse(h, h, gpubuf, n_first, n_first, n_lay_contact, Vd, pbc, X_high_G, X_low_G, X_loop_G, G0, tol, nn_dist, m_e, V0,
                                (int)metals.size(), &dev.imacro, false, true, 1.0);
        double P = 0.0;                                                                   // heat_solver.cpp:316-350, on the device
        GPUBuffers::report(dkmc_update_temperature_global_analytic(gpubuf.site_power, gpubuf.T_bg, N, dt, diss, t_ox, A, c_p, &P));
        if (step < 0) continue;
        if (argc > 4) { out_dt.push_back(dt); out_I.push_back(dev.imacro); out_T.push_back(0.0); continue; }      // timed run: no host copies inside the loop
        out_dt.push_back(dt); out_I.push_back(dev.imacro);
        gpubuf.sync_GPUToHost(dev); out_T.push_back(dev.T_bg);
        fprintf(stderr, "step %d: KMC step time %.6e  Current [uA] %.6f  T_bg %.6f\n", step, dt, dev.imacro * 1e6, dev.T_bg);
    }
    if (argc > 4) {
        dkmc_synchronize();
        const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
        printf("TIMING steps=%d seconds=%.6f\n", steps, sec);
        gpubuf.sync_GPUToHost(dev);
    }
    // the split entry point of the reference's header (gpu_solvers.h:167-172; its call is commented out at current_solver.cpp:32-35):
    // same state, once through each name
    double I_sparse = 0.0, I_split = 0.0;
    if (argc <= 4) {
        update_power_gpu_sparse(h, h, gpubuf, n_first, n_first, n_lay_contact, Vd, pbc, X_high_G, X_low_G, X_loop_G, G0, tol, nn_dist, m_e, V0,
                                (int)metals.size(), &I_sparse, false, false, 1.0);
        update_power_gpu_split(h, h, gpubuf, n_first, n_first, n_lay_contact, Vd, pbc, X_high_G, X_low_G, X_loop_G, G0, tol, nn_dist, m_e, V0,
                               (int)metals.size(), &I_split, false, false, 1.0);
    }
    const double next_u = rng.getRandomNumber();       // proves the stream position
    f = fopen(argv[2], "wb");
    if (!f) { perror(argv[2]); return 1; }
    wr(f, out_dt.data(), steps); wr(f, out_I.data(), steps); wr(f, out_T.data(), steps); wr(f, &next_u, 1);
    for (int i = 0; i < N; ++i) el[i] = (int)dev.site_element[i];
    wr(f, el.data(), N); wr(f, dev.site_charge.data(), N);
    wr(f, dev.site_potential_boundary.data(), N); wr(f, dev.site_potential_charge.data(), N);
    wr(f, &I_sparse, 1); wr(f, &I_split, 1);
    fclose(f);
    gpubuf.freeGPUmemory();
    return 0;
}
