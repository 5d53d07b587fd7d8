/*
 * gpu_buffers.h -- drop-in for the reference's GPUBuffers (gpu_buffers.h:12-162, gpu_buffers.cpp:10-118) on top of
 * the C ABI in devicekmc_hip.h.  Same public member names, same constructor argument order, same methods; memory comes
 * from hipMalloc inside the engine.  The class derives from the C struct so `&gpubuf` is what the C ABI takes.
 *
 * `DeviceT` is any type with the reference Device's public vectors (Device.h:66-106): site_element, site_charge,
 * site_power, site_CB_edge, site_potential_boundary, site_potential_charge, site_temperature, atom_CB_edge, T_bg.
 */
#pragma once
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "devicekmc_types.h"            // ELEMENT (or, with DKMC_HAVE_REFERENCE_TYPES, the reference's own utils.h has defined it)
#define DKMC_ELEMENT_T ELEMENT          // gpubuf.site_element / atom_element / metal_types are ELEMENT *, as in the reference
#include "devicekmc_hip.h"

class GPUBuffers : public dkmc_gpubuf {
public:
    std::vector<double> E_gen_host, E_rec_host, E_Vdiff_host, E_Odiff_host;

    GPUBuffers() : dkmc_gpubuf() {}                                        // CPU-only code path of the reference

    GPUBuffers(std::vector<Layer> layers, std::vector<int> site_layer_in, double freq_in, int N, int N_atom,
               std::vector<double> site_x_in, std::vector<double> site_y_in, std::vector<double> site_z_in,
               int nn, double sigma_in, double k_in, std::vector<double> lattice_in, std::vector<int> neigh_idx_in,
               std::vector<ELEMENT> metals, int num_metals_types) : dkmc_gpubuf()
    {
        for (auto &l : layers) {
            E_gen_host.push_back(l.E_gen_0); E_rec_host.push_back(l.E_rec_1);
            E_Vdiff_host.push_back(l.E_diff_2); E_Odiff_host.push_back(l.E_diff_3);
        }
        report(dkmc_copy_to_const_memory(E_gen_host.data(), E_rec_host.data(), E_Vdiff_host.data(), E_Odiff_host.data(),
                                         (int)layers.size()));
        static_assert(sizeof(ELEMENT) == sizeof(int), "ELEMENT must be a 4-byte enum");
        report(dkmc_gpubuf_create(this, N, N_atom, nn, num_metals_types, site_layer_in.data(), site_x_in.data(),
                                  site_y_in.data(), site_z_in.data(), neigh_idx_in.data(),
                                  reinterpret_cast<const int *>(metals.data()), freq_in, sigma_in, k_in, lattice_in.data()));
    }

    template <class DeviceT> void sync_HostToGPU(DeviceT &device)
    {
        if ((size_t)N_ != device.site_element.size()) { fprintf(stderr, "ERROR: Size mismatch in GPU memory copy.\n"); exit(EXIT_FAILURE); }
        report(dkmc_gpubuf_sync_host_to_gpu(this, reinterpret_cast<const int *>(device.site_element.data()), device.site_charge.data(),
                                            device.site_power.data(), device.site_CB_edge.data(), device.site_potential_boundary.data(),
                                            device.site_potential_charge.data(), device.site_temperature.data(),
                                            device.atom_CB_edge.size() >= (size_t)N_atom_ ? device.atom_CB_edge.data() : nullptr, device.T_bg));
    }
    template <class DeviceT> void sync_GPUToHost(DeviceT &device)
    {
        report(dkmc_gpubuf_sync_gpu_to_host(this, reinterpret_cast<int *>(device.site_element.data()), device.site_charge.data(),
                                            device.site_power.data(), device.site_CB_edge.data(), device.site_potential_boundary.data(),
                                            device.site_potential_charge.data(), device.site_temperature.data(),
                                            device.atom_CB_edge.size() >= (size_t)N_atom_ ? device.atom_CB_edge.data() : nullptr, &device.T_bg));
    }
    void copy_power_fromGPU(std::vector<double> &power) { power.resize(N_); report(dkmc_copy_power_from_gpu(this, power.data())); }
    void copy_charge_toGPU(std::vector<int> &charge) { charge.resize(N_); report(dkmc_copy_charge_to_gpu(this, charge.data())); }
    void copy_Tbg_toGPU(double new_T_bg) { report(dkmc_copy_Tbg_to_gpu(this, new_T_bg)); }
    void freeGPUmemory() { dkmc_gpubuf_free(this); }

    // error behaviour of the reference: print and carry on (utils.h:145-153)
    static void report(int rc) { if (rc) { fprintf(stderr, "GPUassert: %s\n", dkmc_last_error()); dkmc_clear_error(); } }
};
