/*
 * devicekmc_types.h -- the host-side types the drop-in shim needs when it is used WITHOUT the reference's own
 * utils.h / random_num.h (stand-alone builds, tests).  When compiling inside the reference tree, define
 * DKMC_HAVE_REFERENCE_TYPES before including gpu_buffers.h / gpu_solvers.h and the reference's own definitions
 * (utils.h:37-72, random_num.h:4-23) are used instead; the enum values and layouts below are identical to them.
 */
#pragma once
#ifndef DKMC_HAVE_REFERENCE_TYPES
#include <random>
#include <string>

enum ELEMENT { DEFECT, OXYGEN_DEFECT, VACANCY, O_EL, Hf_EL, Ni_EL, Ti_EL, Pt_EL, N_EL, NULL_ELEMENT };
enum EVENTTYPE { VACANCY_GENERATION, VACANCY_RECOMBINATION, VACANCY_DIFFUSION, ION_DIFFUSION, NULL_EVENT };

struct Layer {
    std::string type;
    double E_gen_0 = 0, E_rec_1 = 0, E_diff_2 = 0, E_diff_3 = 0;
    double start_x = 0, end_x = 0;
    double init_vac_percentage = 0;
};

class RandomNumberGenerator {
public:
    RandomNumberGenerator() : rng(0) {}
    void setSeed(unsigned int seed) { rng.seed(seed); }
    double getRandomNumber() { std::uniform_real_distribution<double> d(0.0, 1.0); return d(rng); }
private:
    std::mt19937 rng;
};
#endif

/* Opaque stand-ins for the two library handles the reference threads through its solver calls
 * (cublasHandle_t, cusolverDnHandle_t: kmc_main.cpp:127-128).  The HIP engine needs neither. */
struct dkmc_handle_s;
typedef dkmc_handle_s *dkmc_handle_t;
