// shim_exports.cpp -- libdevicekmc_shim.so: the reference's solver entry points (gpu_solvers.h:36-208) as EXPORTED unmangled symbols
// on top of the C ABI of libdevicekmc_hip.so.  Same bodies as the header-only shim (include/gpu_solvers.h), compiled once with
// external C linkage; plain g++, no HIP headers.
#define DKMC_SHIM_API extern "C" __attribute__((visibility("default")))
#include "gpu_solvers.h"
