// Probe: the DPP forms of "lane i <- lane i ^ OFF" (OFF = 1, 2, 4, 8) used by devicekmc_amd/csrc/common.h (xor_lane) against __shfl_xor on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probe_dpp_xor.hip -o tools/probe_dpp_xor
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int OFF> __device__ __forceinline__ int xor_lane_i32(int x)
{
    if constexpr (OFF == 1) return __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, false);        // quad_perm [1,0,3,2]
    else if constexpr (OFF == 2) return __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
    else if constexpr (OFF == 4) { const int t = __builtin_amdgcn_update_dpp(0, x, 0x104, 0xf, 0x5, false); return __builtin_amdgcn_update_dpp(t, x, 0x114, 0xf, 0xa, false); }   // row_shl:4 into banks 0, 2; row_shr:4 into banks 1, 3
    else return __builtin_amdgcn_update_dpp(0, x, 0x128, 0xf, 0xf, false);                           // row_ror:8
}
__global__ void k(int *out)
{
    const int l = threadIdx.x, x = 1000 + 7 * l;
    out[0 * 64 + l] = xor_lane_i32<1>(x) - __shfl_xor(x, 1, 64);
    out[1 * 64 + l] = xor_lane_i32<2>(x) - __shfl_xor(x, 2, 64);
    out[2 * 64 + l] = xor_lane_i32<4>(x) - __shfl_xor(x, 4, 64);
    out[3 * 64 + l] = xor_lane_i32<8>(x) - __shfl_xor(x, 8, 64);
}
int main()
{
    int *d, h[256];
    if (hipMalloc(&d, sizeof(h)) != hipSuccess) return 2;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 2;
    int bad = 0;
    for (int i = 0; i < 256; ++i) bad += h[i] != 0;
    printf("dpp xor lanes: %s (%d mismatches of 256)\n", bad ? "MISMATCH" : "identical to __shfl_xor for offsets 1, 2, 4, 8", bad);
    return bad != 0;
}
