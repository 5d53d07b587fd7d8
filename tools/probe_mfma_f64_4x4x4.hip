// Lane maps of v_mfma_f64_4x4x4_4b_f64 on gfx950, found from the hardware: A = e_la, B = e_lb (unit vectors over the lanes); the product has
// one non-zero iff la and lb sit in the same block with the same k; the lane holding it gives (block, i, j).
// Build: hipcc --offload-arch=gfx950 -O3 tools/probe_mfma_f64_4x4x4.hip -o tools/probe_mfma_f64_4x4x4
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int *out)
{
    const int la = blockIdx.x, lb = blockIdx.y, l = threadIdx.x;
    const double a = l == la ? 1.0 : 0.0, b = l == lb ? 1.0 : 0.0;
    const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
    if (d != 0.0) out[la * 64 + lb] = l;
}
int main()
{
    int *d, h[4096];
    hipMalloc(&d, sizeof(h)); hipMemset(d, 0xff, sizeof(h));
    hipLaunchKernelGGL(k, dim3(64, 64), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int la = 0; la < 64; ++la) {
        printf("A lane %2d:", la);
        for (int lb = 0; lb < 64; ++lb) if (h[la * 64 + lb] >= 0) printf("  B%2d->D%2d", lb, h[la * 64 + lb]);
        printf("\n");
    }
    return 0;
}
