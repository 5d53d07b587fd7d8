// Issue rate of the fp64 matrix instructions on gfx950 (one wave per SIMD, independent accumulators): cycles per instruction per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/bench_mfma_f64.hip -o tools/bench_mfma_f64
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double dbl4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k16(int n, double a, double b, double *out, long long *cyc)
{
    dbl4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (dbl4)(0.0);
    const double av = a + threadIdx.x, bv = b - threadIdx.x;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < n; ++it)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[i], 0, 0, 0);
    const long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NACC>
__global__ __launch_bounds__(256) void k4(int n, double a, double b, double *out, long long *cyc)
{
    double acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
    const double av = a + threadIdx.x, bv = b - threadIdx.x;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < n; ++it)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, acc[i], 0, 0, 0);
    const long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// the same instruction stream on pseudo-random operands that change from instruction to instruction (16 A and 16 B registers per lane):
// what a kernel on real data sees of the matrix pipe (operand fetch from changing registers, data-dependent power)
__global__ __launch_bounds__(256) void k4r(int n, const double *__restrict__ src, double *out)
{
    double a[16], b[16], acc[8];
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = src[(threadIdx.x * 16 + i) & 4095]; b[i] = src[(threadIdx.x * 16 + i + 1777) & 4095]; }
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.0;
    for (int it = 0; it < n; ++it)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i & 7] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i], b[(i * 5 + 3) & 15], acc[i & 7], 0, 0, 0);
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main()
{
    double *out; long long *cyc;
    const int nb = 256, n = 20000;
    hipMalloc(&out, nb * 256 * 8); hipMalloc(&cyc, nb * 8);
    long long h[256];
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0); hipLaunchKernelGGL(k16<8>, dim3(nb), dim3(256), 0, 0, n, 1.0, 2.0, out, cyc); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); hipMemcpy(h, cyc, nb * 8, hipMemcpyDeviceToHost);
        printf("v_mfma_f64_16x16x4_f64  8 acc: %.1f cycles/instr/SIMD, %.2f TFLOP/s chip-wide (%d CUs x 4 waves)\n", (double)h[0] / (n * 8.0), nb * 4.0 * n * 8 * 2048.0 / (ms * 1e-3) / 1e12, nb);
        hipEventRecord(e0); hipLaunchKernelGGL(k16<1>, dim3(nb), dim3(256), 0, 0, n, 1.0, 2.0, out, cyc); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); hipMemcpy(h, cyc, nb * 8, hipMemcpyDeviceToHost);
        printf("v_mfma_f64_16x16x4_f64  1 acc (dependent chain): %.1f cycles/instr\n", (double)h[0] / (n * 1.0));
        hipEventRecord(e0); hipLaunchKernelGGL(k4<8>, dim3(nb), dim3(256), 0, 0, n, 1.0, 2.0, out, cyc); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); hipMemcpy(h, cyc, nb * 8, hipMemcpyDeviceToHost);
        printf("v_mfma_f64_4x4x4_4b_f64 8 acc: %.1f cycles/instr/SIMD, %.2f TFLOP/s chip-wide\n", (double)h[0] / (n * 8.0), nb * 4.0 * n * 8 * 512.0 / (ms * 1e-3) / 1e12);
    }
    // sustained rate: ~3 s of back-to-back launches on non-trivial operands (the chip lowers its clock under a matrix-dense load; a
    // millisecond burst does not show it)
    for (int which = 0; which < 2; ++which) {
        const int launches = 400;
        hipEventRecord(e0);
        for (int r = 0; r < launches; ++r) {
            if (which == 0) hipLaunchKernelGGL(k4<8>, dim3(nb * 4), dim3(256), 0, 0, n, 1.37 + r, 0.731, out, cyc);
            else hipLaunchKernelGGL(k16<8>, dim3(nb * 4), dim3(256), 0, 0, n / 4, 1.37 + r, 0.731, out, cyc);
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flops = which == 0 ? (double)launches * nb * 4 * 4.0 * n * 8 * 512.0 : (double)launches * nb * 4 * 4.0 * (n / 4) * 8 * 2048.0;
        printf("sustained (%.1f s, 4 waves per SIMD): %s %.2f TFLOP/s\n", ms * 1e-3, which == 0 ? "v_mfma_f64_4x4x4_4b_f64" : "v_mfma_f64_16x16x4_f64 ", flops / (ms * 1e-3) / 1e12);
    }
    {
        double hsrc[4096]; unsigned long long x = 88172645463325252ull;
        for (int i = 0; i < 4096; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; hsrc[i] = (double)(x >> 11) / 9007199254740992.0 * 2e-3 - 1e-3; }
        double *dsrc; hipMalloc(&dsrc, sizeof(hsrc)); hipMemcpy(dsrc, hsrc, sizeof(hsrc), hipMemcpyHostToDevice);
        for (int waves = 1; waves <= 4; waves *= 4) {
            const int launches = 300, nn = n / 2;
            hipEventRecord(e0);
            for (int r = 0; r < launches; ++r) hipLaunchKernelGGL(k4r, dim3(nb * waves), dim3(256), 0, 0, nn, dsrc, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("sustained on pseudo-random operands (%.1f s, %d wave(s) per SIMD): v_mfma_f64_4x4x4_4b_f64 %.2f TFLOP/s\n", ms * 1e-3, waves,
                   (double)launches * nb * waves * 4.0 * nn * 16 * 512.0 / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
