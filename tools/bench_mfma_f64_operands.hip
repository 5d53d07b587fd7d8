// Issue rate of v_mfma_f64_4x4x4_4b_f64 on gfx950 by WHERE its operands live (one wave per SIMD, 16 independent accumulators, explicit registers):
// the tile x panel kernel's matrix instructions take 22.4 cycles each with nothing else in the wave (k_xtb_apply variant 7) against 18 in
// tools/bench_mfma_f64.hip -- which operand placement costs the difference?
// Build: hipcc --offload-arch=gfx950 -O3 tools/bench_mfma_f64_operands.hip -o tools/bench_mfma_f64_operands
#include <hip/hip_runtime.h>
#include <stdio.h>
#define M(d, a, b) "v_mfma_f64_4x4x4_4b_f64 " d ", " a ", " b ", " d "\n"
// 16 instructions on 16 accumulator pairs; A operands cycle over 4 register pairs [64:71], B over 8 pairs [80:95] of their files.
// Accumulators: a[0:31], or v[96:127] for the all-VGPR form (v0.. hold the kernel's own values: never written here)
#define BLOCK16_(D, O0, O1, O2, O3, O4, O5, O6, O7, O8, O9, O10, O11, O12, O13, O14, O15, A, B) \
    M(D O0, A "[64:65]", B "[80:81]") M(D O1, A "[66:67]", B "[80:81]") M(D O2, A "[64:65]", B "[82:83]") M(D O3, A "[66:67]", B "[82:83]") \
    M(D O4, A "[64:65]", B "[84:85]") M(D O5, A "[66:67]", B "[84:85]") M(D O6, A "[64:65]", B "[86:87]") M(D O7, A "[66:67]", B "[86:87]") \
    M(D O8, A "[68:69]", B "[88:89]") M(D O9, A "[70:71]", B "[88:89]") M(D O10, A "[68:69]", B "[90:91]") M(D O11, A "[70:71]", B "[90:91]") \
    M(D O12, A "[68:69]", B "[92:93]") M(D O13, A "[70:71]", B "[92:93]") M(D O14, A "[68:69]", B "[94:95]") M(D O15, A "[70:71]", B "[94:95]")
#define BLOCK16A(A, B) BLOCK16_("a", "[0:1]", "[2:3]", "[4:5]", "[6:7]", "[8:9]", "[10:11]", "[12:13]", "[14:15]", "[16:17]", "[18:19]", "[20:21]", "[22:23]", "[24:25]", "[26:27]", "[28:29]", "[30:31]", A, B)
#define BLOCK16V(A, B) BLOCK16_("v", "[96:97]", "[98:99]", "[100:101]", "[102:103]", "[104:105]", "[106:107]", "[108:109]", "[110:111]", "[112:113]", "[114:115]", "[116:117]", "[118:119]", "[120:121]", "[122:123]", "[124:125]", "[126:127]", A, B)
#define KERNEL(name, BLK)                                                                                                       \
    __global__ __launch_bounds__(256) void name(int n, long long *cyc)                                                          \
    {                                                                                                                           \
        const long long t0 = __builtin_amdgcn_s_memtime();                                                                      \
        for (int it = 0; it < n; ++it) asm volatile(BLK BLK BLK BLK ::: "memory", "v64", "v127", "a95");                        \
        const long long t1 = __builtin_amdgcn_s_memtime();                                                                      \
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;                                       \
    }
// the asm writes a[0:31] or v[96:127] only and reads v/a[64:95]: the kernel's own values (low VGPRs, SGPRs) are untouched; the clobber list sizes the register files
KERNEL(k_avv, BLOCK16A("v", "v"))
KERNEL(k_ava, BLOCK16A("v", "a"))
KERNEL(k_aaa, BLOCK16A("a", "a"))
KERNEL(k_vvv, BLOCK16V("v", "v"))
KERNEL(k_aav, BLOCK16A("a", "v"))
int main()
{
    long long *cyc; const int nb = 256, n = 4000;
    hipMalloc(&cyc, nb * 4 * 8);
    long long h[1024];
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct { const char *what; void (*k)(int, long long *); } ks[] = {
        {"D/C in AGPR, A in VGPR, B in VGPR", k_avv}, {"D/C in AGPR, A in VGPR, B in AGPR", k_ava}, {"D/C, A, B in AGPR", k_aaa},
        {"D/C, A, B in VGPR", k_vvv}, {"D/C in AGPR, A in AGPR, B in VGPR", k_aav}};
    for (int rep = 0; rep < 2; ++rep)
        for (auto &k : ks) {
            hipEventRecord(e0); hipLaunchKernelGGL(k.k, dim3(nb), dim3(256), 0, 0, n, cyc); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); hipMemcpy(h, cyc, nb * 4 * 8, hipMemcpyDeviceToHost);
            printf("%-40s %.2f ns per instruction per SIMD (event), s_memtime ticks per instruction %.3f\n", k.what, ms * 1e6 / (n * 64.0), (double)h[0] / (n * 64.0));
        }
    return 0;
}
