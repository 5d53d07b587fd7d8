// Micro-benchmark for the symmetric-tile kernel shape: one wave per R x C tile of a dense block stored row-major with a row pitch,
// producing R row sums and C column sums.  Build: hipcc --offload-arch=gfx950 -O3 -o bench_tiles tools/bench_tiles.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef double dbl2 __attribute__((ext_vector_type(2)));

// streaming baseline: one wave per 2048-entry contiguous segment, 16-byte loads (the production segment kernel's inner loop)
__global__ __launch_bounds__(256) void k_stream(const double *__restrict__ a, const double *__restrict__ p, long long nseg, double *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const long long seg = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (seg >= nseg) return;
    const dbl2 *av = reinterpret_cast<const dbl2 *>(a + seg * 2048);
    const double *pv = p + (seg * 2048) % 4096;
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    for (int k = lane; k < 1024; k += 256) {
        const dbl2 a0 = av[k], a1 = av[k + 64], a2 = av[k + 128], a3 = av[k + 192];
        s0 += a0.x * pv[2 * k] + a0.y * pv[2 * k + 1];
        s1 += a1.x * pv[2 * (k + 64)] + a1.y * pv[2 * (k + 64) + 1];
        s2 += a2.x * pv[2 * (k + 128)] + a2.y * pv[2 * (k + 128) + 1];
        s3 += a3.x * pv[2 * (k + 192)] + a3.y * pv[2 * (k + 192) + 1];
    }
    double s = (s0 + s1) + (s2 + s3);
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) out[seg] = s;
}

// R x C tile per wave; lanes own C/64 columns each (pairs); strips in phases of PH rows; V = 1: 8-byte loads, V = 2: 16-byte loads
template <int R, int C, int PH, int V>
__global__ __launch_bounds__(256) void k_tile(const double *__restrict__ a, long long pitch, int off, const double *__restrict__ pS, int ntw, long long ntiles,
                                              double *__restrict__ rowpart, double *__restrict__ colpart)
{
    const int lane = threadIdx.x & 63;
    const long long tile = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (tile >= ntiles) return;
    const int k = (int)(tile / ntw), w = (int)(tile % ntw);
    constexpr int NP = C / 128;                   // column pairs per lane
    const double *pc = pS + (size_t)w * C, *pr = pS + 100000 + (size_t)k * R;
    double pcx[NP], pcy[NP], cax[NP], cay[NP];
#pragma unroll
    for (int u = 0; u < NP; ++u) { const int col = 2 * lane + 128 * u; pcx[u] = pc[col]; pcy[u] = pc[col + 1]; cax[u] = 0; cay[u] = 0; }
    double rs[R / PH];
#pragma unroll
    for (int i = 0; i < R / PH; ++i) rs[i] = 0;
#pragma unroll 1
    for (int ph = 0; ph < R / PH; ++ph) {
        double x[PH][NP], y[PH][NP], ra[PH];
#pragma unroll
        for (int q = 0; q < PH; ++q) {
            const double *strip = a + (size_t)(k * R + ph * PH + q) * pitch + off + (size_t)w * C;
#pragma unroll
            for (int u = 0; u < NP; ++u) {
                const int col = 2 * lane + 128 * u;
                if (V == 2) { const dbl2 v = *reinterpret_cast<const dbl2 *>(strip + col); x[q][u] = v.x; y[q][u] = v.y; }
                else { x[q][u] = strip[col]; y[q][u] = strip[col + 1]; }
            }
        }
#pragma unroll
        for (int q = 0; q < PH; ++q) {
            const double prow = pr[ph * PH + q];
            double r_ = 0;
#pragma unroll
            for (int u = 0; u < NP; ++u) { r_ += x[q][u] * pcx[u] + y[q][u] * pcy[u]; cax[u] += x[q][u] * prow; cay[u] += y[q][u] * prow; }
            ra[q] = r_;
        }
        // butterfly over the PH strips (PH = 8: xor 32, 16, 8)
#pragma unroll
        for (int half = PH / 2, bit = 32; half >= 1; half >>= 1, bit >>= 1) {
            const bool up = (lane & bit) != 0;
#pragma unroll
            for (int j = 0; j < half; ++j) { const double keep = up ? ra[j + half] : ra[j], send = up ? ra[j] : ra[j + half]; ra[j] = keep + __shfl_xor(send, bit, 64); }
        }
        double v = ra[0];
        for (int b = 64 / PH / 2; b >= 1; b >>= 1) v += __shfl_xor(v, b, 64);      // finish (benchmark: plain reduction of the rest)
#pragma unroll
        for (int i = 0; i < R / PH; ++i) rs[i] = (i == ph) ? v : rs[i];
    }
    if (lane < R / PH) { double v = 0; for (int i = 0; i < R / PH; ++i) v = (lane == i) ? rs[i] : v; rowpart[tile * (R / PH) + lane] = v; }
    double *cp = colpart + (size_t)tile * C;
#pragma unroll
    for (int u = 0; u < NP; ++u) { const int col = 2 * lane + 128 * u; dbl2 v; v.x = cax[u]; v.y = cay[u]; *reinterpret_cast<dbl2 *>(cp + col) = v; }
}

template <int R, int C, int PH, int V>
static void run(const char *name, const double *a, long long pitch, int off, const double *pS, int rows, int cols, double *rowpart, double *colpart)
{
    const int ntw = cols / C; const long long ntiles = (long long)(rows / R) * ntw;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int blocks = (int)((ntiles + 3) / 4);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k_tile<R, C, PH, V>), dim3(blocks), dim3(256), 0, 0, a, pitch, off, pS, ntw, ntiles, rowpart, colpart);
    CHECK(hipEventRecord(e0));
    const int N = 50;
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL((k_tile<R, C, PH, V>), dim3(blocks), dim3(256), 0, 0, a, pitch, off, pS, ntw, ntiles, rowpart, colpart);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)ntiles * R * C * 8;
    printf("%-28s tiles %6lld  %7.2f us  %6.2f TB/s\n", name, ntiles, ms / N * 1e3, bytes / (ms / N * 1e-3) / 1e12);
}

int main(int argc, char **argv)
{
    const int rows = argc > 1 ? atoi(argv[1]) : 2048, cols = argc > 2 ? atoi(argv[2]) : 6144;
    const long long pitch = cols + 777;                 // rows are not aligned to anything
    double *a, *pS, *rowpart, *colpart, *out;
    CHECK(hipMalloc(&a, (size_t)rows * pitch * 8 + 4096)); CHECK(hipMalloc(&pS, 200000 * 8));
    CHECK(hipMalloc(&rowpart, (size_t)rows * cols / 64 * 8)); CHECK(hipMalloc(&colpart, (size_t)rows * cols / 8 * 8)); CHECK(hipMalloc(&out, (size_t)rows * pitch / 2048 * 8 + 64));
    CHECK(hipMemset(a, 0, (size_t)rows * pitch * 8 + 4096)); CHECK(hipMemset(pS, 0, 200000 * 8));
    printf("dense block %d x %d (%.1f MB), row pitch %lld doubles\n", rows, cols, rows * (double)cols * 8 / 1e6, pitch);
    {   // streaming baseline over the same number of bytes
        const long long nseg = (long long)rows * cols / 2048;
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k_stream, dim3((nseg + 3) / 4), dim3(256), 0, 0, a, pS, nseg, out);
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k_stream, dim3((nseg + 3) / 4), dim3(256), 0, 0, a, pS, nseg, out);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-28s segs  %6lld  %7.2f us  %6.2f TB/s\n", "stream 2048 x dbl2", nseg, ms / 50 * 1e3, (double)nseg * 2048 * 8 / (ms / 50 * 1e-3) / 1e12);
    }
    for (int off = 0; off < 2; ++off) {
        printf("-- strip offset %d (16-byte loads need an even offset + even pitch; here pitch is odd: V=2 only as an upper bound with off=0 rows misaligned -> skipped)\n", off);
        run<32, 256, 8, 1>("32x256 ph8 8B", a, pitch, off, pS, rows, cols, rowpart, colpart);
        run<16, 512, 8, 1>("16x512 ph8 8B", a, pitch, off, pS, rows, cols, rowpart, colpart);
        run<16, 512, 4, 1>("16x512 ph4 8B", a, pitch, off, pS, rows, cols, rowpart, colpart);
        run<8, 1024, 4, 1>("8x1024 ph4 8B", a, pitch, off, pS, rows, cols, rowpart, colpart);
        run<64, 128, 8, 1>("64x128 ph8 8B", a, pitch, off, pS, rows, cols, rowpart, colpart);
        run<32, 128, 8, 1>("32x128 ph8 8B", a, pitch, off, pS, rows, cols, rowpart, colpart);
        run<16, 256, 8, 1>("16x256 ph8 8B", a, pitch, off, pS, rows, cols, rowpart, colpart);
    }
    {   // aligned variant for 16-byte loads: even pitch
        const long long pitch2 = cols + 778;
        printf("-- even pitch, 16-byte loads\n");
        run<32, 256, 8, 2>("32x256 ph8 16B", a, pitch2, 0, pS, rows, cols, rowpart, colpart);
        run<16, 512, 8, 2>("16x512 ph8 16B", a, pitch2, 0, pS, rows, cols, rowpart, colpart);
        run<16, 512, 4, 2>("16x512 ph4 16B", a, pitch2, 0, pS, rows, cols, rowpart, colpart);
        run<32, 256, 8, 1>("32x256 ph8 8B (even pitch)", a, pitch2, 0, pS, rows, cols, rowpart, colpart);
    }
    return 0;
}
