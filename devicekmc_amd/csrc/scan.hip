// scan.hip -- deterministic exclusive prefix sum of int32 (replaces cub::DeviceScan::InclusiveSum,
// iterative_solvers_gpu.cu:988-995,1876-1884,1950-1952, and the thrust::copy_if compactions of
// current_solver_gpu.cu:869-879).  Three small launches: per-tile scan, scan of tile sums, fix-up.
#include "common.h"

#define SCAN_NT 256
#define SCAN_ITEMS 4
#define SCAN_TILE (SCAN_NT * SCAN_ITEMS)

__device__ __forceinline__ int block_scan_excl_i(int v, int *wsum, int &total)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = wave_scan_incl_i(v, lane);
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < SCAN_NT / 64; ++i) { int s = wsum[i]; if (i < w) base += s; tot += s; }
    total = tot;
    __syncthreads();
    return base + inc - v;
}

__global__ __launch_bounds__(SCAN_NT) void k_scan_tiles(const int *in, int *out, int n, int *tile_sums)
{
    __shared__ int wsum[SCAN_NT / 64];
    const int base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    int v[SCAN_ITEMS], s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) { v[k] = (base + k < n) ? in[base + k] : 0; s += v[k]; }
    int total;
    int ex = block_scan_excl_i(s, wsum, total);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) { if (base + k < n) out[base + k] = ex; ex += v[k]; }
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}

// single block: exclusive scan of the tile sums in place, grand total to *d_total
__global__ __launch_bounds__(SCAN_NT) void k_scan_sums(int *tile_sums, int ntiles, int *d_total)
{
    __shared__ int wsum[SCAN_NT / 64];
    int carry = 0;
    for (int base = 0; base < ntiles; base += SCAN_NT) {
        int i = base + threadIdx.x;
        int v = (i < ntiles) ? tile_sums[i] : 0;
        int total;
        int ex = block_scan_excl_i(v, wsum, total);
        if (i < ntiles) tile_sums[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0 && d_total) *d_total = carry;
}

__global__ __launch_bounds__(SCAN_NT) void k_scan_fix(int *out, int n, const int *tile_sums)
{
    const int add = tile_sums[blockIdx.x];
    const int base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) if (base + k < n) out[base + k] += add;
}

int dkmc_exclusive_scan_i32(const int *d_in, int *d_out, int n, int *d_total)
{
    if (n <= 0) { if (d_total) HIPCHK(hipMemsetAsync(d_total, 0, sizeof(int), eng().stream)); return 0; }
    const int ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    int *sums = (int *)scratch(S_SCAN_TMP, (size_t)ntiles * sizeof(int));
    if (!sums) return eng().err_code;
    hipStream_t st = eng().stream;
    hipLaunchKernelGGL(k_scan_tiles, dim3(ntiles), dim3(SCAN_NT), 0, st, d_in, d_out, n, sums);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(SCAN_NT), 0, st, sums, ntiles, d_total);
    hipLaunchKernelGGL(k_scan_fix, dim3(ntiles), dim3(SCAN_NT), 0, st, d_out, n, sums);
    KCHK();
    return 0;
}

// ---- 32-bit counts -> 64-bit offsets (row pointers of X beyond 2^31 non-zeros) -----------------------------------------
__global__ __launch_bounds__(SCAN_NT) void k_scan_sums64(const int *tile_sums, long long *tile_off, int ntiles, long long *d_total)
{
    // single block, sequential over chunks of SCAN_NT tiles; per-chunk scan in 64 bit through LDS
    __shared__ long long buf[SCAN_NT];
    long long carry = 0;
    for (int base = 0; base < ntiles; base += SCAN_NT) {
        const int i = base + threadIdx.x;
        buf[threadIdx.x] = (i < ntiles) ? (long long)tile_sums[i] : 0;
        __syncthreads();
        for (int off = 1; off < SCAN_NT; off <<= 1) {           // Hillis-Steele inclusive scan
            long long v = (threadIdx.x >= off) ? buf[threadIdx.x - off] : 0;
            __syncthreads();
            buf[threadIdx.x] += v;
            __syncthreads();
        }
        const long long incl = buf[threadIdx.x], total = buf[SCAN_NT - 1];
        if (i < ntiles) tile_off[i] = carry + incl - (long long)tile_sums[i];
        carry += total;
        __syncthreads();
    }
    if (threadIdx.x == 0 && d_total) *d_total = carry;
}

__global__ __launch_bounds__(SCAN_NT) void k_scan_fix64(const int *in_tile, long long *out, int n, const long long *tile_off)
{
    const long long add = tile_off[blockIdx.x];
    const int base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) if (base + k < n) out[base + k] = add + in_tile[base + k];
}

// exclusive prefix sum of n int32 counts into int64 offsets; out[n] is NOT written, total goes to *d_total
int dkmc_exclusive_scan_i32_i64(const int *d_in, long long *d_out, int n, long long *d_total)
{
    if (n <= 0) { if (d_total) HIPCHK(hipMemsetAsync(d_total, 0, sizeof(long long), eng().stream)); return 0; }
    const int ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    int *sums = (int *)scratch(S_SCAN_TMP, (size_t)ntiles * sizeof(int));
    long long *offs = (long long *)scratch(S_SCAN_OFF64, (size_t)ntiles * sizeof(long long));
    int *intile = (int *)scratch(S_SCAN_INTILE, (size_t)n * sizeof(int));
    if (!sums || !offs || !intile) return eng().err_code;
    hipStream_t st = eng().stream;
    hipLaunchKernelGGL(k_scan_tiles, dim3(ntiles), dim3(SCAN_NT), 0, st, d_in, intile, n, sums);
    hipLaunchKernelGGL(k_scan_sums64, dim3(1), dim3(SCAN_NT), 0, st, (const int *)sums, offs, ntiles, d_total);
    hipLaunchKernelGGL(k_scan_fix64, dim3(ntiles), dim3(SCAN_NT), 0, st, (const int *)intile, d_out, n, (const long long *)offs);
    KCHK();
    return 0;
}
