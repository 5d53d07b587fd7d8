// neighbors.hip -- padded neighbour index on the device.
// Replaces the host's O(N^2) Device::constructSiteNeighborList (Device.cpp:98-136) and the padding loop (:69-80):
// row i holds every j != i with site_dist(i, j) < nn_dist in ascending j, padded with -1 to the global maximum nn.
// Cell list with cells of edge >= nn_dist: O(N) work, same distance arithmetic as gpu_solvers.h:225-257, so the
// result is identical to the reference's (asserted against the oracle in tests/test_gpu_parity.py).
#include "common.h"
#include <vector>

struct CellGrid {
    double x0, y0, z0, hx, hy, hz;     // origin and cell edge per axis
    int nx, ny, nz, pbc;
    double laty, latz;
};

__device__ __forceinline__ void cell_of(const CellGrid &G, double x, double y, double z, int &cx, int &cy, int &cz)
{
    cx = (int)((x - G.x0) / G.hx);
    if (G.pbc) {                        // y, z are periodic: bin the wrapped coordinate
        double fy = y / G.laty; fy -= floor(fy);
        double fz = z / G.latz; fz -= floor(fz);
        cy = (int)(fy * G.ny); cz = (int)(fz * G.nz);
    } else { cy = (int)((y - G.y0) / G.hy); cz = (int)((z - G.z0) / G.hz); }
    cx = min(max(cx, 0), G.nx - 1); cy = min(max(cy, 0), G.ny - 1); cz = min(max(cz, 0), G.nz - 1);
}

__global__ void k_minmax(int N, const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ z, double *out)
{
    __shared__ double red[6][256];
    double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < N; i += gridDim.x * 256) {
        mn[0] = fmin(mn[0], x[i]); mx[0] = fmax(mx[0], x[i]);
        mn[1] = fmin(mn[1], y[i]); mx[1] = fmax(mx[1], y[i]);
        mn[2] = fmin(mn[2], z[i]); mx[2] = fmax(mx[2], z[i]);
    }
    for (int k = 0; k < 3; ++k) { red[k][threadIdx.x] = mn[k]; red[3 + k][threadIdx.x] = mx[k]; }
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s)
            for (int k = 0; k < 3; ++k) {
                red[k][threadIdx.x] = fmin(red[k][threadIdx.x], red[k][threadIdx.x + s]);
                red[3 + k][threadIdx.x] = fmax(red[3 + k][threadIdx.x], red[3 + k][threadIdx.x + s]);
            }
        __syncthreads();
    }
    if (threadIdx.x < 6) out[blockIdx.x * 6 + threadIdx.x] = red[threadIdx.x][0];
}

__global__ void k_cell_count(int N, CellGrid G, const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ z,
                             int *__restrict__ cell_id, int *__restrict__ cell_cnt)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    int cx, cy, cz; cell_of(G, x[i], y[i], z[i], cx, cy, cz);
    const int c = (cx * G.ny + cy) * G.nz + cz;
    cell_id[i] = c;
    atomicAdd(&cell_cnt[c], 1);
}

__global__ void k_cell_scatter(int N, const int *__restrict__ cell_id, const int *__restrict__ cell_start, int *__restrict__ cell_fill,
                               int *__restrict__ cell_sites)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int c = cell_id[i];
    cell_sites[cell_start[c] + atomicAdd(&cell_fill[c], 1)] = i;     // order inside a cell is irrelevant: rows are sorted afterwards
}

// FILL = 0: count neighbours (row_cnt, global max); FILL = 1: write the row, sort it ascending, pad with -1
template <int FILL>
__global__ __launch_bounds__(128) void k_neigh(int N, CellGrid G, double nn_dist, const double *__restrict__ x, const double *__restrict__ y,
                                               const double *__restrict__ z, const int *__restrict__ cell_start, const int *__restrict__ cell_cnt,
                                               const int *__restrict__ cell_sites, int *__restrict__ max_nn, int nn, int *__restrict__ neigh)
{
    const int i = blockIdx.x * 128 + threadIdx.x;
    if (i >= N) return;
    const double xi = x[i], yi = y[i], zi = z[i];
    int cx, cy, cz; cell_of(G, xi, yi, zi, cx, cy, cz);
    int n = 0;
    int *row = FILL ? neigh + (size_t)i * nn : nullptr;
    for (int dx = -1; dx <= 1; ++dx) {
        const int ax = cx + dx;
        if (ax < 0 || ax >= G.nx) continue;
        for (int dy = -1; dy <= 1; ++dy) {
            int ay = cy + dy;
            if (G.pbc) ay = (ay + G.ny) % G.ny; else if (ay < 0 || ay >= G.ny) continue;
            for (int dz = -1; dz <= 1; ++dz) {
                int az = cz + dz;
                if (G.pbc) az = (az + G.nz) % G.nz; else if (az < 0 || az >= G.nz) continue;
                const int c = (ax * G.ny + ay) * G.nz + az;
                const int s0 = cell_start[c], s1 = s0 + cell_cnt[c];
                for (int s = s0; s < s1; ++s) {
                    const int j = cell_sites[s];
                    if (j == i) continue;
                    if (site_dist(xi, yi, zi, x[j], y[j], z[j], G.laty, G.latz, G.pbc) < nn_dist) {
                        if (FILL) { if (n < nn) row[n] = j; }
                        ++n;
                    }
                }
            }
        }
    }
    if (!FILL) { atomicMax(max_nn, n); return; }
    const int cnt = min(n, nn);
    for (int a = 1; a < cnt; ++a) {                       // insertion sort of the thread's own row (<= nn entries)
        const int v = row[a]; int b = a - 1;
        while (b >= 0 && row[b] > v) { row[b + 1] = row[b]; --b; }
        row[b + 1] = v;
    }
    for (int a = cnt; a < nn; ++a) row[a] = -1;
}

// Two-call protocol: d_neigh_out == NULL computes the maximum neighbour count into *nn_out; the second call (same
// positions) fills d_neigh_out[N * nn] with nn = *nn_out.
extern "C" int dkmc_build_neighbor_index(int N, const double *d_x, const double *d_y, const double *d_z, const double *h_lattice,
                                         int pbc, double nn_dist, int *nn_out, int *d_neigh_out)
{
    Engine &e = eng(); hipStream_t st = e.stream;
    if (N <= 0) return dkmc_fail(13, "build_neighbor_index: no sites", __FILE__, __LINE__);
    // bounding box
    const int nb = 256;
    double *mm = (double *)scratch(S_MISC0, (size_t)nb * 6 * 8);
    if (!mm) return e.err_code;
    hipLaunchKernelGGL(k_minmax, dim3(nb), dim3(256), 0, st, N, d_x, d_y, d_z, mm);
    std::vector<double> h((size_t)nb * 6);
    HIPCHK(hipMemcpyAsync(h.data(), mm, h.size() * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int b = 0; b < nb; ++b) for (int k = 0; k < 3; ++k) { mn[k] = fmin(mn[k], h[b * 6 + k]); mx[k] = fmax(mx[k], h[b * 6 + 3 + k]); }
    CellGrid G; G.pbc = pbc; G.laty = h_lattice[1]; G.latz = h_lattice[2];
    G.x0 = mn[0]; G.y0 = mn[1]; G.z0 = mn[2];
    G.nx = (int)((mx[0] - mn[0]) / nn_dist) + 1; G.hx = nn_dist;
    if (pbc) {
        G.ny = (int)(G.laty / nn_dist); G.nz = (int)(G.latz / nn_dist);       // cells at least nn_dist wide
        if (G.ny < 3 || G.nz < 3) return dkmc_fail(14, "build_neighbor_index: periodic box narrower than 3 cells", __FILE__, __LINE__);
        G.hy = G.laty / G.ny; G.hz = G.latz / G.nz;
    } else {
        G.ny = (int)((mx[1] - mn[1]) / nn_dist) + 1; G.nz = (int)((mx[2] - mn[2]) / nn_dist) + 1; G.hy = G.hz = nn_dist;
    }
    const long long ncell = (long long)G.nx * G.ny * G.nz;
    if (ncell > (1ll << 30)) return dkmc_fail(15, "build_neighbor_index: too many cells", __FILE__, __LINE__);
    int *cell_id = (int *)scratch(S_AT_FLAG, (size_t)N * 4);
    int *cell_cnt = (int *)scratch(S_MISC1, (size_t)(ncell + 4) * 4 * 3);
    int *cell_sites = (int *)scratch(S_AT_SITE, (size_t)N * 4);
    int *mx_nn = (int *)scratch(S_MISC2, 16);
    if (!cell_id || !cell_cnt || !cell_sites || !mx_nn) return e.err_code;
    int *cell_start = cell_cnt + (ncell + 4), *cell_fill = cell_start + (ncell + 4);
    HIPCHK(hipMemsetAsync(cell_cnt, 0, (size_t)(ncell + 4) * 4 * 3, st));
    HIPCHK(hipMemsetAsync(mx_nn, 0, 16, st));
    const int gb = (N + 255) / 256;
    hipLaunchKernelGGL(k_cell_count, dim3(gb), dim3(256), 0, st, N, G, d_x, d_y, d_z, cell_id, cell_cnt);
    int rc = dkmc_exclusive_scan_i32(cell_cnt, cell_start, (int)ncell, nullptr); if (rc) return rc;
    hipLaunchKernelGGL(k_cell_scatter, dim3(gb), dim3(256), 0, st, N, (const int *)cell_id, (const int *)cell_start, cell_fill, cell_sites);
    const int gn = (N + 127) / 128;
    if (!d_neigh_out) {
        hipLaunchKernelGGL((k_neigh<0>), dim3(gn), dim3(128), 0, st, N, G, nn_dist, d_x, d_y, d_z, (const int *)cell_start, (const int *)cell_cnt,
                           (const int *)cell_sites, mx_nn, 0, (int *)nullptr);
        KCHK();
        HIPCHK(hipMemcpyAsync(nn_out, mx_nn, sizeof(int), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        return 0;
    }
    hipLaunchKernelGGL((k_neigh<1>), dim3(gn), dim3(128), 0, st, N, G, nn_dist, d_x, d_y, d_z, (const int *)cell_start, (const int *)cell_cnt,
                       (const int *)cell_sites, mx_nn, *nn_out, d_neigh_out);
    KCHK();
    HIPCHK(hipStreamSynchronize(st));
    return 0;
}
