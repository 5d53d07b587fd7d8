// common.h -- shared device helpers and the process-wide engine state (one GPU per process).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include "../../include/devicekmc_hip.h"
#include "../../include/devicekmc_hip_debug.h"

// ELEMENT / EVENTTYPE (utils.h:37-60)
enum { DEFECT = 0, OXYGEN_DEFECT = 1, VACANCY = 2, O_EL = 3, Hf_EL = 4, Ni_EL = 5, Ti_EL = 6, Pt_EL = 7, N_EL = 8, NULL_ELEMENT = 9 };
enum { EV_GEN = 0, EV_REC = 1, EV_VDIFF = 2, EV_IDIFF = 3, EV_NULL = 4 };

#define DKMC_KB 8.617333262e-5       // kmc_events.cu:4
#define DKMC_Q 1.60217663e-19        // gpu_solvers.h:261
#define DKMC_HBAR 1.054571817e-34    // iterative_solvers_gpu.cu:8
#define DKMC_MAX_LAYERS 5            // kmc_events.cu:7
#define WAVE 64

// the list of metallic elements stays in device memory, as in the reference (is_in_array_gpu, gpu_solvers.h:211-220)
struct MetalSet { int n; const int *e; };

__device__ __forceinline__ bool is_metal(int el, const MetalSet &ms)
{
    bool r = false;
    for (int t = 0; t < ms.n; ++t) r |= (ms.e[t] == el);
    return r;
}

// gpu_solvers.h:225-257
__device__ __forceinline__ double site_dist(double x1, double y1, double z1, double x2, double y2, double z2,
                                            double laty, double latz, int pbc)
{
    if (pbc) {
        double dx = x1 - x2;
        double fy = (y1 - y2) / laty; fy -= round(fy);
        double fz = (z1 - z2) / latz; fz -= round(fz);
        double dy = fy * laty, dz = fz * latz;
        return sqrt(dx * dx + dy * dy + dz * dz);
    }
    double dx = x2 - x1, dy = y2 - y1, dz = z2 - z1;
    return sqrt(dx * dx + dy * dy + dz * dz);
}

// gpu_solvers.h:259-265
__device__ __forceinline__ double v_solve(double r, int charge, double sigma, double k)
{
    return (double)charge * erfc(r / (sigma * sqrt(2.0))) * k * DKMC_Q / r;
}

// ---- lane exchange without the LDS ------------------------------------------------------------
// hipcc lowers every __shfl_xor to ds_bpermute_b32 -- an LDS instruction per 32 bits, two per double.  Inside a row of 16 lanes the exchange
// "lane i <- lane i ^ OFF" (OFF = 1, 2, 4, 8) is a DPP move on the vector ALU (tools/probe_dpp_xor.hip: identical to __shfl_xor on gfx950); the
// per-row reductions of the sparse kernels were bound by the bpermutes, not by their memory traffic.  Offsets 16 / 32 stay on __shfl_xor.
template <int OFF> __device__ __forceinline__ int xor_lane_i(int x)
{
    static_assert(OFF == 1 || OFF == 2 || OFF == 4 || OFF == 8 || OFF == 16 || OFF == 32, "xor_lane: power of two below the wave size");
    if constexpr (OFF == 1) return __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, false);        // quad_perm [1,0,3,2]
    else if constexpr (OFF == 2) return __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
    else if constexpr (OFF == 4) {                                                                   // row_shl:4 into banks 0, 2; row_shr:4 into banks 1, 3
        const int t = __builtin_amdgcn_update_dpp(0, x, 0x104, 0xf, 0x5, false);
        return __builtin_amdgcn_update_dpp(t, x, 0x114, 0xf, 0xa, false);
    }
    else if constexpr (OFF == 8) return __builtin_amdgcn_update_dpp(0, x, 0x128, 0xf, 0xf, false);  // row_ror:8
    else return __shfl_xor(x, OFF, WAVE);
}
template <int OFF> __device__ __forceinline__ double xor_lane(double x)
{
    if constexpr (OFF >= 16) return __shfl_xor(x, OFF, WAVE);
    else return __hiloint2double(xor_lane_i<OFF>(__double2hiint(x)), xor_lane_i<OFF>(__double2loint(x)));
}
// butterfly sum over groups of W lanes (W a power of two), offsets W / 2, ..., 1 in that order: every lane of a group ends with the group's sum
template <int W> __device__ __forceinline__ double group_sum(double v)
{
    if constexpr (W >= 64) v += xor_lane<32>(v);
    if constexpr (W >= 32) v += xor_lane<16>(v);
    if constexpr (W >= 16) v += xor_lane<8>(v);
    if constexpr (W >= 8) v += xor_lane<4>(v);
    if constexpr (W >= 4) v += xor_lane<2>(v);
    if constexpr (W >= 2) v += xor_lane<1>(v);
    return v;
}
template <int W> __device__ __forceinline__ int group_sum_i(int v)
{
    if constexpr (W >= 64) v += xor_lane_i<32>(v);
    if constexpr (W >= 32) v += xor_lane_i<16>(v);
    if constexpr (W >= 16) v += xor_lane_i<8>(v);
    if constexpr (W >= 8) v += xor_lane_i<4>(v);
    if constexpr (W >= 4) v += xor_lane_i<2>(v);
    if constexpr (W >= 2) v += xor_lane_i<1>(v);
    return v;
}

// ---- wave64 / block reductions with a fixed combination order (deterministic) -----------------
// (lane 0 of the butterfly sees the same additions in the same order as the shift-down form it replaces: same bits)
__device__ __forceinline__ double wave_sum(double v) { return group_sum<WAVE>(v); }       // valid in lane 0 (and everywhere)
__device__ __forceinline__ double wave_sum_all(double v) { return group_sum<WAVE>(v); }   // same value in every lane
__device__ __forceinline__ int wave_sum_all_i(int v) { return group_sum_i<WAVE>(v); }
// inclusive scan across the wave
__device__ __forceinline__ double wave_scan_incl(double v, int lane)
{
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) { double t = __shfl_up(v, off, WAVE); if (lane >= off) v += t; }
    return v;
}
__device__ __forceinline__ int wave_scan_incl_i(int v, int lane)
{
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) { int t = __shfl_up(v, off, WAVE); if (lane >= off) v += t; }
    return v;
}

// block sum, result broadcast to all threads; red must hold blockDim.x/64 doubles
template <int NT>
__device__ __forceinline__ double block_sum_all(double v, double *red)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) s += red[i];
    return s;
}

// blocked form of a K pattern (kcg.hip: kblocked_build, once per pattern; device arrays)
struct KBlocked {
    int m, R, nb, total, maxwin, maxints;      // maxints: ints of the largest block's padded rows
    long long winsum;   // columns in all windows together
    int *perm;          // [m] row of the blocked order -> row of the pattern
    int *pcol;          // [total] columns in the blocked order, rows padded to 64 / 32 entries with the row itself, diagonal left out
    int4 *blk;          // [nb] {first column of the window, columns in the window, first int of the block in pcol, rows padded to 64}
};
// ---- host side ---------------------------------------------------------------------------------
struct Engine {
    hipStream_t stream = nullptr;
    int device = 0;
    double cg_tol = 1e-6;
    int current_warm_start = 1;    // start vector of the current solve (dkmc_set_current_warm_start): 1 previous solution (private unscaled copy), 0 the reference code's G0-scaled buffer
    int profiling = 0;
    int cb_edge_domain = 0;        // 0: CB-edge system over every site (snapshot source); 1: over atoms only (dkmc_set_cb_edge_domain)
    long long tcache_budget = -1;  // bytes the tunnelling-coefficient cache may take; -1 = a third of the free device memory, 8-128 GiB (dkmc_set_tcache_budget)
    double pair_cut = 6.5;         // screening cut-off of the pair sum in units of sigma sqrt 2 (dkmc_set_pair_cutoff; 0 = all pairs like the reference)
    int k_blocked = 1;             // build the blocked form of K patterns (dkmc_set_k_blocked; kcg.hip)
    int x_aux = 2;                 // auxiliary columns of the block-CG (dkmc_set_x_aux; xtb.hip): 0 hash set, 1 smooth set, 2 smooth at tolerances >= 1e-8
    int x_slab = 1;                // > 1 rank: distribute the STATE of the block-CG by row slabs (xtb_slab.inc; dkmc_set_x_slab); 0: all-gather variant (tile stream sharded only)
    int k_slab = 1;                // > 1 rank, system above the size of the blocked form: CG on K distributed by row slabs (kcg.hip; dkmc_set_k_slab); 0: replicated
    int x_poly = 8;                // degree d of the split polynomial preconditioner of the block-CG on one GPU (dkmc_set_x_poly; xtb.hip): the loop runs on L A L, L = the degree-d series of (I - N)^(-1/2) on the neighbour part; 0 = off
    int x_aux_warm = 0;            // 1: with the warm start of the current solve the hash auxiliary columns start from the previous solve's solutions too (dkmc_set_x_aux_warm; off: no gain beyond 1e4 rows, profiles/r05_ab_aux_warm.json)
    int x_items_kc = 0;            // > 0: overrides the nominal run length kc (tiles) of the tile runs (dkmc_set_x_items; measurement)
    int x_apply_form = 0;          // tile x panel kernel of the block-CG: 0 = the product form, 1 = the round-4 form of its loop (same results; same-box comparisons, dkmc_set_x_apply_form)
    int x_block = 16;              // block-CG width of the current solve on the tiled X (dkmc_set_x_block; xtb.hip): 16 by default, 1 = the reference's single-vector loop (its iterate sequence)
    int x_format = 1;              // 1: tiled X (xt.hip, default); 0: CSR X as the reference stores it (current.hip + cg.hip)
    int x_iter_hint = 0;           // iteration count of the previous CG solve of X (sizes the first launch batch)
    dkmc_stats stats{};
    char err[512] = {0};
    int err_code = 0;
    // per-layer energies (copytoConstMemory)
    double E_gen[DKMC_MAX_LAYERS] = {0}, E_rec[DKMC_MAX_LAYERS] = {0}, E_Vdiff[DKMC_MAX_LAYERS] = {0}, E_Odiff[DKMC_MAX_LAYERS] = {0};
    int num_layers = 0;
    // named persistent device buffers (grown on demand, never shrunk)
    static const int NBUF = 160;
    void *buf[NBUF] = {nullptr};
    size_t bufsz[NBUF] = {0};
};

Engine &eng();
int dkmc_fail(int code, const char *what, const char *file, int line);
// persistent scratch: returns a device buffer of at least `bytes`, identified by slot
void *scratch(int slot, size_t bytes);
inline MetalSet load_metals(const int *d_metals, int num_metals) { MetalSet ms; ms.n = num_metals; ms.e = d_metals; return ms; }

#define HIPCHK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) return dkmc_fail((int)e__, hipGetErrorString(e__), __FILE__, __LINE__); } while (0)
#define KCHK() do { hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess) return dkmc_fail((int)e__, hipGetErrorString(e__), __FILE__, __LINE__); } while (0)

// scratch slots
enum {
    S_CG_S = 0, S_CG_R, S_CG_P, S_CG_T, S_CG_PART, S_CG_CTRL, S_CG_RUNS, S_CG_REM, S_CG_NRUNS, S_CG_PS, S_CG_SEGOFF, S_CG_SEGS, S_CG_SEGPART, S_CG_XCHG, S_CG_PARTS,
    S_K_DATA, S_K_RHS,
    S_PW_LIST, S_PW_CNT, S_PW_CELLS, S_PW_PERM, S_PW_LIST2,
    S_EV_PROB, S_EV_ROWSUM, S_EV_G2, S_EV_G3, S_EV_CTRL, S_EV_UNI, S_EV_LOG,
    S_SCAN_TMP, S_SCAN_TMP2, S_SCAN_OFF64, S_SCAN_INTILE,
    S_AT_FLAG, S_AT_SITE, S_AT_OFSITE, S_AT_NEIGH,
    S_X_ROWPTR, S_X_COL, S_X_DATA, S_X_DATA2, S_X_CNT, S_X_SLIST, S_X_SCB, S_X_SFLAG, S_X_RHS, S_X_SRANK,
    S_P_IMACRO, S_HEAT, S_HEAT_A, S_HEAT_B, S_HEAT_Y,
    S_MISC0, S_MISC1, S_MISC2, S_MISC3,
    S_XT_DPOS, S_XT_SNODE_D, S_XT_SNODE_I, S_XT_CMASK, S_XT_ISTILE, S_XT_NSUBC, S_XT_TOFF, S_XT_SOFF, S_XT_TILES, S_XT_NITEMW, S_XT_WRANGE,
    S_XT_ITEMS, S_XT_SPLIT, S_XT_TVAL, S_XT_ROWPART, S_XT_COLPART, S_XT_CNT, S_XT_Q,
    S_XT_T_NITEMW, S_XT_T_ITEMS, S_XT_T_SPLIT, S_XT_T_COLPART, S_XT_T_MISC,
    S_XTB_PANELS, S_XTB_QS, S_XTB_ROWPART, S_XTB_COLPART, S_XTB_GRAM, S_XTB_SMALL, S_XTB_XI,
    S_XTB_SLAB_BOX, S_XTB_SLAB_TAB, S_XTB_SLAB_OWNER, S_XTB_SLAB_LISTS, S_XTB_SLAB_SDST, S_XTB_SLAB_FLAG, S_XTB_SLAB_RLISTS, S_XTB_SLAB_GX, S_XTB_SLAB_S1, S_XTB_SLAB_S3, S_XTB_SLAB_R3,
    S_XTB_EMU_Y, S_XTB_EMU_CTRL, S_XTB_YPANEL, S_XTB_PRE_V, S_XTB_PRE_W1, S_XTB_PRE_W2, S_XTB_PRE_Z,
    S_KS_TAB, S_KS_OWNER, S_KS_LISTS, S_KS_FLAG, S_KS_BOX, S_KS_RLISTS, S_KS_XA, S_KS_XB, S_KS_SEND, S_KS_RECV, S_KS_YBUF, S_KS_EMU,
    S_NSLOTS
};

// exchange step of the sharded current solve (comm.hip)
int comm_attached();
int comm_nranks();
int comm_rank();
int comm_allgather_f64(double *buf, size_t count);     // in place on the engine's stream; rank r owns buf[r*count, (r+1)*count)
int comm_alltoallv_f64(const double *sendbuf, double *recvbuf, const long long *cnt);   // rank s sends cnt[s * nranks + d] doubles to rank d (full table on every rank; pieces packed in destination / source order)
int comm_allreduce_sum_f64(double *buf, size_t count); // in place; the sum over the ranks (grouping of the additions is the transport's)
int comm_agree(int local_rc, const char *what);         // agreement point: non-zero on EVERY rank if any rank passed a non-zero local_rc (comm.hip)
int comm_agree_count();
int comm_peer_ready(size_t count);                      // the one-shot peer-write exchange is attached and its slots hold `count` doubles (comm.hip)
double *comm_peer_slots(int parity);
void comm_peer_drop();                                  // after a failed solve: the peers' sequence counters may have drifted (comm.hip)
int comm_peer_exchange(int parity, size_t count, int *ctrl_done, int *ctrl_aborted, int *ctrl_timeout, int stamp);
int comm_bcast0_f64(double *buf, size_t count);        // in place; every rank ends with rank 0's bits

// shared primitives (scan.hip)
// exclusive prefix sum of n ints (in -> out, out may alias in); total written to d_total (device int) if non-null
int dkmc_exclusive_scan_i32(const int *d_in, int *d_out, int n, int *d_total);
// same with 64-bit offsets (counts stay 32 bit)
int dkmc_exclusive_scan_i32_i64(const int *d_in, long long *d_out, int n, long long *d_total);
