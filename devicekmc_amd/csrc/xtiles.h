// xtiles.h -- layout of the tiled form of X shared by xt.hip (assembly, single-vector CG, power pass) and xtb.hip (block-CG with
// the MFMA tile x panel product).  See the head of xt.hip for the storage scheme.
#pragma once
#include "xshared.h"

#define XT_R 32                      // S-rows per tile
#define XT_C 256                     // S-columns per tile
#define XT_SBW 32                    // columns per sub-block
#define XT_SUB (XT_R * XT_SBW)       // doubles per sub-block (8 KiB)
#define XT_NT 256
#define XT_MAXKC 32                   // largest nominal run length (tiles) of a wave: min(XT_MAXKC, tiles / ranks / 4096); 16 until round 5 (profiles/r05_ab_tile_run_lists.jsonl)
#define XT_PROF_STRIDE 8

typedef double dbl2 __attribute__((ext_vector_type(2)));

struct __attribute__((aligned(16))) XTile { int k, w; unsigned mask; int soff; };   // cell (k, w); present sub-blocks; first sub-block slot
struct __attribute__((aligned(32))) XItem { int t0, t1, w, c, k0; unsigned mask0; int soff0, pad; };   // tiles [t0, t1) of strip w; c = position of the run in its strip; descriptor of tile t0; pad = record of its column sums, colpart[pad * 256]: the run's own (c = 0; pad = its index in the item list) or, on one GPU, one record per aligned group of four runs -- the four waves of a workgroup (c = 1; pad = index / 4; every strip's run count padded to a multiple of four with empty runs)
struct XCtrl { double rr[2]; int done_local, sharded; int done; int iters; int abort_local, aborted; int pad[2]; int xchg_timeout, pad2; };
// `done` (non-zero) gates every kernel of the loop; see k_xt_step for its iteration stamp.  Single GPU: set by the step kernel.  Sharded solve: the direction kernel only sets
// done_local; rank 0's done_local travels in the all-reduced buffer (slot ns) and k_xt_rows_apply turns it into `done` on every
// rank in the same iteration -- all control flow derives from data every rank received from the same collective, so the ranks
// cannot leave the loop at different iterations even if their arithmetic differed in a bit.

struct SNodes {                      // S in rank order, padded to a multiple of XT_C (flag 0 = no entries)
    const double *x, *y, *z, *cb;
    const int *flag, *slot, *mr;     // class flags; row of the coefficient cache (vacancies) / column (metals), -1 if none
    const int *slotA;                // row of the cache's part A (sharded solve: left-contact columns of this rank's vacancies), -1 if none
};

struct XTState {
    // shape of the last assembly
    int Nsub = 0, ns = 0, ns_pad = 0, nK = 0, nW = 0, ntiles = 0, nitems = 0, kc = 1, maxchunk = 1, rec_shift = 0;
    long long nsub_total = 0, xs_nnz = 0;
    unsigned long long t_upper = 0;
    // this rank's share
    int item_lo = 0, item_n = 0, tile_lo = 0, tile_n = 0, w_lo = 0, w_hi = 0; long long sub_base = 0, sub_n = 0;
    bool valid = false;
};

struct XTBuffers {
    SNodes S; int *srow; XTile *tiles; XItem *items; int2 *wrange; int *nitem_w; double *tval, *rowpart, *colpart;
    xrp_t *rp, *dpos; int *ci; double *val; int *nsrank;
    unsigned *cmask; int *toff;      // census of the last assembly (kept for dkmc_xt_time_share)
    const double *ax, *ay, *az;      // atom positions of the last assembly (dkmc_xtb_emulate_slabs)
};

extern XTState g_xt;
extern XTBuffers g_xb;

static inline int xt_grid(long long work, int per_block, int cap)
{
    long long b = (work + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (int)b;
}

// ---- block-CG on the tiled X (xtb.hip) ----
#define DKMC_XTB_AGAIN 1001          // xtb_cg_body (preconditioned solve): the true residual does not meet the stop test yet; y holds the iterate reached
#define DKMC_XTB_BREAKDOWN 1000      // xtb_cg: an s x s system lost definiteness; y holds the last good iterate (not an error)
struct XtbArgs {
    int m, ns, ns_pad, nK, nW;                    // system rows; |S|; padded |S|; row blocks; windows
    int s;                                        // block width (2 ... 16)
    const XItem *items; int item_n;               // this rank's runs (padded to groups of four per strip)
    const XTile *tiles; int sub_base; const double *tval;
    const int2 *wrange; const int *nitem_w; int nrecords;
    const int *srow; const double *sS; const int *nsrank;
    const xrp_t *rp; const int *ci; const double *val;      // neighbour part Xs (CSR, unscaled, diagonal included)
    const double *sc;                             // Jacobi scaling 1 / sqrt(diag)
    const double *ax, *ay, *az;                   // position of the atom of row r >= 2 at [r - 2] (smooth auxiliary columns; may be null)
    const double *b;                              // scaled right-hand side
    double *y;                                    // in: scaled start vector y / s; out: scaled solution
    double *yaux; int yaux_valid;                 // UNSCALED solutions of the auxiliary columns, [m][16] (may be null): in (if valid) the previous solve's, out this solve's
    XCtrl *ctrl; double tol2; bool nt_loads;
    bool sharded; int w_lo, w_hi;                 // sharded solve (comm.hip): windows of this rank's tiles
};
int xtb_cg(const XtbArgs &A, int *iters_out, double *rr_out);
// one rank's share of the work items of an nranks-way split (xt.hip: xt_build_items)
struct XShare { int nitems, maxchunk, item_lo, item_n, tile_lo, tile_n, w_lo, w_hi; long long sub_base, sub_n; XItem *items; int *nitem_w; int rec_shift; };
// the slab-distributed block-CG with nr VIRTUAL ranks in this process (xtb_slab.inc); times_us[8]: mean kernel times of virtual rank time_rank
// (apply, neighbour part, fold, rows, Gram reduction, s x s algebra, step, pack + unpack); xdoubles[3]: doubles a rank receives per sweep in the three exchanges
int xtb_cg_slab_emulate(const XtbArgs &A, int nr, const XShare *shares, int time_rank, int sweep_cap, int *iters_out, double *rr_out, double *times_us, long long *xdoubles);
__global__ void k_xt_vec_mul(int m, double *__restrict__ y, const double *__restrict__ s);
