// comm.hip -- the exchange step of the multi-GPU current solve (SURVEY 8e, row "X-CG").
//
// The reference is single-GPU.  Here N processes (one per GPU) advance the SAME simulation in lockstep; every phase is
// computed redundantly and identically on every rank except the dominant one, the matrix stream of A*p in the CG solve of X.
// Two variants, one collective per matrix-vector product each:
//   * tiled X (xt.hip, default): the work items (runs of tiles of the tunnelling block) are dealt to the ranks in contiguous,
//     byte-balanced shares; a rank generates, stores and streams only its tiles, forms its partial sum of every S-row and ONE
//     in-place all-reduce (|S| + 1 doubles: the row sums and rank 0's stop decision) completes them.  Every rank receives the
//     same stop flag (only rank 0 contributes to it, so it is exact whatever the all-reduce's grouping) and takes its control
//     decisions from that buffer alone, so no rank can leave the iteration loop while another waits in the collective.  The
//     solution and the power row sums are re-published from rank 0, so the ranks' states stay bit-identical even under an
//     all-reduce whose ranks add in different orders.  The result equals the single-GPU one to rounding.
//   * CSR X (cg.hip, dkmc_set_x_format(0)): the long rows are dealt to the ranks at row boundaries, balanced by segment count; a
//     rank multiplies the segments of its rows and adds them per row; one in-place all-gather hands every rank every row sum.
//     Values and summation orders are those of the single-GPU kernels: bit-identical to the single-GPU run.  (This variant
//     relies on every rank computing identical dot products from identical data; it carries no flag in the exchange.)
// Everything downstream (dot products, vector updates) is computed identically everywhere: no dot-product all-reduce is needed.
// The pair sum (potential.hip) uses the all-gather too: a slab of sites per rank, the potentials gathered in place.
//
// Two transports behind the same call:
//   * RCCL over xGMI (production): ncclAllGather on the engine's stream.  librccl is opened at run time (dlopen) so that
//     the library keeps loading on machines without RCCL; inside a PyTorch process this binds to the librccl torch has
//     already loaded (same soname), exactly like libamdhip64.
//   * host callback (rehearsal and tests): the chunk is staged through pinned host memory and a caller-supplied function
//     (e.g. torch.distributed over gloo) performs the all-gather.  Needs a stream synchronisation per exchange; it exists so
//     that the partitioning and lockstep logic can be exercised by several ranks sharing one GPU, which RCCL refuses.
#include "common.h"
#include <dlfcn.h>

struct nccl_id_t { char internal[128]; };          // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128), passed by value
static const int NCCL_FLOAT64 = 8;                 // ncclDataType_t: ncclDouble
typedef int (*fn_get_unique_id)(void *);
typedef int (*fn_comm_init_rank)(void **, int, nccl_id_t, int);
typedef int (*fn_comm_destroy)(void *);
typedef int (*fn_all_gather)(const void *, void *, size_t, int, void *, hipStream_t);
typedef int (*fn_all_reduce)(const void *, void *, size_t, int, int, void *, hipStream_t);
typedef const char *(*fn_error_string)(int);

struct Comm {
    int transport = DKMC_COMM_NONE, nranks = 1, rank = 0;
    // RCCL
    void *dl = nullptr, *nccl = nullptr;
    fn_get_unique_id get_unique_id = nullptr;
    fn_comm_init_rank comm_init_rank = nullptr;
    fn_comm_destroy comm_destroy = nullptr;
    fn_all_gather all_gather = nullptr;
    fn_all_reduce all_reduce = nullptr;
    fn_error_string error_string = nullptr;
    // host callback
    dkmc_allgather_fn cb = nullptr; void *cb_user = nullptr;
    double *stage = nullptr; size_t stage_bytes = 0;
};
static Comm g_comm;

static int rccl_open()
{
    Comm &c = g_comm;
    if (c.dl) return 0;
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);          // the copy already in the process (torch's), if any
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return dkmc_fail(40, "comm: librccl.so.1 not found (dlopen)", __FILE__, __LINE__);
    c.get_unique_id = (fn_get_unique_id)dlsym(h, "ncclGetUniqueId");
    c.comm_init_rank = (fn_comm_init_rank)dlsym(h, "ncclCommInitRank");
    c.comm_destroy = (fn_comm_destroy)dlsym(h, "ncclCommDestroy");
    c.all_gather = (fn_all_gather)dlsym(h, "ncclAllGather");
    c.all_reduce = (fn_all_reduce)dlsym(h, "ncclAllReduce");
    c.error_string = (fn_error_string)dlsym(h, "ncclGetErrorString");
    if (!c.get_unique_id || !c.comm_init_rank || !c.comm_destroy || !c.all_gather || !c.all_reduce || !c.error_string)
        return dkmc_fail(41, "comm: librccl lacks a required symbol", __FILE__, __LINE__);
    c.dl = h;
    return 0;
}

#define RCCLCHK(x) do { int r__ = (x); if (r__ != 0) return dkmc_fail(42, g_comm.error_string(r__), __FILE__, __LINE__); } while (0)

extern "C" int dkmc_comm_unique_id(char *id128)
{
    if (int rc = rccl_open()) return rc;
    nccl_id_t id; memset(&id, 0, sizeof(id));
    RCCLCHK(g_comm.get_unique_id(&id));
    memcpy(id128, id.internal, 128);
    return 0;
}

extern "C" int dkmc_comm_destroy(void)
{
    Comm &c = g_comm;
    if (c.transport == DKMC_COMM_RCCL && c.nccl) { (void)hipStreamSynchronize(eng().stream); c.comm_destroy(c.nccl); c.nccl = nullptr; }
    if (c.stage) { (void)hipHostFree(c.stage); c.stage = nullptr; c.stage_bytes = 0; }
    c.transport = DKMC_COMM_NONE; c.nranks = 1; c.rank = 0; c.cb = nullptr; c.cb_user = nullptr;
    eng().x_iter_hint = 0;         // rank-local history must not shape the launch plan of a sharded solve
    return 0;
}

extern "C" int dkmc_comm_init_rccl(int nranks, int rank, const char *id128)
{
    if (nranks < 1 || rank < 0 || rank >= nranks) return dkmc_fail(43, "comm: bad rank / nranks", __FILE__, __LINE__);
    if (int rc = rccl_open()) return rc;
    dkmc_comm_destroy();
    nccl_id_t id; memcpy(id.internal, id128, 128);
    RCCLCHK(g_comm.comm_init_rank(&g_comm.nccl, nranks, id, rank));
    g_comm.transport = DKMC_COMM_RCCL; g_comm.nranks = nranks; g_comm.rank = rank;
    return 0;
}

extern "C" int dkmc_comm_init_host(int nranks, int rank, dkmc_allgather_fn fn, void *user)
{
    if (nranks < 1 || rank < 0 || rank >= nranks || !fn) return dkmc_fail(43, "comm: bad rank / nranks / callback", __FILE__, __LINE__);
    dkmc_comm_destroy();
    g_comm.transport = DKMC_COMM_HOST; g_comm.nranks = nranks; g_comm.rank = rank; g_comm.cb = fn; g_comm.cb_user = user;
    return 0;
}

extern "C" int dkmc_comm_info(int *nranks, int *rank, int *transport)
{
    if (nranks) *nranks = g_comm.nranks;
    if (rank) *rank = g_comm.rank;
    if (transport) *transport = g_comm.transport;
    return 0;
}

// host half of the callback transport, callable without a GPU (tests): buf holds nranks * count doubles, the caller's
// chunk already in place
extern "C" int dkmc_comm_allgather_host(double *buf, size_t count)
{
    Comm &c = g_comm;
    if (c.transport != DKMC_COMM_HOST) return dkmc_fail(44, "comm: host transport not attached", __FILE__, __LINE__);
    if (int rc = c.cb(buf, count * sizeof(double), c.rank, c.nranks, c.cb_user)) return dkmc_fail(45, "comm: all-gather callback failed", __FILE__, __LINE__);
    return 0;
}

// ---- used by the solver -------------------------------------------------------------------------------------------------
int comm_attached() { return g_comm.transport != DKMC_COMM_NONE; }
int comm_nranks() { return g_comm.nranks; }
int comm_rank() { return g_comm.rank; }

// in-place all-gather of doubles on the engine's stream: rank r's chunk is buf[r*count .. (r+1)*count)
int comm_allgather_f64(double *buf, size_t count)
{
    Comm &c = g_comm; hipStream_t st = eng().stream;
    if (c.transport == DKMC_COMM_RCCL) {
        RCCLCHK(c.all_gather(buf + (size_t)c.rank * count, buf, count, NCCL_FLOAT64, c.nccl, st));
        return 0;
    }
    if (c.transport == DKMC_COMM_HOST) {
        const size_t bytes = (size_t)c.nranks * count * sizeof(double);
        if (c.stage_bytes < bytes) {
            if (c.stage) (void)hipHostFree(c.stage);
            c.stage = nullptr; c.stage_bytes = 0;
            HIPCHK(hipHostMalloc((void **)&c.stage, bytes, hipHostMallocDefault));
            c.stage_bytes = bytes;
        }
        double *mine = c.stage + (size_t)c.rank * count;
        HIPCHK(hipMemcpyAsync(mine, buf + (size_t)c.rank * count, count * sizeof(double), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (int rc = c.cb(c.stage, count * sizeof(double), c.rank, c.nranks, c.cb_user)) return dkmc_fail(45, "comm: all-gather callback failed", __FILE__, __LINE__);
        HIPCHK(hipMemcpyAsync(buf, c.stage, bytes, hipMemcpyHostToDevice, st));
        return 0;
    }
    return 0;
}

// in-place sum over the ranks of `count` doubles on the engine's stream.  Ring and tree all-reduces hand every rank the same bits;
// nothing here RELIES on that: control flow is taken from a flag only rank 0 contributes to (exact under any grouping of the
// additions), and results that feed later phases are re-published from rank 0 (comm_bcast0_f64).
int comm_allreduce_sum_f64(double *buf, size_t count)
{
    Comm &c = g_comm; hipStream_t st = eng().stream;
    if (c.transport == DKMC_COMM_RCCL) {
        RCCLCHK(c.all_reduce(buf, buf, count, NCCL_FLOAT64, /* ncclSum */ 0, c.nccl, st));
        return 0;
    }
    if (c.transport == DKMC_COMM_HOST) {
        const size_t bytes = (size_t)c.nranks * count * sizeof(double);
        if (c.stage_bytes < bytes) {
            if (c.stage) (void)hipHostFree(c.stage);
            c.stage = nullptr; c.stage_bytes = 0;
            HIPCHK(hipHostMalloc((void **)&c.stage, bytes, hipHostMallocDefault));
            c.stage_bytes = bytes;
        }
        double *mine = c.stage + (size_t)c.rank * count;
        HIPCHK(hipMemcpyAsync(mine, buf, count * sizeof(double), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (int rc = c.cb(c.stage, count * sizeof(double), c.rank, c.nranks, c.cb_user)) return dkmc_fail(45, "comm: all-gather callback failed", __FILE__, __LINE__);
        for (int r = 1; r < c.nranks; ++r) { const double *src = c.stage + (size_t)r * count; for (size_t i = 0; i < count; ++i) c.stage[i] += src[i]; }
        HIPCHK(hipMemcpyAsync(buf, c.stage, count * sizeof(double), hipMemcpyHostToDevice, st));
        HIPCHK(hipStreamSynchronize(st));                 // the staging buffer is reused by the next call
        return 0;
    }
    return 0;
}

// Agreement point of a sharded phase: every rank contributes whether its LOCAL work so far succeeded; if any rank failed, every rank
// returns an error (the failing rank its own code, the others 46) -- nobody is left waiting in the collective that would have
// followed.  Callers make sure every rank reaches exactly one agreement point per phase, whatever happened before it.
static int g_agree_count = 0;
int comm_agree_count() { return g_agree_count; }
int comm_agree(int local_rc, const char *what)
{
    Comm &c = g_comm; Engine &e = eng(); hipStream_t st = e.stream;
    if (c.transport == DKMC_COMM_NONE) return local_rc;
    ++g_agree_count;
    // the word travels in its own small device buffer: the engine's scratch may be what failed
    static double *d_word = nullptr;
    if (!d_word && hipMalloc((void **)&d_word, 64) != hipSuccess) d_word = nullptr;
    const double mine = local_rc ? 1.0 : 0.0;
    double total = mine;
    bool moved = false;
    if (d_word && hipMemcpyAsync(d_word, &mine, 8, hipMemcpyHostToDevice, st) == hipSuccess) {
        const int saved_code = e.err_code; char saved_msg[sizeof(e.err)]; memcpy(saved_msg, e.err, sizeof(e.err));
        if (comm_allreduce_sum_f64(d_word, 1) == 0 && hipStreamSynchronize(st) == hipSuccess &&
            hipMemcpy(&total, d_word, 8, hipMemcpyDeviceToHost) == hipSuccess) moved = true;
        if (local_rc) { e.err_code = saved_code; memcpy(e.err, saved_msg, sizeof(e.err)); }       // keep the first error's text
    }
    if (local_rc) return local_rc;
    if (!moved) return dkmc_fail(45, "comm: agreement exchange failed", __FILE__, __LINE__);
    if (total != 0.0) { char msg[160]; snprintf(msg, sizeof(msg), "a peer rank failed in: %s", what ? what : "a sharded phase"); return dkmc_fail(46, msg, __FILE__, __LINE__); }
    return 0;
}

// every rank ends with rank 0's `count` doubles: the other ranks contribute zeros to a sum, which is exact whatever order the
// transport adds in (x + 0 + ... + 0 = x), so no rank can be left with different bits.  Used once per solve on its results, so that
// the replicated phases downstream (power, temperature, event rates) start from identical state on every rank.
int comm_bcast0_f64(double *buf, size_t count)
{
    Comm &c = g_comm; hipStream_t st = eng().stream;
    if (c.transport == DKMC_COMM_NONE || count == 0) return 0;
    if (c.rank != 0) HIPCHK(hipMemsetAsync(buf, 0, count * sizeof(double), st));
    return comm_allreduce_sum_f64(buf, count);
}
