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
#include <stdlib.h>

struct nccl_id_t { char internal[128]; };          // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128), passed by value
static const int NCCL_FLOAT64 = 8;                 // ncclDataType_t: ncclDouble
typedef int (*fn_get_unique_id)(void *);
typedef int (*fn_comm_init_rank)(void **, int, nccl_id_t, int);
typedef int (*fn_comm_destroy)(void *);
typedef int (*fn_all_gather)(const void *, void *, size_t, int, void *, hipStream_t);
typedef int (*fn_all_reduce)(const void *, void *, size_t, int, int, void *, hipStream_t);
typedef const char *(*fn_error_string)(int);
typedef int (*fn_send)(const void *, size_t, int, int, void *, hipStream_t);
typedef int (*fn_recv)(void *, size_t, int, int, void *, hipStream_t);
typedef int (*fn_group)(void);

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
    fn_send send = nullptr; fn_recv recv = nullptr; fn_group group_start = nullptr, group_end = nullptr;       // point-to-point (all-to-all-v of the slab-distributed block-CG)
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
    c.send = (fn_send)dlsym(h, "ncclSend"); c.recv = (fn_recv)dlsym(h, "ncclRecv");
    c.group_start = (fn_group)dlsym(h, "ncclGroupStart"); c.group_end = (fn_group)dlsym(h, "ncclGroupEnd");
    if (!c.get_unique_id || !c.comm_init_rank || !c.comm_destroy || !c.all_gather || !c.all_reduce || !c.error_string)
        return dkmc_fail(41, "comm: librccl lacks a required symbol", __FILE__, __LINE__);
    c.dl = h;
    return 0;
}

static void peer_release();
#define PEER_MAXR_A2A 64
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
    peer_release();
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

#include <algorithm>

// ---- one-shot peer-write exchange of the sharded block-CG ----------------------------------------------------------
// Beside either transport: every rank owns an exchange buffer of 2 x nranks slots and a row of sequence words, both exported with
// hipIpcGetMemHandle and mapped by every peer once (dkmc_comm_peer_prepare -> handles exchanged by the host through the process group ->
// dkmc_comm_peer_attach).  One exchange = three small launches on the engine's stream, no collective call and no host synchronisation:
//   k_peer_push    copies this rank's slot into the same slot of every peer's buffer (direct stores over xGMI / within the device),
//   k_peer_signal  after a system-scope fence, stores the exchange's sequence number into word `me` of every rank's sequence row,
//   k_peer_wait    one lane per rank polls this rank's own row until every word has reached the sequence number (bounded: after
//                  PEER_TIMEOUT_S it gives up, marks the solve aborted and the caller returns an error instead of hanging the GPU).
// The slots are then added IN RANK ORDER by the caller's kernel (k_xtb_rows<., 1>): identical bits on every rank by construction.
// Two slot sets alternate (parity of the exchange): a peer may push exchange n + 1 while this rank still reads the slots of n; it cannot
// push n + 2 before this rank has pushed n + 1, i.e. finished reading n.
// Memory: slots and sequence rows are allocated FINE-GRAINED (hipExtMallocWithFlags, hipDeviceMallocFinegrained): a peer device's
// stores become visible to this device's loads without a kernel boundary, which the poll in k_peer_wait and the slot reads of
// k_xtb_rows behind it rely on (coarse-grained hipMalloc memory is only coherent across devices at kernel boundaries: a poll could
// sit on a stale line, a slot read could hit the L2 copy of exchange n - 2).  Flags are written / read with system-scope release /
// acquire.  Tested with two processes sharing ONE GPU (tests/test_dist_sharded.py), which is all the development box allows: an attach
// whose ranks sit on DIFFERENT devices (PCI bus ids travel with the handles) is refused unless DKMC_PEER_CROSS_DEVICE=1 is set AND the
// fine-grained allocation succeeded -- the caller (parallel.attach_peer_exchange) then stays on the communicator's all-gather.
#define PEER_MAXR 16
#define PEER_TIMEOUT_S 10.0
struct Peer {
    bool ready = false, finegrained = false;
    size_t slot = 0;                                   // doubles per slot
    double *buf = nullptr;                             // local: 2 * nranks * slot doubles
    unsigned long long *flags = nullptr;               // local: PEER_MAXR words (sequence number of the last complete push of rank r)
    void *mapped[2 * PEER_MAXR] = {};                  // what hipIpcOpenMemHandle returned (closed on destroy)
    double **rbuf_d = nullptr; unsigned long long **rflags_d = nullptr;      // device tables of the nranks buffers / sequence rows
    unsigned long long seq = 0;
    long long exchanges = 0; double ms = 0.0; int timed = 0;
    hipEvent_t ev[2] = {nullptr, nullptr};
};
static Peer g_peer;

static void peer_release()
{
    Peer &p = g_peer;
    if (p.ready || p.buf) (void)hipStreamSynchronize(eng().stream);
    for (auto &m : p.mapped) if (m) { (void)hipIpcCloseMemHandle(m); m = nullptr; }
    if (p.buf) (void)hipFree(p.buf);
    if (p.flags) (void)hipFree(p.flags);
    if (p.rbuf_d) (void)hipFree(p.rbuf_d);
    if (p.rflags_d) (void)hipFree(p.rflags_d);
    for (auto &ev : p.ev) if (ev) { (void)hipEventDestroy(ev); ev = nullptr; }
    p = Peer{};
}

// allocates this rank's exchange buffer and sequence row; handles192 receives the two IPC handles (64 bytes each) and, in the last 64
// bytes, this rank's device identity: [0] 'F' / 'C' (fine- / coarse-grained allocation), [1...] the PCI bus id string of its device
static int peer_alloc(void **ptr, size_t bytes, bool fine)
{
    if (fine) { if (hipExtMallocWithFlags(ptr, bytes, hipDeviceMallocFinegrained) == hipSuccess) return 0; (void)hipGetLastError(); *ptr = nullptr; return 1; }
    HIPCHK(hipMalloc(ptr, bytes));
    return 0;
}
extern "C" int dkmc_comm_peer_prepare(size_t slot_doubles, char *handles192)
{
    Comm &c = g_comm; Peer &p = g_peer;
    if (c.transport == DKMC_COMM_NONE || c.nranks > PEER_MAXR || slot_doubles == 0 || !handles192)
        return dkmc_fail(47, "comm: peer exchange needs an attached communicator of at most 16 ranks", __FILE__, __LINE__);
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "handle size");
    peer_release();
    p.slot = slot_doubles;
    const size_t bbytes = (size_t)2 * c.nranks * slot_doubles * 8, fbytes = PEER_MAXR * sizeof(unsigned long long);
    hipIpcMemHandle_t h[2];
    // fine-grained first (see the header comment); a runtime that cannot allocate or export such memory gets the coarse-grained
    // allocation, which is only valid between ranks on ONE device (attach checks)
    bool fine = getenv("DKMC_PEER_COARSE") == nullptr;
    for (int attempt = 0; attempt < 2; ++attempt, fine = false) {
        if (peer_alloc((void **)&p.buf, bbytes, fine) == 0 && peer_alloc((void **)&p.flags, fbytes, fine) == 0 &&
            hipIpcGetMemHandle(&h[0], p.buf) == hipSuccess && hipIpcGetMemHandle(&h[1], p.flags) == hipSuccess) { p.finegrained = fine; break; }
        (void)hipGetLastError();
        if (p.buf) { (void)hipFree(p.buf); p.buf = nullptr; }
        if (p.flags) { (void)hipFree(p.flags); p.flags = nullptr; }
        if (!fine) return dkmc_fail(47, "comm: peer exchange: could not allocate / export the exchange buffers", __FILE__, __LINE__);
    }
    HIPCHK(hipMemset(p.flags, 0, fbytes));
    HIPCHK(hipDeviceSynchronize());
    memcpy(handles192, &h[0], 64); memcpy(handles192 + 64, &h[1], 64);
    char *id = handles192 + 128; memset(id, 0, 64);
    id[0] = p.finegrained ? 'F' : 'C';
    int dev = 0; HIPCHK(hipGetDevice(&dev));
    if (hipDeviceGetPCIBusId(id + 1, 62, dev) != hipSuccess) { (void)hipGetLastError(); snprintf(id + 1, 62, "device-%d", dev); }
    return 0;
}
// all_handles: nranks x 192 bytes in rank order (this rank's own handles are ignored).  Every rank must have called prepare with the same slot size.
extern "C" int dkmc_comm_peer_attach(const char *all_handles)
{
    Comm &c = g_comm; Peer &p = g_peer;
    if (!p.buf || !all_handles) return dkmc_fail(47, "comm: peer exchange not prepared", __FILE__, __LINE__);
    // ranks on different devices: only with fine-grained buffers on EVERY rank and the explicit opt-in (never run on this pool: see above)
    {
        const char *mine = all_handles + (size_t)c.rank * 192 + 128;
        bool cross = false, all_fine = true;
        for (int r = 0; r < c.nranks; ++r) {
            const char *id = all_handles + (size_t)r * 192 + 128;
            if (strncmp(id + 1, mine + 1, 62) != 0) cross = true;
            if (id[0] != 'F') all_fine = false;
        }
        if (cross && !(all_fine && getenv("DKMC_PEER_CROSS_DEVICE") && getenv("DKMC_PEER_CROSS_DEVICE")[0] == '1')) {
            peer_release();
            return dkmc_fail(49, "comm: peer exchange refused: the ranks sit on different devices (validated on one device only; DKMC_PEER_CROSS_DEVICE=1 with "
                                 "fine-grained buffers on every rank overrides) -- the communicator's all-gather stays in use", __FILE__, __LINE__);
        }
    }
    double *rb[PEER_MAXR] = {}; unsigned long long *rf[PEER_MAXR] = {};
    for (int r = 0; r < c.nranks; ++r) {
        if (r == c.rank) { rb[r] = p.buf; rf[r] = p.flags; continue; }
        hipIpcMemHandle_t h[2];
        memcpy(&h[0], all_handles + (size_t)r * 192, 64); memcpy(&h[1], all_handles + (size_t)r * 192 + 64, 64);
        HIPCHK(hipIpcOpenMemHandle(&p.mapped[2 * r], h[0], hipIpcMemLazyEnablePeerAccess));
        HIPCHK(hipIpcOpenMemHandle(&p.mapped[2 * r + 1], h[1], hipIpcMemLazyEnablePeerAccess));
        rb[r] = (double *)p.mapped[2 * r]; rf[r] = (unsigned long long *)p.mapped[2 * r + 1];
    }
    HIPCHK(hipMalloc((void **)&p.rbuf_d, sizeof(rb))); HIPCHK(hipMalloc((void **)&p.rflags_d, sizeof(rf)));
    HIPCHK(hipMemcpy(p.rbuf_d, rb, sizeof(rb), hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(p.rflags_d, rf, sizeof(rf), hipMemcpyHostToDevice));
    HIPCHK(hipEventCreate(&p.ev[0])); HIPCHK(hipEventCreate(&p.ev[1]));
    p.seq = 0; p.ready = true;
    return 0;
}
extern "C" int dkmc_comm_peer_detach(void) { peer_release(); return 0; }
// a solve that failed while the exchange was in use (time-out, abort, launch error): the ranks' sequence counters may have drifted apart
// (a rank that timed out stops at the end of its batch, its peers queue further exchanges), so every later exchange would wait
// PEER_TIMEOUT_S for a sequence number the peer never reaches.  Drop the attachment: later solves use the communicator's all-gather
// until the host attaches again (which resets sequence numbers and flags on every rank).
void comm_peer_drop() { peer_release(); }
int comm_peer_finegrained() { return g_peer.ready && g_peer.finegrained ? 1 : 0; }
// ready; doubles per slot; exchanges since attach; mean duration of the timed ones [us] (profiling on: HIP events round push .. wait)
extern "C" int dkmc_comm_peer_info(int *ready, long long *slot_doubles, long long *exchanges, double *mean_us)
{
    const Peer &p = g_peer;
    if (ready) *ready = p.ready ? 1 : 0;
    if (slot_doubles) *slot_doubles = (long long)p.slot;
    if (exchanges) *exchanges = p.exchanges;
    if (mean_us) *mean_us = p.timed ? p.ms * 1e3 / p.timed : 0.0;
    return 0;
}

__global__ void k_peer_push(int me, double *const *rbuf, size_t off, size_t count)
{
    const int r = blockIdx.y;
    if (r == me) return;
    const double *src = rbuf[me] + off;
    double *dst = rbuf[r] + off;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
    __threadfence_system();
}
__global__ void k_peer_signal(int nr, int me, unsigned long long *const *rflags, unsigned long long seq)
{
    __threadfence_system();
    const int r = threadIdx.x;
    if (r < nr) __hip_atomic_store(&rflags[r][me], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void k_peer_wait(int nr, const unsigned long long *flags, unsigned long long seq, long long limit_ticks, int *ctrl_done, int *ctrl_aborted, int *ctrl_timeout, int stamp)
{
    const int r = threadIdx.x;
    bool ok = true;
    if (r < nr) {
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(&flags[r], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
            if (wall_clock64() - t0 > limit_ticks) { ok = false; break; }
            __builtin_amdgcn_s_sleep(32);
        }
    }
    if (!ok) { *ctrl_timeout = 1; *ctrl_aborted = 1; *ctrl_done = stamp; }      // every kernel of the loop is gated by `done`
    __threadfence_system();
}

int comm_peer_ready(size_t count) { return g_peer.ready && count <= g_peer.slot; }
// base of the nranks slots of this exchange's parity; slot r starts at r * count (count = the caller's doubles per rank <= the slot size)
double *comm_peer_slots(int parity) { return g_peer.buf + (size_t)(parity & 1) * g_comm.nranks * g_peer.slot; }
// this rank's slot of `parity` has been written by kernels already queued on the engine's stream; on return (asynchronously) all slots are complete
int comm_peer_exchange(int parity, size_t count, int *ctrl_done, int *ctrl_aborted, int *ctrl_timeout, int stamp)
{
    Comm &c = g_comm; Peer &p = g_peer; Engine &e = eng(); hipStream_t st = e.stream;
    if (!comm_peer_ready(count)) return dkmc_fail(47, "comm: peer exchange not attached or slot too small", __FILE__, __LINE__);
    const size_t off = (size_t)(parity & 1) * c.nranks * p.slot + (size_t)c.rank * count;
    // (slot r of a parity set starts at r * count: the layout the all-gather produces)
    const bool timed = e.profiling != 0 && (p.exchanges % 16) == 0;
    if (timed && p.exchanges >= 16) {          // the pair recorded 16 exchanges ago has long completed
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.ev[0], p.ev[1]) == hipSuccess) { p.ms += ms; ++p.timed; } else (void)hipGetLastError();
    }
    if (timed) HIPCHK(hipEventRecord(p.ev[0], st));
    ++p.seq; ++p.exchanges;
    const int bx = (int)std::min<size_t>((count + 255) / 256, 64);
    hipLaunchKernelGGL(k_peer_push, dim3(bx, c.nranks), dim3(256), 0, st, c.rank, (double *const *)p.rbuf_d, off, count);
    hipLaunchKernelGGL(k_peer_signal, dim3(1), dim3(64), 0, st, c.nranks, c.rank, (unsigned long long *const *)p.rflags_d, p.seq);
    hipLaunchKernelGGL(k_peer_wait, dim3(1), dim3(64), 0, st, c.nranks, (const unsigned long long *)p.flags, p.seq, (long long)(PEER_TIMEOUT_S * 1e8),
                       ctrl_done, ctrl_aborted, ctrl_timeout, stamp);
    if (timed) HIPCHK(hipEventRecord(p.ev[1], st));
    KCHK();
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

// All-to-all-v of doubles on the engine's stream (slab-distributed block-CG, xtb.hip): rank s sends cnt[s * nranks + d] doubles to rank d.
// cnt is the FULL table (every rank holds the same one: piece sizes derive from replicated data), so no size exchange is needed.  In a
// rank's send buffer the pieces lie in destination order, in its receive buffer in source order, both densely packed; a rank's piece for
// itself is copied on the device.
//   RCCL: one group of ncclSend / ncclRecv pairs -- on the xGMI mesh every pair of ranks has its own link, so the n - 1 pieces travel in
//         parallel (this is what RCCL's own all-to-all does); never run here with more than one rank (one-GPU box).
//   host: every rank's whole send buffer is all-gathered through the callback (padded to the longest) and the pieces are copied out of the
//         staging area -- rehearsal and tests only (several ranks on one GPU).
int comm_alltoallv_f64(const double *sendbuf, double *recvbuf, const long long *cnt)
{
    Comm &c = g_comm; hipStream_t st = eng().stream;
    const int nr = c.nranks, me = c.rank;
    if (c.transport == DKMC_COMM_NONE) return 0;
    long long soff[PEER_MAXR_A2A + 1], roff[PEER_MAXR_A2A + 1];
    if (nr > PEER_MAXR_A2A) return dkmc_fail(43, "comm: all-to-all-v supports at most 64 ranks", __FILE__, __LINE__);
    soff[0] = roff[0] = 0;
    for (int r = 0; r < nr; ++r) { soff[r + 1] = soff[r] + cnt[(size_t)me * nr + r]; roff[r + 1] = roff[r] + cnt[(size_t)r * nr + me]; }
    if (c.transport == DKMC_COMM_RCCL) {
        if (!c.send || !c.recv || !c.group_start || !c.group_end) return dkmc_fail(41, "comm: librccl lacks ncclSend / ncclRecv / ncclGroup*", __FILE__, __LINE__);
        if (cnt[(size_t)me * nr + me] > 0)
            HIPCHK(hipMemcpyAsync(recvbuf + roff[me], sendbuf + soff[me], (size_t)cnt[(size_t)me * nr + me] * 8, hipMemcpyDeviceToDevice, st));
        if (nr > 1) {
            RCCLCHK(c.group_start());
            for (int r = 0; r < nr; ++r) {
                if (r == me) continue;
                if (cnt[(size_t)me * nr + r] > 0) RCCLCHK(c.send(sendbuf + soff[r], (size_t)cnt[(size_t)me * nr + r], NCCL_FLOAT64, r, c.nccl, st));
                if (cnt[(size_t)r * nr + me] > 0) RCCLCHK(c.recv(recvbuf + roff[r], (size_t)cnt[(size_t)r * nr + me], NCCL_FLOAT64, r, c.nccl, st));
            }
            RCCLCHK(c.group_end());
        }
        return 0;
    }
    // host transport
    long long maxsend = 1;
    for (int s = 0; s < nr; ++s) { long long t = 0; for (int d = 0; d < nr; ++d) t += cnt[(size_t)s * nr + d]; if (t > maxsend) maxsend = t; }
    const size_t bytes = (size_t)nr * maxsend * sizeof(double);
    if (c.stage_bytes < bytes) {
        if (c.stage) (void)hipHostFree(c.stage);
        c.stage = nullptr; c.stage_bytes = 0;
        HIPCHK(hipHostMalloc((void **)&c.stage, bytes, hipHostMallocDefault));
        c.stage_bytes = bytes;
    }
    if (soff[nr] > 0) HIPCHK(hipMemcpyAsync(c.stage + (size_t)me * maxsend, sendbuf, (size_t)soff[nr] * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (int rc = c.cb(c.stage, (size_t)maxsend * sizeof(double), c.rank, c.nranks, c.cb_user)) return dkmc_fail(45, "comm: all-gather callback failed", __FILE__, __LINE__);
    for (int s = 0; s < nr; ++s) {
        const long long n = cnt[(size_t)s * nr + me];
        if (n <= 0) continue;
        long long o = 0; for (int d = 0; d < me; ++d) o += cnt[(size_t)s * nr + d];
        HIPCHK(hipMemcpyAsync(recvbuf + roff[s], c.stage + (size_t)s * maxsend + o, (size_t)n * 8, hipMemcpyHostToDevice, st));
    }
    HIPCHK(hipStreamSynchronize(st));          // the staging buffer is reused by the next call
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
